"""Pin the oracle against the REAL reference and write golden fixtures  (authoring container only).

Run:  python oracle/gen_golden.py            (needs /root/reference; never runs on the GPU box)

What it does
  1. injects a ~20-line `timm` stub and imports the reference's modules
     from /root/reference (model.model_dupl, model.PAR, model.losses, utils.cam_helper,
     utils.camutils, utils.optimizer) -- nothing is copied, the reference is only *called*;
  2. loads the same hash-generated weights into the reference modules and into the oracle's
     parameter dict, runs both on the same hash-generated inputs and asserts agreement
     (bit-exact for label / integer outputs, <=2e-5 relative for floating point);
  3. writes `tests/golden/*.npz`: inputs + the REFERENCE's outputs.  These are data only.

The fixtures are what `tests/test_oracle_golden.py` (CPU) and the `-m gpu` parity tests replay.
"""
from __future__ import annotations

import os
import sys
import types
from functools import partial

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

from oracle import dupl_oracle as O  # noqa: E402

REF = "/root/reference"
GOLD = os.path.join(ROOT, "tests", "golden")


def install_timm_stub():
    def mod(name):
        m = types.ModuleType(name)
        sys.modules[name] = m
        return m

    timm = mod("timm")
    data = mod("timm.data")
    models = mod("timm.models")
    helpers = mod("timm.models.helpers")
    layers = mod("timm.models.layers")
    registry = mod("timm.models.registry")
    data.IMAGENET_DEFAULT_MEAN = (0.485, 0.456, 0.406)
    data.IMAGENET_DEFAULT_STD = (0.229, 0.224, 0.225)
    helpers.load_pretrained = lambda *a, **k: None

    class DropPath(nn.Identity):
        def __init__(self, *a, **k):
            super().__init__()

    layers.DropPath = DropPath
    layers.to_2tuple = lambda x: x if isinstance(x, tuple) else (x, x)
    layers.trunc_normal_ = torch.nn.init.trunc_normal_
    models.resnet26d = None
    models.resnet50d = None
    registry.register_model = lambda f: f
    timm.data, timm.models = data, models
    models.helpers, models.layers, models.registry = helpers, layers, registry


def import_reference():
    install_timm_stub()
    sys.path.insert(0, REF)
    import model.backbone as backbone
    from model.backbone.vit import VisionTransformer
    from model.model_dupl import network, siamese_network
    from model.PAR import PAR
    from model import losses
    import utils.cam_helper as cam_helper
    import utils.camutils as camutils
    from utils.optimizer import PolyWarmupAdamW

    def tiny_test(pretrained=False, **kw):
        return VisionTransformer(patch_size=16, embed_dim=96, depth=4, num_heads=3, mlp_ratio=4,
                                 qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6),
                                 num_classes=10, **kw)

    backbone.tiny_test = tiny_test
    return dict(network=network, siamese=siamese_network, PAR=PAR, losses=losses,
                cam_helper=cam_helper, camutils=camutils, PolyWarmupAdamW=PolyWarmupAdamW)


def close(a, b, tol=2e-5, what=""):
    a, b = a.detach().double(), b.detach().double()
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-30)
    assert err <= tol * max(ref, 1.0), f"{what}: max-abs-err {err:.3e} (ref max {ref:.3e})"
    return err


def same(a, b, what="", budget=0):
    """Integer / label outputs must be identical.  `budget` > 0 is only used for the 448^2 refine
    maps, where the oracle's gather-based neighbour sum and the reference's conv2d-based one differ by
    ~2e-7 in the propagated masks and a pixel sitting on an argmax tie may flip (reported)."""
    assert a.shape == b.shape and a.dtype == b.dtype, f"{what}: {a.shape}{a.dtype} vs {b.shape}{b.dtype}"
    n = (a != b).sum().item()
    assert n <= budget, f"{what}: {n} mismatching elements (budget {budget})"
    if n:
        print(f"    note: {what}: {n} of {a.numel()} elements differ (argmax near-ties)")


def npz(name, **arrays):
    out = {}
    for k, v in arrays.items():
        out[k] = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
    path = os.path.join(GOLD, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"  wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


synth_cams = O.synthetic_cams


def main():
    torch.set_num_threads(8)
    os.makedirs(GOLD, exist_ok=True)
    R = import_reference()
    CH = R["cam_helper"]

    # ------------------------------------------------------------------ tiny student: forward
    print("[1] tiny student (embed 96, depth 4, heads 3): network.forward, cam_only, ms-CAM")
    cfg = O.VIT_TINY
    NC = 21
    sp = O.make_student_params(cfg, NC, seed=1)
    net = R["network"]("tiny_test", num_classes=NC, pretrained=False, aux_layer=-3)
    missing = net.load_state_dict(sp, strict=True)
    net.eval()
    x = O.hash_normal("tiny_x", (2, 3, 64, 96), std=1.0, seed=3)
    with torch.no_grad():
        r_cls, r_seg, r_x4, r_clsaux = net(x)
        r_camaux, r_cam = net(x, cam_only=True)
        o_cls, o_seg, o_x4, o_clsaux = O.network_forward(sp, x, cfg)
        o_camaux, o_cam = O.network_forward(sp, x, cfg, cam_only=True)
    for a, b_, n in ((o_cls, r_cls, "cls"), (o_seg, r_seg, "seg"), (o_x4, r_x4, "x4"),
                     (o_clsaux, r_clsaux, "cls_aux"), (o_camaux, r_camaux, "cam_aux"), (o_cam, r_cam, "cam")):
        print(f"    {n}: err {close(a, b_, what=n):.2e}")
    xs = O.hash_normal("tiny_xs", (2, 3, 64, 64), std=1.0, seed=4)
    r_mscam, r_mscam_aux = R["camutils"].multi_scale_cam2(net, xs, (1.0, 0.5, 1.5))
    o_mscam, o_mscam_aux = O.multi_scale_cam(sp, xs, cfg, (1.0, 0.5, 1.5))
    print(f"    ms-cam: err {close(o_mscam, r_mscam, what='mscam'):.2e} aux {close(o_mscam_aux, r_mscam_aux, what='mscam_aux'):.2e}")
    npz("tiny_forward", x=x, cls=r_cls, seg=r_seg, x4=r_x4, cls_aux=r_clsaux, cam_aux=r_camaux, cam=r_cam,
        xs=xs, mscam=r_mscam[:, ::4], mscam_aux=r_mscam_aux[:, ::4])

    # ------------------------------------------------------------------ tiny siamese: train step grads
    print("[2] tiny siamese: phase A / B loss assembly + gradients (reference functions composed as train_final_voc.py:194-456)")
    pp = O.make_siamese_params(cfg, NC, seed=2)
    sia = R["siamese"]("tiny_test", num_classes=NC, pretrained=False, aux_layer=-3)
    sia.load_state_dict(pp, strict=True)
    sia.train()
    S = 64
    inputs, cls_label, img_box = O.synthetic_batch(2, NC - 1, S, seed=5)
    par = R["PAR"](num_iter=10, dilations=[1, 2, 4, 8, 12, 24])
    args = O.StepArgs()

    from sklearn.mixture import GaussianMixture
    ce_criterion = nn.CrossEntropyLoss(ignore_index=args.ignore_index, reduction="none")

    def ref_step(n_iter, inputs=inputs, cls_label=cls_label, img_box=img_box, inputs_aug=None):
        sia.zero_grad()
        inputs_denorm = O.denormalize_img2(inputs.clone())   # utils.imutils not importable (torchvision)
        cams_1, cams_aux_1 = CH.multi_scale_cam2_siamese(sia, inputs=inputs, scales=args.cam_scales, branch=1)
        cams_2, cams_aux_2 = CH.multi_scale_cam2_siamese(sia, inputs=inputs, scales=args.cam_scales, branch=2)
        if n_iter < args.gmm_iters:
            res = sia(inputs)
        else:
            res = sia(torch.cat([inputs, inputs_aug], dim=0), need_sp=True)     # train_final_voc.py:291-295
            segs_1_aug, segs_2_aug = res["branch1_aug"], res["branch2_aug"]
        cls_1, segs_1, fmap_1, cls_aux_1 = res["branch1"]
        cls_2, segs_2, fmap_2, cls_aux_2 = res["branch2"]
        seg_lowres = (segs_1.detach().clone(), segs_2.detach().clone())
        msm = F.multilabel_soft_margin_loss
        cls_loss = msm(cls_1, cls_label) + msm(cls_aux_1, cls_label) + msm(cls_2, cls_label) + msm(cls_aux_2, cls_label)
        b, _, h, w = inputs.shape
        if n_iter < args.cam_iters:
            lab = []
            for ca, fm in ((cams_aux_1, fmap_1), (cams_aux_2, fmap_2)):
                rc = F.interpolate(ca, size=fm.shape[2:], mode="bilinear", align_corners=False)
                _, pl = CH.cam_to_label(rc.detach(), cls_label=cls_label, img_box=img_box, ignore_mid=True,
                                        bkg_thre=args.bkg_thre, high_thre=args.high_thre, low_thre=args.low_thre,
                                        ignore_index=args.ignore_index)
                lab.append(pl)
            seg_loss = torch.ones(1)
            refined = None
            reg_loss = torch.zeros(1)
            extra_c = {}
        else:
            thr = O.cosine_descent(torch.ones(NC - 1) * args.high_thre, torch.tensor(args.high_target),
                                   n_iter - args.cam_iters, args.max_iters - args.cam_iters)
            hl, hm = [], []
            for i in range(b):
                t = torch.max(thr[torch.nonzero(cls_label[i]).squeeze(-1)])
                hl.append(t)
                hm.append(torch.ones((h, w)) * t)
            high_thres = torch.stack(hl, dim=0)
            high_mask = torch.stack(hm, dim=0).unsqueeze(1)
            lab = []
            for ca, fm in ((cams_aux_1, fmap_1), (cams_aux_2, fmap_2)):
                rc = F.interpolate(ca, size=fm.shape[2:], mode="bilinear", align_corners=False)
                _, pl = CH.cam_to_label_dynamic_cls(rc.detach(), cls_label=cls_label, img_box=img_box, ignore_mid=True,
                                                    bkg_thre=args.bkg_thre, high_thre=high_thres,
                                                    low_thre=args.low_thre, ignore_index=args.ignore_index)
                lab.append(pl)
            rep = cls_label.unsqueeze(-1).unsqueeze(-1).repeat([1, 1, h, w])
            r1 = CH.refine_cams_with_dynamic_thres(par, inputs_denorm, cams=cams_1.detach() * rep, cls_labels=cls_label,
                                                   high_thre_map=high_mask, low_thre=args.low_thre,
                                                   ignore_index=args.ignore_index, img_box=img_box)
            r2 = CH.refine_cams_with_dynamic_thres(par, inputs_denorm, cams=cams_2.detach() * rep, cls_labels=cls_label,
                                                   high_thre_map=high_mask, low_thre=args.low_thre,
                                                   ignore_index=args.ignore_index, img_box=img_box)
            s1 = F.interpolate(segs_1, size=r1.shape[1:], mode="bilinear", align_corners=False)
            s2 = F.interpolate(segs_2, size=r2.shape[1:], mode="bilinear", align_corners=False)
            reg_loss = torch.zeros(1)
            extra_c = {}
            if n_iter < args.gmm_iters:
                seg_loss = R["losses"].get_seg_loss(s1, r2.type(torch.long)) + R["losses"].get_seg_loss(s2, r1.type(torch.long))
            else:
                # ---- phase C, composed exactly as train_final_voc.py:358-436
                refined_pre = (r1.clone(), r2.clone())
                sl1 = ce_criterion(s1, r1.type(torch.long)).detach()
                sl2 = ce_criterion(s2, r2.type(torch.long)).detach()
                roi1 = (r1 != 0).bool() & (r1 != 255).bool()
                roi2 = (r2 != 0).bool() & (r2 != 255).bool()
                hits = [0, 0]
                for i in range(b):
                    for k, (sl, roi, rr) in enumerate(((sl1, roi1, r1), (sl2, roi2, r2))):
                        m = sl[i][roi[i]]
                        if (m > 0.1).sum().item() > 1000:
                            gmm = GaussianMixture(n_components=2, max_iter=10, tol=1e-2, reg_covar=5e-4, random_state=0)
                            gmm.fit(m[m > 0.1].unsqueeze(-1).cpu().detach().numpy())
                            means = gmm.means_
                            if abs(means[0, 0] - means[1, 0]) > 1.0:
                                noise_idx = gmm.means_.argmax()
                                prob = gmm.predict_proba(sl[i].view(-1).unsqueeze(-1).cpu().detach().numpy())
                                noise_mask = torch.tensor(prob[:, noise_idx] > 0.95).reshape(h, w)
                                noise_mask = noise_mask & (rr[i] != 0).bool()
                                rr[i][noise_mask] = 255
                                hits[k] += 1
                seg_loss_1 = R["losses"].get_seg_loss(s1, r2.type(torch.long), ignore_index=args.ignore_index)
                seg_loss_2 = R["losses"].get_seg_loss(s2, r1.type(torch.long), ignore_index=args.ignore_index)
                seg_loss = seg_loss_1 + seg_loss_2
                sa1 = torch.flip(segs_1_aug, dims=[3])
                sa2 = torch.flip(segs_2_aug, dims=[3])
                sa1 = F.interpolate(sa1, size=inputs_denorm.shape[2:], mode="bilinear", align_corners=False)
                sa2 = F.interpolate(sa2, size=inputs_denorm.shape[2:], mode="bilinear", align_corners=False)
                pseudo_seg_1 = s1.detach().data.max(1)[1]
                pseudo_seg_2 = s2.detach().data.max(1)[1]
                conf1 = torch.softmax(s1.detach(), dim=1).max(1)[0]
                conf2 = torch.softmax(s2.detach(), dim=1).max(1)[0]
                un1 = (r2 == args.ignore_index).bool() & (conf1 > 0.9)
                un2 = (r1 == args.ignore_index).bool() & (conf2 > 0.9)
                pseudo_seg_1[~un1] = args.ignore_index
                pseudo_seg_2[~un2] = args.ignore_index
                reg_1, reg_2 = seg_loss_1 * 0.0, seg_loss_2 * 0.0
                if un1.sum() > 0:
                    reg_1 = (ce_criterion(sa1, pseudo_seg_1)).sum() / un1.sum()
                if un2.sum() > 0:
                    reg_2 = (ce_criterion(sa2, pseudo_seg_2)).sum() / un2.sum()
                reg_loss = reg_1 + reg_2
                extra_c = dict(refined_pre_1=refined_pre[0].to(torch.uint8), refined_pre_2=refined_pre[1].to(torch.uint8),
                               ce_map_1=sl1[:, ::2, ::2].clone(), gmm_hits=np.array(hits), pseudo_seg_1=pseudo_seg_1.to(torch.uint8),
                               pseudo_seg_2=pseudo_seg_2.to(torch.uint8), n_uncertain=np.array([int(un1.sum()), int(un2.sum())]),
                               reg_loss=reg_loss.detach(), segs_1_aug=segs_1_aug.detach(), segs_2_aug=segs_2_aug.detach())
            refined = (r1, r2)
        ptc = R["losses"].get_masked_ptc_loss(fmap_1, CH.label_to_aff_mask(lab[0])) + \
            R["losses"].get_masked_ptc_loss(fmap_2, CH.label_to_aff_mask(lab[1]))
        f1 = fmap_1.view(fmap_1.shape[0], fmap_1.shape[1], -1)
        f2 = fmap_2.view(fmap_2.shape[0], fmap_2.shape[1], -1)
        cs = nn.CosineSimilarity(dim=-1, eps=1e-6)
        sim = (1 + cs(f1.detach(), f2).mean()) + (1 + cs(f2.detach(), f1).mean())
        if n_iter <= args.cam_iters:
            loss = 1.0 * cls_loss + args.w_ptc * ptc + 0.0 * seg_loss + 0.1 * sim
        elif n_iter <= args.gmm_iters:
            loss = 1.0 * cls_loss + args.w_ptc * ptc + args.w_seg * seg_loss + 0.1 * sim + 0.00 * reg_loss
        else:
            loss = 1.0 * cls_loss + args.w_ptc * ptc + args.w_seg * seg_loss + 0.1 * sim + 0.05 * reg_loss
        loss.backward()
        grads = {k: (v.grad.clone() if v.grad is not None else None) for k, v in sia.named_parameters()}
        return dict(loss=loss.detach(), cls_loss=cls_loss.detach(), ptc=ptc.detach(), seg=seg_loss.detach(),
                    sim=sim.detach(), lab=lab, refined=refined, cams=(cams_1, cams_aux_1, cams_2, cams_aux_2),
                    grads=grads, segs=seg_lowres, fmaps=(fmap_1.detach(), fmap_2.detach()),
                    extra_c=(extra_c if n_iter >= args.cam_iters else {}))

    for tag, n_iter in (("A", 100), ("B", 5000)):
        ref = ref_step(n_iter)
        leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] != "encoder.pos_embed") for k, v in pp.items()}
        loss, pc = O.train_step_losses(leaf, inputs, cls_label, img_box, n_iter, cfg, args)
        loss.backward()
        print(f"    phase {tag}: loss ref {ref['loss'].item():.6f} oracle {loss.item():.6f}")
        close(loss, ref["loss"], what="loss")
        close(pc["ptc_loss"], ref["ptc"], what="ptc")
        close(pc["seg_loss"], ref["seg"], what="seg")
        close(pc["sim_loss"], ref["sim"], what="sim")
        same(pc["pseudo_label_aux_1"], ref["lab"][0], "pseudo_label_aux_1")
        same(pc["pseudo_label_aux_2"], ref["lab"][1], "pseudo_label_aux_2")
        close(pc["cams_1"], ref["cams"][0], what="cams_1")
        if ref["refined"] is not None:
            same(pc["refined_1"], ref["refined"][0], "refined_1")
            same(pc["refined_2"], ref["refined"][1], "refined_2")
        worst = 0.0
        gsave = {}
        for k, g in ref["grads"].items():
            og = leaf[k].grad
            if g is None:
                assert og is None or og.abs().max() == 0, k
                continue
            e = (og - g).abs().max().item() / max(g.abs().max().item(), 1e-12)
            worst = max(worst, e)
            # big decoder tensors are stored as a flat strided subsample (tensors > 4096 elements: stride 7) to keep fixtures small
            gsave["grad." + k] = g if g.numel() <= 4096 else g.reshape(-1)[::7].clone()
        print(f"      worst relative grad err over {len(gsave)} tensors: {worst:.2e}")
        assert worst < 5e-4
        extra = {}
        if ref["refined"] is not None:
            extra = dict(refined_1=ref["refined"][0].to(torch.uint8), refined_2=ref["refined"][1].to(torch.uint8))
        npz(f"tiny_step_{tag}", inputs=inputs, cls_label=cls_label, img_box=img_box, n_iter=n_iter,
            loss=ref["loss"], cls_loss=ref["cls_loss"], ptc_loss=ref["ptc"], seg_loss=ref["seg"], sim_loss=ref["sim"],
            pseudo_label_aux_1=ref["lab"][0].to(torch.uint8), pseudo_label_aux_2=ref["lab"][1].to(torch.uint8),
            cams_1=ref["cams"][0][:, ::4], cams_aux_1=ref["cams"][1][:, ::4], cams_2=ref["cams"][2][:, ::4],
            cams_aux_2=ref["cams"][3][:, ::4],
            segs_1=ref["segs"][0], segs_2=ref["segs"][1], fmap_1=ref["fmaps"][0], fmap_2=ref["fmaps"][1],
            **extra, **gsave)

    print("[2c] tiny siamese: phase C (GMM noise filter via sklearn + consistency regularisation), 128^2 inputs")
    inputs_c, cls_c, box_c = O.synthetic_batch(2, NC - 1, 128, seed=9)
    aug_c, _, _ = O.synthetic_batch(2, NC - 1, 128, seed=19)
    aug_c = torch.flip(0.7 * inputs_c + 0.3 * aug_c, dims=[3]).contiguous()   # stand-in for RandAugment + flip (imutils.py:305-317)
    n_c = 9000
    # random-init logits are never confident: sharpen the seg heads so that the GMM filter (needs > 1000 roi pixels with
    # CE > 0.1 and two well separated modes) and the confidence-gated consistency loss are actually exercised
    pp_c = {k: (v * 40.0 if k.endswith("decoder.conv8.weight") else v.clone()) for k, v in pp.items()}
    sia.load_state_dict(pp_c, strict=True)
    ref = ref_step(n_c, inputs_c, cls_c, box_c, aug_c)
    sia.load_state_dict(pp, strict=True)
    leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] != "encoder.pos_embed") for k, v in pp_c.items()}
    loss, pc = O.train_step_losses(leaf, inputs_c, cls_c, box_c, n_c, cfg, args, inputs_aug=aug_c)
    loss.backward()
    print(f"    phase C: loss ref {ref['loss'].item():.6f} oracle {loss.item():.6f}; gmm hits {ref['extra_c']['gmm_hits']}, "
          f"uncertain px {ref['extra_c']['n_uncertain']}, reg {ref['extra_c']['reg_loss'].item():.6f}")
    close(loss, ref["loss"], what="loss C")
    close(pc["reg_loss"], ref["extra_c"]["reg_loss"], what="reg C")
    same(pc["refined_1"], ref["refined"][0], "refined_1 C", budget=2)
    same(pc["pseudo_seg_1"].to(torch.uint8), ref["extra_c"]["pseudo_seg_1"], "pseudo_seg_1 C")
    worst, gsave = 0.0, {}
    for k, g in ref["grads"].items():
        og = leaf[k].grad
        if g is None:
            continue
        worst = max(worst, (og - g).abs().max().item() / max(g.abs().max().item(), 1e-12))
        gsave["grad." + k] = g if g.numel() <= 4096 else g.reshape(-1)[::7].clone()
    print(f"      worst relative grad err over {len(gsave)} tensors: {worst:.2e}")
    assert worst < 5e-4
    ec = ref["extra_c"]
    npz("tiny_step_C", n_iter=n_c, loss=ref["loss"], cls_loss=ref["cls_loss"], ptc_loss=ref["ptc"], seg_loss=ref["seg"],
        sim_loss=ref["sim"], reg_loss=ec["reg_loss"], refined_1=ref["refined"][0].to(torch.uint8),
        refined_2=ref["refined"][1].to(torch.uint8), refined_pre_1=ec["refined_pre_1"], refined_pre_2=ec["refined_pre_2"],
        ce_map_1_sub=ec["ce_map_1"], gmm_hits=ec["gmm_hits"], pseudo_seg_1=ec["pseudo_seg_1"], pseudo_seg_2=ec["pseudo_seg_2"],
        n_uncertain=ec["n_uncertain"], segs_1=ref["segs"][0], segs_2=ref["segs"][1], segs_1_aug=ec["segs_1_aug"],
        segs_2_aug=ec["segs_2_aug"], **gsave)

    # ------------------------------------------------------------------ optimiser
    print("[3] PolyWarmupAdamW: 3 steps on two small tensors")
    w0 = O.hash_normal("opt_w", (7, 5), seed=1)
    w1 = O.hash_normal("opt_b", (11,), seed=1)
    gs = [(O.hash_normal(f"opt_gw{t}", (7, 5), seed=1), O.hash_normal(f"opt_gb{t}", (11,), seed=1)) for t in range(3)]
    pa, pb = nn.Parameter(w0.clone()), nn.Parameter(w1.clone())
    opt = R["PolyWarmupAdamW"](params=[{"params": [pa], "lr": 6e-5, "weight_decay": 0.01},
                                       {"params": [pb], "lr": 6e-4, "weight_decay": 0.01}],
                               lr=6e-5, weight_decay=0.01, betas=(0.9, 0.999), warmup_iter=2, max_iter=20,
                               warmup_ratio=1e-6, power=0.9)
    oa, ob = w0.clone(), w1.clone()
    st = [(torch.zeros_like(oa), torch.zeros_like(oa)), (torch.zeros_like(ob), torch.zeros_like(ob))]
    traj = []
    for t in range(3):
        pa.grad, pb.grad = gs[t][0].clone(), gs[t][1].clone()
        opt.step()
        mult = O.poly_warmup_lr_mult(t, 2, 20, 1e-6, 0.9)
        O.adamw_update(oa, gs[t][0], st[0][0], st[0][1], t + 1, 6e-5 * mult)
        O.adamw_update(ob, gs[t][1], st[1][0], st[1][1], t + 1, 6e-4 * mult)
        close(oa, pa.data, 1e-6, "adamw a")
        close(ob, pb.data, 1e-6, "adamw b")
        traj.append((pa.data.clone(), pb.data.clone()))
    npz("adamw", w0=w0, w1=w1, **{f"gw{t}": gs[t][0] for t in range(3)}, **{f"gb{t}": gs[t][1] for t in range(3)},
        **{f"pa{t}": traj[t][0] for t in range(3)}, **{f"pb{t}": traj[t][1] for t in range(3)})

    # ------------------------------------------------------------------ PAR / refine / labels at full size
    print("[4] PAR, refine (both variants), cam_to_label, aff mask, PTC, seg loss at 448^2 (C=20)")
    b, C, S = 2, 20, 448
    inputs, cls_label, img_box = O.synthetic_batch(b, C, S, seed=7)
    img_dn = O.denormalize_img2(inputs.clone())
    cams = synth_cams(b, C, S, S, seed=8)
    rep = cls_label[:, :, None, None]
    par = R["PAR"](num_iter=10, dilations=[1, 2, 4, 8, 12, 24])
    # PAR alone on one half-res image with K=3 masks
    img_half = F.interpolate(img_dn[:1], size=[S // 2, S // 2], mode="bilinear", align_corners=False)
    m0 = synth_cams(1, 3, S // 2, S // 2, seed=9).softmax(dim=1)
    with torch.no_grad():
        r_par = par(img_half, m0)
    o_par = O.par_forward(img_half, m0)
    print(f"    PAR: err {close(o_par, r_par, 1e-5, 'par'):.2e}")
    o_aff = O.par_affinity(img_half)
    r_v2 = CH.refine_cams_with_bkg_v2(par, img_dn, cams=cams * rep, cls_labels=cls_label, high_thre=0.65,
                                      low_thre=0.25, ignore_index=255, img_box=img_box)
    o_v2 = O.refine_cams(img_dn, cams * rep, cls_label, 0.65, 0.25, 255, img_box)
    same(o_v2, r_v2, "refine_v2", budget=4)
    hm = torch.stack([torch.ones(S, S) * 0.62, torch.ones(S, S) * 0.68]).unsqueeze(1)
    r_dyn = CH.refine_cams_with_dynamic_thres(par, img_dn, cams=cams * rep, cls_labels=cls_label, high_thre_map=hm,
                                              low_thre=0.25, ignore_index=255, img_box=img_box)
    o_dyn = O.refine_cams(img_dn, cams * rep, cls_label, hm, 0.25, 255, img_box)
    same(o_dyn, r_dyn, "refine_dyn", budget=4)
    print("    refine v2 / dynamic: identical label maps; label histogram", torch.unique(r_dyn, return_counts=True))
    # cam_to_label at 28x28 and 448x448
    c28 = F.interpolate(cams, size=(28, 28), mode="bilinear", align_corners=False)
    box28 = img_box.clone()
    r_valid, r_l28 = CH.cam_to_label(c28.clone(), cls_label=cls_label, img_box=img_box, ignore_mid=True, bkg_thre=0.5,
                                     high_thre=0.7, low_thre=0.25, ignore_index=255)
    o_valid, o_l28 = O.cam_to_label(c28.clone(), cls_label, img_box=img_box, ignore_mid=True, bkg_thre=0.5,
                                    high_thre=0.7, low_thre=0.25, ignore_index=255)
    same(o_l28, r_l28, "cam_to_label 28")
    box_small = torch.tensor([[0, 28, 0, 28], [3, 25, 5, 20]], dtype=torch.int16)
    hd = torch.tensor([0.55, 0.66])
    _, r_l28d = CH.cam_to_label_dynamic_cls(c28.clone(), cls_label=cls_label, img_box=box_small, ignore_mid=True,
                                            bkg_thre=0.5, high_thre=hd, low_thre=0.25, ignore_index=255)
    _, o_l28d = O.cam_to_label(c28.clone(), cls_label, img_box=box_small, ignore_mid=True, bkg_thre=0.5,
                               high_thre=hd, low_thre=0.25, ignore_index=255)
    same(o_l28d, r_l28d, "cam_to_label_dynamic 28")
    r_lfull = CH.cam_to_label(cams.clone(), cls_label=cls_label, bkg_thre=0.45)
    o_lfull = O.cam_to_label(cams.clone(), cls_label, bkg_thre=0.45)
    same(o_lfull, r_lfull, "cam_to_label full (no box)")
    r_aff = CH.label_to_aff_mask(r_l28d)
    o_affm = O.label_to_aff_mask(o_l28d)
    same(o_affm, r_aff, "aff mask")
    fmap = O.hash_normal("fmap", (b, 64, 28, 28), seed=10)
    r_ptc = R["losses"].get_masked_ptc_loss(fmap, r_aff)
    o_ptc = O.masked_ptc_loss(fmap, o_affm)
    close(o_ptc, r_ptc, 1e-6, "ptc")
    seg = O.hash_normal("seglogit", (b, C + 1, 28, 28), std=2.0, seed=11)
    segu = F.interpolate(seg, size=(S, S), mode="bilinear", align_corners=False)
    r_sl = R["losses"].get_seg_loss(segu, r_dyn.type(torch.long))
    o_sl = O.seg_loss(segu, r_dyn.long())
    close(o_sl, r_sl, 1e-6, "seg loss")
    print(f"    ptc {r_ptc.item():.6f}  seg-loss {r_sl.item():.6f}")
    # inputs / cams are NOT stored: tests regenerate them with O.synthetic_batch(2,20,448,seed=7) /
    # O.synthetic_cams(2,20,448,448,seed=8) (platform-independent generators)
    npz("labels_448", cls_label=cls_label, img_box=img_box, img_u8_checksum=np.int64((img_dn * 255).round().long().sum().item()),
        cams_checksum=np.float64(cams.double().sum().item()),
        refine_v2=r_v2.to(torch.uint8), refine_dyn=r_dyn.to(torch.uint8), high_map_vals=np.array([0.62, 0.68], np.float32),
        label28=r_l28.to(torch.uint8), label28_dyn=r_l28d.to(torch.uint8), box_small=box_small, high_dyn=hd,
        label_full=r_lfull.to(torch.uint8), valid28=r_valid,
        fmap=fmap, ptc=r_ptc, seg_logits=seg, seg_loss=r_sl)
    npz("par_224", out_sub=r_par[:, :, ::2, ::2], out_sum=r_par.double().sum(dim=(2, 3)), aff_sub=o_aff[0, 0, :, ::8, ::8])

    # ------------------------------------------------------------------ ViT-B, one student, 224^2
    print("[5] ViT-B/16 single student at 224^2 (config 1 shape): network.forward + cam_only")
    cfgb = O.VIT_BASE
    spb = O.make_student_params(cfgb, NC, seed=11)
    netb = R["network"]("deit_base_patch16_224", num_classes=NC, pretrained=False, aux_layer=-3)
    netb.load_state_dict(spb, strict=True)
    netb.eval()
    xb, _, _ = O.synthetic_batch(2, 20, 224, seed=12)
    with torch.no_grad():
        rb = netb(xb)
        rbc = netb(xb, cam_only=True)
        ob = O.network_forward(spb, xb, cfgb)
        obc = O.network_forward(spb, xb, cfgb, cam_only=True)
    for a, b_, n in zip(ob + obc, rb + rbc, ("cls", "seg", "x4", "cls_aux", "cam_aux", "cam")):
        print(f"    {n}: err {close(a, b_, what=n):.2e}")
    npz("vitb_224", cls=rb[0], seg=rb[1], x4_sub=rb[2][:, ::16], cls_aux=rb[3], cam_aux=rbc[0], cam=rbc[1])
    print("oracle pinned against the reference; fixtures written.")


if __name__ == "__main__":
    main()
