"""One DuPL training iteration (reference: train_final_voc.py:174-472 / train_final_coco.py:170-462).

`compute_losses` is the loss assembly of phases A and B (and the GMM-free part of C) written against the
reference's own module API (siamese_network.forward, cam_helper.*, losses.*), so it reads like the
reference's loop; `train_step` adds zero_grad / backward / optimiser step.  The per-step host syncs of
the reference (six .item() calls + an sklearn F1 on .cpu() tensors, train_final_voc.py:458-468) are not
part of the step: callers log from the returned device scalars when they want to.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import ops
from .model import losses as LS
from .utils import cam_helper, imutils
from .utils.train_helper import cosine_descent

VOC_HIGH_TARGET = (0.70, 0.70, 0.70, 0.70, 0.55, 0.55, 0.55, 0.55, 0.70, 0.55,
                   0.55, 0.55, 0.55, 0.55, 0.55, 0.55, 0.55, 0.55, 0.70, 0.55)   # train_final_voc.py:162-166


@dataclass
class StepArgs:
    """The argparse fields the iteration reads (defaults = train_final_voc.py:35-90)."""
    cam_iters: int = 2000
    gmm_iters: int = 8000
    max_iters: int = 20000
    bkg_thre: float = 0.5
    high_thre: float = 0.7
    low_thre: float = 0.25
    ignore_index: int = 255
    w_ptc: float = 0.2
    w_seg: float = 0.2
    cam_scales: Tuple[float, ...] = (1.0, 0.5, 1.5)
    high_target: Tuple[float, ...] = VOC_HIGH_TARGET
    samples_per_gpu: int = 2
    gmm_valid_thre: float = 1.0
    gamma: float = 0.95
    strong_aug_n: int = 5        # train_final_voc.py:191: imutils.augment_data_strong(inputs_denorm.clone(), n=5, m=10)
    strong_aug_m: int = 10
    share_encoder_pass: bool = True   # reuse ms-CAM's scale-1.0 encoder pass as the training forward (identical values)
    schedule: str = "voc"        # "voc": train_final_voc.py:194-456; "coco": train_final_coco.py:190-448
    coco_switch_iter: int = 12000  # train_final_coco.py:241,312: bkg_v2 on aux CAMs until here, then dynamic thresholds


def coco_step_args(**kw) -> StepArgs:
    """train_final_coco.py:75-86,161-162: cam 8000 / gmm 32000 / max 80000, bkg 0.45, high 0.65, target 0.55 x 80."""
    d = dict(cam_iters=8000, gmm_iters=32000, max_iters=80000, bkg_thre=0.45, high_thre=0.65, low_thre=0.25,
             high_target=tuple([0.55] * 80), schedule="coco")
    d.update(kw)
    return StepArgs(**d)


def per_image_high_thres(cls_label: torch.Tensor, n_iter: int, args: StepArgs, device=None) -> torch.Tensor:
    """train_final_voc.py:263-275: cosine-descended per-class thresholds, per image the max over present classes.
    Host-side (C floats) -> (b,) device tensor.  `cls_label` should be the HOST copy of the labels (no sync)."""
    C = cls_label.shape[1]
    off = args.cam_iters if args.schedule == "voc" else args.coco_switch_iter   # train_final_coco.py:241
    thr = cosine_descent(np.ones(C, dtype=np.float32) * np.float32(args.high_thre),
                         np.asarray(args.high_target[:C], dtype=np.float32),
                         n_iter - off, args.max_iters - off)
    thr = np.asarray(thr, dtype=np.float32)
    cl = cls_label.detach().cpu().numpy() > 0
    out = np.array([thr[cl[i]].max() if cl[i].any() else thr.max() for i in range(cl.shape[0])], dtype=np.float32)
    device = device if device is not None else cls_label.device
    return ops.to_device_async(out, torch.float32, device)


def compute_losses(model, par, inputs, cls_label, img_box, n_iter: int, args: StepArgs, cls_label_host=None,
                   inputs_aug=None):
    """Loss assembly of one iteration, phases A / B / C; returns (loss, dict of device scalars / tensors).
    `cls_label_host`: CPU copy of `cls_label` (the data loader has it anyway); with it phases A and B contain no
    host<->device synchronisation at all and the host can run a full step ahead of the GPU.
    Phase C (n_iter >= gmm_iters, train_final_voc.py:358-436) uses `inputs_aug`, the strongly augmented + w-flipped
    batch (train_final_voc.py:191): computed here on the device (utils/imutils.augment_data_strong)
    unless given; its GMM label-noise filter runs on the device (csrc/gmm.hip),
    so phase C has no host<->device synchronisation either."""
    if cls_label_host is None:
        cls_label_host = cls_label.detach().cpu()
    phase_c = n_iter >= args.gmm_iters
    coco = args.schedule == "coco"
    b, _, h, w = inputs.shape
    phase_a = n_iter < args.cam_iters
    inputs_denorm = ops.denormalize_img(inputs.contiguous()) if not phase_a else None
    if phase_c and inputs_aug is None:
        # train_final_voc.py:191: RandAugment(n=5, m=10) + w-flip of the de-normalised batch, here without leaving the
        # device (the reference evaluates it every iteration and only uses it from gmm_iters on)
        inputs_aug = imutils.augment_data_strong(inputs_denorm, n=args.strong_aug_n, m=args.strong_aug_m)

    core = model.module if hasattr(model, "module") else model
    if not args.share_encoder_pass:
        (cams_1, cams_aux_1), (cams_2, cams_aux_2) = core.per_student(
            lambda: cam_helper.multi_scale_cam2_siamese(model, inputs=inputs, scales=args.cam_scales, branch=1),
            lambda: cam_helper.multi_scale_cam2_siamese(model, inputs=inputs, scales=args.cam_scales, branch=2))
        if phase_c:
            res = model(torch.cat([inputs, inputs_aug], dim=0), need_sp=True)     # train_final_voc.py:291-295
        else:
            res = model(inputs)
    else:
        # the scale-1.0 un-flipped ms-CAM encoder pass and the training forward see the same weights and input:
        # run it once, with activation saving, and feed both (reference: cam_helper.py:171 then train_final_voc.py:204)
        (cams_1, cams_aux_1), (cams_2, cams_aux_2), res = core.ms_cam_and_forward(
            inputs, args.cam_scales, inputs_aug=inputs_aug if phase_c else None)
    cls_1, segs_1, fmap_1, cls_aux_1 = res["branch1"]
    cls_2, segs_2, fmap_2, cls_aux_2 = res["branch2"]

    msm = LS.multilabel_soft_margin_loss
    # the scalar arithmetic of the loss assembly (train_final_voc.py:210-216,247-254,451-456) is ONE launch at the end of this function
    # (LS.weighted_total: the same fp32 operations in the same order); here only the terms are collected
    cls_terms = [msm(cls_1, cls_label), msm(cls_aux_1, cls_label), msm(cls_2, cls_label), msm(cls_aux_2, cls_label)]

    fh, fw = fmap_1.shape[2:]
    out = {"cams_1": cams_1, "cams_2": cams_2, "cams_aux_1": cams_aux_1, "cams_aux_2": cams_aux_2}
    high = None
    if phase_a and coco:
        ptc_terms = [torch.ones(1, device=inputs.device)]     # train_final_coco.py:216: no PTC in phase A
    else:
        if phase_a:
            high = args.high_thre
            to_label = cam_helper.cam_to_label
        else:
            high = per_image_high_thres(cls_label_host, n_iter, args, device=inputs.device)
            to_label = cam_helper.cam_to_label_dynamic_cls
        # per student on its own stream (round 4): the label / PTC / refinement kernels are small (a PAR iteration is 46 us on a
        # fraction of the chip), so the two students' chains overlap almost completely instead of queueing on one stream
        def aux_label_and_ptc(ca, fmap):
            r = ops.resize_bilinear(ca, fh, fw)
            _, pl = to_label(r, cls_label=cls_label, img_box=img_box, ignore_mid=True, bkg_thre=args.bkg_thre,
                             high_thre=high, low_thre=args.low_thre, ignore_index=args.ignore_index)
            return pl, LS.get_masked_ptc_loss_from_label(fmap, pl, args.ignore_index)
        (l1, p1), (l2, p2) = core.per_student(lambda: aux_label_and_ptc(cams_aux_1, fmap_1),
                                              lambda: aux_label_and_ptc(cams_aux_2, fmap_2))
        labels = [l1, l2]
        ptc_terms = [p1, p2]
        out["pseudo_label_aux_1"], out["pseudo_label_aux_2"] = labels
    reg_terms = None
    if phase_a:
        seg_terms = [torch.ones(1, device=inputs.device)]
    else:
        # the reference passes cams * cls_label_rep (train_final_voc.py:336); refine only reads the channels of
        # PRESENT classes (label == 1), for which that product is the identity, so the (b,C,H,W) multiply is skipped
        # the colour affinity depends on the images only: once per image for both students (the reference rebuilds it in
        # each of its four PAR calls per image, PAR.py:52), on the stream both students' streams fork from
        aff = cam_helper.par_affinity_of(par, inputs_denorm)
        if coco and n_iter <= args.coco_switch_iter:
            # train_final_coco.py:312-322: scalar high threshold on the AUX CAMs
            def refine(cams_aux_k):
                return cam_helper.refine_cams_with_bkg_v2(par, inputs_denorm, cams=cams_aux_k, cls_labels=cls_label_host,
                                                          high_thre=args.high_thre, low_thre=args.low_thre,
                                                          ignore_index=args.ignore_index, img_box=img_box, aff=aff)
            ref_in = (cams_aux_1, cams_aux_2)
        else:
            hmap = high.view(b, 1, 1, 1).expand(b, 1, h, w).contiguous()
            def refine(cams_k):
                return cam_helper.refine_cams_with_dynamic_thres(par, inputs_denorm, cams=cams_k, cls_labels=cls_label_host,
                                                                 high_thre_map=hmap, low_thre=args.low_thre,
                                                                 ignore_index=args.ignore_index, img_box=img_box, aff=aff)
            ref_in = (cams_1, cams_2)

        def refine_and_filter(cams_k, segs_k):
            r_k = refine(cams_k)
            st = None
            if phase_c:
                # GMM label-noise filter on the detached per-pixel CE of each student w.r.t. ITS OWN labels (:360-394)
                # one workgroup per image fits sklearn's 2-component mixture on the device (csrc/gmm.hip): no host round trip
                ce = LS.seg_ce_map(segs_k, r_k, (h, w), args.ignore_index)
                st = LS.gmm_noise_filter_(ce, r_k, args.ignore_index, args.gmm_valid_thre, args.gamma)
            return r_k, st
        (r1, st1), (r2, st2) = core.per_student(lambda: refine_and_filter(ref_in[0], segs_1),
                                                lambda: refine_and_filter(ref_in[1], segs_2))
        if phase_c:
            out["gmm_stats"] = [st1, st2]      # (b, 16) per student; column 1 = image was filtered
        # cross supervision: student 1 learns from student 2's labels and vice versa (train_final_voc.py:351-352)
        sl1, sl2 = core.per_student(lambda: LS.get_seg_loss_lowres(segs_1, r2, (h, w), args.ignore_index),
                                    lambda: LS.get_seg_loss_lowres(segs_2, r1, (h, w), args.ignore_index))
        seg_terms = [sl1, sl2]
        out["refined_1"], out["refined_2"] = r1, r2
        if phase_c:
            # consistency regularisation on the 0.75x strong-aug branch (:407-436)
            def pseudo_and_reg(segs_k, r_other, aug_k):
                ps, n = LS.seg_pseudo_label(segs_k, r_other, (h, w), args.ignore_index, 0.9)
                return ps, n, LS.get_reg_loss(aug_k, ps, (h, w), args.ignore_index)
            (ps1, n1, g1), (ps2, n2, g2) = core.per_student(lambda: pseudo_and_reg(segs_1, r2, res["branch1_aug"]),
                                                            lambda: pseudo_and_reg(segs_2, r1, res["branch2_aug"]))
            reg_terms = [g1, g2]
            out.update(pseudo_seg_1=ps1, pseudo_seg_2=ps2, n_uncertain=(n1, n2))
    c1, c2 = LS.sim_loss_terms(fmap_1, fmap_2)
    sim_terms = [(1.0, c1), (1.0, c2)]           # sim_loss = (1 + cos_1) + (1 + cos_2), train_final_voc.py:251-254
    if coco:    # hard-coded weights, train_final_coco.py:441-448
        if n_iter <= 8000:
            w = (1.0, 0.0, 0.0, 0.0)
        elif n_iter <= args.coco_switch_iter:
            w = (1.0, 0.0, 0.2, 0.05)
        else:
            w = (1.0, 0.2, 0.2, 0.05)
    elif n_iter <= args.cam_iters:
        w = (1.0, args.w_ptc, 0.0, 0.1)
    else:
        w = (1.0, args.w_ptc, args.w_seg, 0.1)
    groups = [(w[0], cls_terms), (w[1], ptc_terms), (w[2], seg_terms), (w[3], sim_terms)]
    if reg_terms is not None and (n_iter > args.gmm_iters or (coco and n_iter > args.coco_switch_iter)):
        groups.append((0.05, reg_terms))         # ... + 0.05 * reg_loss (train_final_voc.py:456, train_final_coco.py:446-448)
    loss, gs = LS.weighted_total(groups)
    cls_loss, ptc_loss, seg_loss, sim = gs[:4]
    if reg_terms is not None:
        out["reg_loss"] = gs[4] if len(gs) > 4 else reg_terms[0] + reg_terms[1]
    out.update(loss=loss, cls_loss=cls_loss, ptc_loss=ptc_loss, seg_loss=seg_loss, sim_loss=sim, cls_1=cls_1, segs_1=segs_1,
               fmap_1=fmap_1, cls_aux_1=cls_aux_1, cls_2=cls_2, segs_2=segs_2, fmap_2=fmap_2, cls_aux_2=cls_aux_2)
    return loss, out


def train_step(model, optim, par, inputs, cls_label, img_box, n_iter: int, args: StepArgs, cls_label_host=None,
               inputs_aug=None):
    """zero_grad -> losses -> backward -> optimiser step (train_final_voc.py:470-472)."""
    optim.zero_grad()
    loss, out = compute_losses(model, par, inputs, cls_label, img_box, n_iter, args, cls_label_host, inputs_aug)
    if hasattr(optim, "begin_step"):
        optim.begin_step(model)       # the update of each gradient range is issued as the backward pass (+ exchange) finalises it
    (loss if loss.numel() == 1 and loss.dim() == 0 else loss.sum()).backward()
    optim.step()
    return out
