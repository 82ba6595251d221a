"""Golden vectors for the COCO training schedule (train_final_coco.py:190-448)  (authoring container only).
Run:  python oracle/gen_golden_coco.py

The reference's model / CAM / loss functions are composed exactly as train_final_coco.py composes them for
  n_iter =   100  (phase A: classification loss only, PTC / seg / sim weights 0, ms-CAM still evaluated),
  n_iter = 10000  (8000 < n <= 12000: refine_cams_with_bkg_v2 on the AUX CAMs, weights 1 / 0 / 0.2 / 0.05),
  n_iter = 20000  (12000 < n < gmm_iters: thresholds descending from iteration 12000, dynamic refine, 1 / 0.2 / 0.2 / 0.05)
on the tiny backbone with 81 classes; the oracle's schedule="coco" restatement is asserted equal and the reference's
outputs are written to tests/golden/tiny_step_coco_{A,B1,B2}.npz (data only)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.dont_write_bytecode = True

from oracle import dupl_oracle as O  # noqa: E402
from oracle.gen_golden import import_reference, npz, close, same  # noqa: E402


def main():
    torch.set_num_threads(8)
    R = import_reference()
    CH, LS = R["cam_helper"], R["losses"]
    cfg, NC = O.VIT_TINY, 81
    pp = O.make_siamese_params(cfg, NC, seed=4)
    sia = R["siamese"]("tiny_test", num_classes=NC, pretrained=False, aux_layer=9 % cfg.depth - cfg.depth)
    sia.load_state_dict(pp, strict=True)
    sia.train()
    args = O.coco_step_args()
    par = R["PAR"](num_iter=10, dilations=[1, 2, 4, 8, 12, 24])
    inputs, cls_label, img_box = O.synthetic_batch(2, NC - 1, 64, seed=15)
    high_thres_start = torch.ones(NC - 1) * args.high_thre
    high_thres_target = torch.tensor(args.high_target)

    def ref_step(n_iter):
        sia.zero_grad()
        inputs_denorm = O.denormalize_img2(inputs.clone())
        lab, refined = None, None
        if n_iter < args.cam_iters:
            cams_1, cams_aux_1 = CH.multi_scale_cam2_siamese(sia, inputs=inputs, scales=args.cam_scales, branch=1)
            cams_2, cams_aux_2 = CH.multi_scale_cam2_siamese(sia, inputs=inputs, scales=args.cam_scales, branch=2)
            res = sia(inputs)
            cls_1, segs_1, fmap_1, cls_aux_1 = res["branch1"]
            cls_2, segs_2, fmap_2, cls_aux_2 = res["branch2"]
            ptc_loss, seg_loss = torch.ones(1), torch.ones(1)
        else:
            high_thres = O.cosine_descent(high_thres_start, high_thres_target, n_iter - 12000, args.max_iters - 12000)
            b, _, h, w = inputs.shape
            hl, hm = [], []
            for i in range(b):
                t = torch.max(high_thres[torch.nonzero(cls_label[i]).squeeze(-1)])
                hl.append(t)
                hm.append(torch.ones((h, w)) * t)
            high_thres = torch.stack(hl, dim=0)
            high_thres_mask = torch.stack(hm, dim=0).unsqueeze(1)
            cams_1, cams_aux_1 = CH.multi_scale_cam2_siamese(sia, inputs=inputs, scales=args.cam_scales, branch=1)
            cams_2, cams_aux_2 = CH.multi_scale_cam2_siamese(sia, inputs=inputs, scales=args.cam_scales, branch=2)
            res = sia(inputs)
            cls_1, segs_1, fmap_1, cls_aux_1 = res["branch1"]
            cls_2, segs_2, fmap_2, cls_aux_2 = res["branch2"]
            lab = []
            for ca, fm in ((cams_aux_1, fmap_1), (cams_aux_2, fmap_2)):
                rc = F.interpolate(ca, size=fm.shape[2:], mode="bilinear", align_corners=False)
                _, pl = CH.cam_to_label_dynamic_cls(rc.detach(), cls_label=cls_label, img_box=img_box, ignore_mid=True,
                                                    bkg_thre=args.bkg_thre, high_thre=high_thres, low_thre=args.low_thre,
                                                    ignore_index=args.ignore_index)
                lab.append(pl)
            ptc_loss = LS.get_masked_ptc_loss(fmap_1, CH.label_to_aff_mask(lab[0])) + \
                LS.get_masked_ptc_loss(fmap_2, CH.label_to_aff_mask(lab[1]))
            rep = cls_label.unsqueeze(-1).unsqueeze(-1).repeat([1, 1, h, w])
            if n_iter <= 12000:
                r1 = CH.refine_cams_with_bkg_v2(par, inputs_denorm, cams=cams_aux_1.detach() * rep, cls_labels=cls_label,
                                                high_thre=args.high_thre, low_thre=args.low_thre,
                                                ignore_index=args.ignore_index, img_box=img_box)
                r2 = CH.refine_cams_with_bkg_v2(par, inputs_denorm, cams=cams_aux_2.detach() * rep, cls_labels=cls_label,
                                                high_thre=args.high_thre, low_thre=args.low_thre,
                                                ignore_index=args.ignore_index, img_box=img_box)
            else:
                r1 = CH.refine_cams_with_dynamic_thres(par, inputs_denorm, cams=cams_1.detach() * rep, cls_labels=cls_label,
                                                       high_thre_map=high_thres_mask, low_thre=args.low_thre,
                                                       ignore_index=args.ignore_index, img_box=img_box)
                r2 = CH.refine_cams_with_dynamic_thres(par, inputs_denorm, cams=cams_2.detach() * rep, cls_labels=cls_label,
                                                       high_thre_map=high_thres_mask, low_thre=args.low_thre,
                                                       ignore_index=args.ignore_index, img_box=img_box)
            s1 = F.interpolate(segs_1, size=r1.shape[1:], mode="bilinear", align_corners=False)
            s2 = F.interpolate(segs_2, size=r2.shape[1:], mode="bilinear", align_corners=False)
            seg_loss = LS.get_seg_loss(s1, r2.type(torch.long), ignore_index=args.ignore_index) + \
                LS.get_seg_loss(s2, r1.type(torch.long), ignore_index=args.ignore_index)
            refined = (r1, r2)
        msm = F.multilabel_soft_margin_loss
        cls_loss = msm(cls_1, cls_label) + msm(cls_aux_1, cls_label) + msm(cls_2, cls_label) + msm(cls_aux_2, cls_label)
        f1 = fmap_1.view(fmap_1.shape[0], fmap_1.shape[1], -1)
        f2 = fmap_2.view(fmap_2.shape[0], fmap_2.shape[1], -1)
        cs = nn.CosineSimilarity(dim=-1, eps=1e-6)
        sim_loss = (1 + cs(f1.detach(), f2).mean()) + (1 + cs(f2.detach(), f1).mean())
        reg_loss = torch.zeros(1)
        if n_iter <= 8000:
            loss = 1.0 * cls_loss + 0.0 * ptc_loss + 0.0 * seg_loss + 0.0 * sim_loss
        elif n_iter <= 12000:
            loss = 1.0 * cls_loss + 0.0 * ptc_loss + 0.2 * seg_loss + 0.05 * sim_loss
        else:
            loss = 1.0 * cls_loss + 0.2 * ptc_loss + 0.2 * seg_loss + 0.05 * sim_loss + 0.05 * reg_loss
        loss.sum().backward()
        grads = {k: (v.grad.clone() if v.grad is not None else None) for k, v in sia.named_parameters()}
        return dict(loss=loss.detach().sum(), cls_loss=cls_loss.detach(), ptc=ptc_loss.detach(), seg=seg_loss.detach(),
                    sim=sim_loss.detach(), lab=lab, refined=refined, cams=(cams_1, cams_aux_1, cams_2, cams_aux_2), grads=grads)

    for tag, n_iter in (("A", 100), ("B1", 10000), ("B2", 20000)):
        ref = ref_step(n_iter)
        leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] != "encoder.pos_embed") for k, v in pp.items()}
        loss, pc = O.train_step_losses(leaf, inputs, cls_label, img_box, n_iter, cfg, args)
        loss.sum().backward()
        print(f"coco {tag} (n_iter {n_iter}): loss ref {ref['loss'].item():.6f} oracle {loss.sum().item():.6f}  "
              f"ptc {ref['ptc'].sum().item():.6f} seg {ref['seg'].sum().item():.6f} sim {ref['sim'].item():.6f}")
        close(loss.sum(), ref["loss"], what="loss")
        close(pc["ptc_loss"].sum(), ref["ptc"].sum(), what="ptc")
        close(pc["seg_loss"].sum(), ref["seg"].sum(), what="seg")
        close(pc["sim_loss"], ref["sim"], what="sim")
        if ref["lab"] is not None:
            same(pc["pseudo_label_aux_1"], ref["lab"][0], "pseudo_label_aux_1")
            same(pc["pseudo_label_aux_2"], ref["lab"][1], "pseudo_label_aux_2")
            same(pc["refined_1"], ref["refined"][0], "refined_1", budget=2)
            same(pc["refined_2"], ref["refined"][1], "refined_2", budget=2)
        worst, gsave = 0.0, {}
        for k, g in ref["grads"].items():
            og = leaf[k].grad
            if g is None or g.abs().max() == 0:
                assert og is None or og.abs().max() == 0, k
                if g is not None:
                    gsave["zero." + k] = np.zeros(1, dtype=np.float32)
                continue
            worst = max(worst, (og - g).abs().max().item() / max(g.abs().max().item(), 1e-12))
            gsave["grad." + k] = g if g.numel() <= 4096 else g.reshape(-1)[::7].clone()
        print(f"      worst relative grad err over {len(gsave)} tensors: {worst:.2e}")
        assert worst < 5e-4
        extra = {}
        if ref["lab"] is not None:
            extra = dict(pseudo_label_aux_1=ref["lab"][0].to(torch.uint8), pseudo_label_aux_2=ref["lab"][1].to(torch.uint8),
                         refined_1=ref["refined"][0].to(torch.uint8), refined_2=ref["refined"][1].to(torch.uint8))
        npz(f"tiny_step_coco_{tag}", n_iter=n_iter, loss=ref["loss"], cls_loss=ref["cls_loss"], ptc_loss=ref["ptc"],
            seg_loss=ref["seg"], sim_loss=ref["sim"], cams_aux_1=ref["cams"][1][:, ::8], cams_2=ref["cams"][2][:, ::8],
            **extra, **gsave)


if __name__ == "__main__":
    main()
