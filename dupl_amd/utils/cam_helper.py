"""CAM / pseudo-label helpers -- the reference's utils/cam_helper.py (superset of utils/camutils.py) API on
the HIP kernels: multi_scale_cam2[_siamese], cam_to_label[_dynamic_cls], label_to_aff_mask,
refine_cams_with_bkg_v2, refine_cams_with_dynamic_thres.  Signatures, return order and dtypes follow the
reference (cam_helper.py:8-55,164-204,323-440); inputs must live on the GPU.
"""
from typing import Optional, Sequence

import torch

from .. import engine, ops


def _unwrap(model):
    return model.module if hasattr(model, "module") and not hasattr(model, "_P") and not hasattr(model, "branch1") else model


def _student(model, branch):
    """Resolve `model` (network | siamese_network | a DDP-style wrapper) + branch to the student engine handle."""
    m = _unwrap(model)
    if hasattr(m, "branch1"):
        return (m.branch1 if branch in (None, 1) else m.branch2)._P
    return m._P


def _ms_cam(P, inputs, scales, share=None):
    """cam_helper.py:164-204 fused (`share`: optional dict that receives the encoder state of the un-flipped
    scale-1.0 pass under key "enc" for reuse by the training forward of the same step): every scale's [x ; flip(x)]
    batch goes through the encoder (all no-grad scales in ONE merged pass, engine.cam_logits_multi) and one skinny GEMM
    per classifier; all scales are then up-sampled / flip-max'ed / ReLU'ed / summed and min-max normalised by two
    streaming kernels (the low-resolution logits stay token-major)."""
    inputs = inputs.contiguous().float()
    b, _, h, w = inputs.shape
    C = P.num_classes - 1
    patch = P.cfg.patch
    order = [1.0] + [s for s in scales if s != 1.0]   # 1.0 first, then tuple order (cam_helper.py:169-196)
    with torch.no_grad():
        dims = [(h, w) if s == 1.0 else (int(s * h), int(s * w)) for s in order]
        xs = [ops.resize_bilinear(inputs, hs, ws, flip_cat=True) for hs, ws in dims]
        sizes = [(hs // patch, ws // patch) for hs, ws in dims]
        if share is not None:
            # scale 1.0 runs WITH activation saving and doubles as the training forward; the remaining scales share
            # one merged no-grad pass (engine.cam_logits_multi)
            share["x"] = xs[0][:b]
            rows_all = sum(x_.shape[0] * ((x_.shape[2] // patch) * (x_.shape[3] // patch) + 1) for x_ in xs)
            if engine.GEMM_MODE == "f16x3":
                # the range guard's verdicts are taken in where the operand planes are brought up to date: do that BEFORE the route is
                # chosen from them (a site that turns f32-routed at this step -- a harvest, a rewritten parameter -- would otherwise
                # meet a merged pass that can no longer save a row prefix)
                P.store.ensure_w16(P.student)
            if len(xs) > 1 and rows_all <= engine.MERGED_PASS and engine.partial_save_ok(P):
                # round 5: every scale AND the training forward in one encoder pass (21 976 token rows at 448^2, 4 images)
                res, share["enc"] = engine.cam_logits_shared_multi(P, xs, b)
            else:
                cam_aux_t, cam_t, share["enc"] = engine.cam_logits_shared(P, xs[0], b)
                res = [(cam_aux_t, cam_t)] + (engine.cam_logits_multi(P, xs[1:]) if len(xs) > 1 else [])
        else:
            res = engine.cam_logits_multi(P, xs)
        lows_aux = [r[0] for r in res]
        lows = [r[1] for r in res]
        cam, mm = ops.cam_fuse(lows, sizes, b, C, h, w, row_off=1, ldc=C)
        ops.cam_normalise_(cam, mm)
        cam_aux, mm2 = ops.cam_fuse(lows_aux, sizes, b, C, h, w, row_off=1, ldc=C)
        ops.cam_normalise_(cam_aux, mm2)
    return cam, cam_aux


def multi_scale_cam2(model, inputs, scales):
    """camutils.py:87-127: single-network variant -> (cam, cam_aux), each (b,C,H,W) in [0,1)."""
    return _ms_cam(_student(model, None), inputs, scales)


def multi_scale_cam2_siamese(model, inputs, scales, branch=1):
    """cam_helper.py:164-204 -> (cam, cam_aux) of student `branch`."""
    return _ms_cam(_student(model, branch), inputs, scales)


def _box_i32(img_box, device):
    if img_box is None:
        return None
    t = img_box if torch.is_tensor(img_box) else torch.as_tensor(img_box)
    if t.is_cuda:
        return t.to(dtype=torch.int32).contiguous()
    return ops.to_device_async(t.numpy(), torch.int32, device)


def _cam_to_label(cam, cls_label, img_box, bkg_thre, high_thre, low_thre, ignore_mid, ignore_index):
    cam = cam.contiguous().float()
    b = cam.shape[0]
    dev = cam.device
    box = _box_i32(img_box, dev)
    high = None
    if box is not None and ignore_mid:
        if torch.is_tensor(high_thre):
            if high_thre.is_cuda:
                high = high_thre.to(dtype=torch.float32).reshape(-1).contiguous()
            else:
                high = ops.to_device_async(high_thre.reshape(-1).numpy(), torch.float32, dev)
            if high.numel() == 1:
                high = high.repeat(b)
        else:
            high = torch.full((b,), float(high_thre), device=dev, dtype=torch.float32)
    valid, label = ops.cam_to_label(cam, cls_label.to(dev).contiguous().float(), box, high, bkg_thre,
                                    low_thre if low_thre is not None else 0.0, bool(ignore_mid),
                                    ignore_index if ignore_index is not None else 0, want_valid=box is not None)
    if box is None:
        return label
    return valid, label


def cam_to_label(cam, cls_label, img_box=None, bkg_thre=None, high_thre=None, low_thre=None, ignore_mid=False,
                 ignore_index=None):
    """cam_helper.py:8-30.  Returns label only when img_box is None, else (valid_cam, pseudo_label int64)."""
    return _cam_to_label(cam, cls_label, img_box, bkg_thre, high_thre, low_thre, ignore_mid, ignore_index)


def cam_to_label_dynamic_cls(cam, cls_label, img_box=None, bkg_thre=None, high_thre=None, low_thre=None, ignore_mid=False,
                             ignore_index=None):
    """cam_helper.py:33-55: `high_thre` is a (b,) tensor of per-image thresholds."""
    return _cam_to_label(cam, cls_label, img_box, bkg_thre, high_thre, low_thre, ignore_mid, ignore_index)


def label_to_aff_mask(cam_label, ignore_index=255):
    """cam_helper.py:323-335 -> (b,hw,hw) int64 in {0,1,ignore}.  API-parity helper built from broadcasting
    comparisons; the training step never materialises it (losses.get_masked_ptc_loss_from_label)."""
    b, h, w = cam_label.shape
    l = cam_label.reshape(b, -1)
    aff = (l[:, :, None] == l[:, None, :]).long()
    ign = l == ignore_index
    aff.masked_fill_(ign[:, :, None] | ign[:, None, :], ignore_index)
    idx = torch.arange(h * w, device=cam_label.device)
    aff[:, idx, idx] = ignore_index
    return aff


def par_affinity_of(ref_mod, images, down_scale=2):
    """The colour affinity _refine builds from `images` (PAR.py:39-85 on the down-scaled image): depends on the images only, so a
    caller that refines several CAM sets of the SAME images -- the two students of a step -- builds it once and passes it as `aff`."""
    images = images.contiguous().float()
    H, W = images.shape[2:]
    return ref_mod.affinity(ops.resize_bilinear(images, H // int(down_scale), W // int(down_scale)))


def _refine(ref_mod, images, cams, cls_labels, thr_scalar: Optional[float], thr_map, low_thre, ignore_index, img_box,
            down_scale, aff=None):
    """Shared body of refine_cams_with_bkg_v2 / refine_cams_with_dynamic_thres (cam_helper.py:338-431).
    Jobs: for every image, one PAR run with the high threshold and one with the low threshold; the colour affinity
    is built once per image and shared (the reference rebuilds it per run)."""
    down_scale = int(down_scale)
    assert down_scale >= 1
    images = images.contiguous().float()
    cams = cams.contiguous().float()
    dev = cams.device
    b, C, H, W = cams.shape
    cl = cls_labels.detach().to("cpu")   # pass a HOST copy of the labels to avoid a stream sync here
    keys_l, K_l = [], []
    for i in range(b):
        ks = [0] + [int(c) + 1 for c in torch.nonzero(cl[i])[:, 0].tolist()]
        keys_l.append(ks)
        K_l.append(len(ks))
    Kmax = max(K_l)
    njobs = 2 * b   # [high jobs of images 0..b-1][low jobs of images 0..b-1]
    import numpy as np
    keys_h = np.zeros((njobs, Kmax), dtype=np.int32)
    for j in range(njobs):
        ks = keys_l[j % b]
        keys_h[j, :len(ks)] = ks
    # one packed host->device transfer (pinned, non-blocking) for all job tables
    tab = np.concatenate([np.array([j % b for j in range(njobs)], np.int32),
                          np.array([K_l[j % b] for j in range(njobs)], np.int32), keys_h.reshape(-1)])
    tab_d = ops.to_device_async(tab, torch.int32, dev)
    job_img, job_K, keys = tab_d[:njobs], tab_d[njobs:2 * njobs], tab_d[2 * njobs:].view(njobs, Kmax)
    box = _box_i32(img_box, dev)
    if aff is None:
        aff = ref_mod.affinity(ops.resize_bilinear(images, H // down_scale, W // down_scale))
    thr_low = torch.full((b,), float(low_thre), device=dev, dtype=torch.float32)
    if thr_map is not None:
        m_h = ops.refine_pre(cams, thr_map.to(dev).contiguous().float(), None, job_img[:b], job_K[:b], keys[:b], down_scale)
    else:
        thr_hi = torch.full((b,), float(thr_scalar), device=dev, dtype=torch.float32)
        m_h = ops.refine_pre(cams, None, thr_hi, job_img[:b], job_K[:b], keys[:b], down_scale)
    m_l = ops.refine_pre(cams, None, thr_low, job_img[b:], job_K[b:], keys[b:], down_scale)
    masks = torch.cat([m_h, m_l], dim=0)
    masks = ref_mod.propagate(aff, masks, job_img, job_K)
    lab = ops.refine_post(masks, job_img, job_K, keys, box, float(ignore_index), out_size=(H, W))
    return ops.refine_merge(lab[:b].contiguous(), lab[b:].contiguous(), float(ignore_index))


def refine_cams_with_bkg_v2(ref_mod=None, images=None, cams=None, cls_labels=None, high_thre=None, low_thre=None,
                            ignore_index=False, img_box=None, down_scale=2, aff=None):
    """cam_helper.py:338-383 (= camutils.py:145-185) -> (b,H,W) float32 labels in {0..C, ignore_index}.
    aff (extension): par_affinity_of(ref_mod, images, down_scale), when the caller already has it."""
    return _refine(ref_mod, images, cams, cls_labels, high_thre, None, low_thre, ignore_index, img_box, down_scale, aff=aff)


def refine_cams_with_dynamic_thres(ref_mod=None, images=None, cams=None, cls_labels=None, high_thre_map=None,
                                   low_thre=None, ignore_index=False, img_box=None, down_scale=2, aff=None):
    """cam_helper.py:386-431: high threshold given as a (b,1,H,W) map.  aff: as in refine_cams_with_bkg_v2."""
    return _refine(ref_mod, images, cams, cls_labels, None, high_thre_map, low_thre, ignore_index, img_box, down_scale, aff=aff)
