"""Optimiser factory, threshold schedule and in-loop validation (reference: utils/train_helper.py)."""
import numpy as np
import torch

from .. import ops
from ..datasets import coco, voc
from . import cam_helper, evaluate, optimizer
from .pyutils import format_tabs


def get_optimizer(param_groups, args):
    """train_helper.py:21-53: groups 0/1 at lr, groups 2/3 at 10*lr, weight decay on every group."""
    return getattr(optimizer, args.optimizer)(
        params=[
            {"params": param_groups[0], "lr": args.lr, "weight_decay": args.wt_decay},
            {"params": param_groups[1], "lr": args.lr, "weight_decay": args.wt_decay},
            {"params": param_groups[2], "lr": args.lr * 10, "weight_decay": args.wt_decay},
            {"params": param_groups[3], "lr": args.lr * 10, "weight_decay": args.wt_decay},
        ],
        lr=args.lr, weight_decay=args.wt_decay, betas=args.betas, warmup_iter=args.warmup_iters,
        max_iter=args.max_iters, warmup_ratio=args.warmup_lr, power=args.power)


def cosine_descent(max_thres, min_thres, step, num_steps):
    """train_helper.py:340-349."""
    if step < 0:
        return max_thres
    if step >= num_steps:
        return min_thres
    interpolation_factor = step / (num_steps - 1)
    return max_thres + (min_thres - max_thres) * (1 - np.cos(np.pi * interpolation_factor)) / 2


# ------------------------------------------------------------------------------------------------------------------
# In-loop validation (reference: utils/train_helper.py:90-338).  Same signatures and return values; the per-image label
# maps never leave the device: each one is folded into a device-resident confusion matrix as soon as it exists
# (the reference keeps ~1.4k x 7 int16 host arrays and histograms them at the end), the per-image F1 scores are summed
# on the device, and the two students run on their own streams when the model has dual-stream enabled.
# ------------------------------------------------------------------------------------------------------------------
def _device_of(model):
    core = model.module if hasattr(model, "module") else model
    for p in core.parameters():
        return p.device
    raise ValueError("model without parameters")


def _fetch(data, dev):
    _, inputs, labels, cls_label = data
    inputs = inputs.to(dev, non_blocking=True).float().contiguous()
    labels = labels.to(dev, non_blocking=True).long().contiguous()
    cls_label = cls_label.to(dev, non_blocking=True).float().contiguous()
    return inputs, labels, cls_label


def _fold_student(cms, prefix_names, labels, cls_label, cls, segs, cams, cams_aux, f1_slot, args):
    """One student's share of a validation image: F1, CAM / aux-CAM label maps, seg prediction -> confusion matrices."""
    H, W = labels.shape[1:]
    ops.multilabel_f1_accum(cls, cls_label, f1_slot)
    for nm, c in zip(prefix_names[:2], (cams, cams_aux)):
        rc = ops.resize_bilinear(c, H, W)
        lab = cam_helper.cam_to_label(rc, cls_label, bkg_thre=args.bkg_thre, high_thre=args.high_thre,
                                      low_thre=args.low_thre, ignore_index=args.ignore_index)
        cms[nm].update(labels, lab)
    cms[prefix_names[2]].update(labels, ops.upsample_argmax(segs, H, W))


def _validate_siamese(model, data_loader, args, num_classes, cat_list, return_item):
    core = model.module if hasattr(model, "module") else model
    dev = _device_of(model)
    names = ["CAM_1", "aux_CAM_1", "Seg_1", "CAM_2", "aux_CAM_2", "Seg_2"]
    cms = {n: evaluate.ConfusionMatrix(num_classes, dev) for n in names}
    f1 = torch.zeros(2, device=dev, dtype=torch.float32)
    n_img = 0
    model.eval()
    with torch.no_grad():
        for data in data_loader:
            inputs, labels, cls_label = _fetch(data, dev)
            inputs = ops.resize_bilinear(inputs, int(args.crop_size), int(args.crop_size))
            # model(inputs, val=True) + multi_scale_cam2_siamese(branch=1 / 2): one fused front-end per student
            (c1, ca1), (c2, ca2), res = core.ms_cam_and_forward(inputs, args.cam_scales)
            cls_1, segs_1, _, _ = res["branch1"]
            cls_2, segs_2, _, _ = res["branch2"]
            core.per_student(
                lambda: _fold_student(cms, names[:3], labels, cls_label, cls_1, segs_1, c1, ca1, f1[0:1], args),
                lambda: _fold_student(cms, names[3:], labels, cls_label, cls_2, segs_2, c2, ca2, f1[1:2], args))
            n_img += inputs.shape[0]
    f1 = (f1 / max(n_img, 1)).cpu().tolist()
    sc = [cms[n].scores() for n in names]
    model.train()
    tab_results, score_item = format_tabs(scores=sc, name_list=names, cat_list=cat_list, return_item=True)
    if return_item:
        return f1[0], f1[1], tab_results, score_item
    return f1[0], f1[1], tab_results


def validate_siamase(model=None, data_loader=None, args=None, return_item=False):
    """train_helper.py:90-185 (VOC, 21 classes) -> (cls_score_1, cls_score_2, table[, per-column mIoU x100])."""
    return _validate_siamese(model, data_loader, args, 21, voc.class_list, return_item)


def validate_siamase_coco(model=None, data_loader=None, args=None, return_item=False):
    """train_helper.py:188-283 (COCO, 81 classes)."""
    return _validate_siamese(model, data_loader, args, 81, coco.class_list, return_item)


def validate(model=None, data_loader=None, args=None):
    """train_helper.py:286-338: single `network` -> (cls_score, table of CAM / aux_CAM / Seg_Pred IoUs)."""
    dev = _device_of(model)
    names = ["CAM", "aux_CAM", "Seg_Pred"]
    cms = {n: evaluate.ConfusionMatrix(21, dev) for n in names}
    f1 = torch.zeros(1, device=dev, dtype=torch.float32)
    n_img = 0
    model.eval()
    with torch.no_grad():
        for data in data_loader:
            inputs, labels, cls_label = _fetch(data, dev)
            inputs = ops.resize_bilinear(inputs, int(args.crop_size), int(args.crop_size))
            cls, segs, _, _ = model(inputs, val=True)
            cams, cams_aux = cam_helper.multi_scale_cam2(model, inputs, args.cam_scales)
            _fold_student(cms, names, labels, cls_label, cls, segs, cams, cams_aux, f1, args)
            n_img += inputs.shape[0]
    cls_score = float((f1 / max(n_img, 1)).item())
    model.train()
    return cls_score, format_tabs([cms[n].scores() for n in names], name_list=names, cat_list=voc.class_list)
