"""Offline multi-scale + flip segmentation inference (reference: tools/eval_seg_voc.py:38-91,154-193 and
tools/eval_seg_coco_ddp.py:54-139), pre-CRF.

    python -m dupl_amd.tools.eval_seg --model_path work_dir/checkpoints/checkpoint.pth ...   (needs a data loader)

`msc_seg_logits` is the per-image core: for every scale the two students run on [x_s; flip(x_s)], the low-resolution
logits are up-sampled to the label size, the flipped half is flipped back and added, and the scales are combined with
an element-wise max -- fused into one accumulate kernel per scale (csrc/eval.hip::msc_seg_accum_kernel), so the
(3, 2, C, H, W) stack the reference builds is never materialised.  `load_checkpoint` reads the reference's checkpoint
format (torch.save(model.state_dict()) of the DDP-wrapped model: keys prefixed `module.`, train_final_voc.py:514-519).
DenseCRF post-processing (utils/dcrf.py) is a CPU third-party step and stays outside this package."""
from collections import OrderedDict

import torch

from .. import ops
from ..utils import evaluate
from ..utils.pyutils import format_tabs


def load_checkpoint(model, path_or_state):
    """tools/eval_seg_voc.py:173-178: strip the DDP `module.` prefix and load strictly."""
    sd = torch.load(path_or_state, map_location="cpu") if isinstance(path_or_state, str) else path_or_state
    new = OrderedDict((k.replace("module.", ""), v) for k, v in sd.items())
    model.load_state_dict(new, strict=True)
    return model


def msc_seg_logits(model, inputs, out_size, scales=(1.0, 1.5, 1.25)):
    """inputs (1,3,h,w) on the device -> (seg_1, seg_2), each (1,C1,H,W): eval_seg_voc.py:52-75."""
    assert inputs.shape[0] == 1
    core = model.module if hasattr(model, "module") else model
    _, _, h, w = inputs.shape
    H, W = int(out_size[0]), int(out_size[1])
    accs = None
    with torch.no_grad():
        for i, sc in enumerate(scales):
            # F.interpolate(inputs, [int(h*sc), int(w*sc)]) and cat([x, x.flip(-1)]) in one kernel (flip_cat)
            cat = ops.resize_bilinear(inputs.contiguous().float(), int(h * sc), int(w * sc), flip_cat=True)
            res = core(cat)
            segs = (res["branch1"][1], res["branch2"][1])
            if accs is None:
                C1 = segs[0].shape[1]
                accs = [torch.empty((1, C1, H, W), device=inputs.device, dtype=torch.float32) for _ in range(2)]
            for acc, s in zip(accs, segs):
                ops.msc_seg_accum_(acc, s, mode=(0 if i == 0 else 1))
    return accs[0], accs[1]


def msc_seg_logits_coco(model, inputs, scales=(1.0, 1.25, 1.5), size=448):
    """tools/eval_seg_coco_ddp.py:76-119: the image is first resized to size x size; per scale the logits of
    [x_s; flip(x_s)] are resized to the scale-1 logit size, the flipped half is flipped back and added, and the scales
    are SUMMED (VOC takes the max at label size).  -> (seg_1, seg_2), each (1,C1,size/16,size/16); the caller up-samples
    the sum to the label size (ops.upsample_argmax)."""
    assert inputs.shape[0] == 1
    core = model.module if hasattr(model, "module") else model
    x = ops.resize_bilinear(inputs.contiguous().float(), size, size)
    accs = None
    order = [1.0] + [float(s) for s in scales if float(s) != 1.0]      # the reference always starts with scale 1
    with torch.no_grad():
        for i, sc in enumerate(order):
            cat = ops.resize_bilinear(x, int(size * sc), int(size * sc), flip_cat=True)
            res = core(cat)
            segs = (res["branch1"][1], res["branch2"][1])
            if accs is None:
                C1, hs, ws = segs[0].shape[1:]
                accs = [torch.empty((1, C1, hs, ws), device=x.device, dtype=torch.float32) for _ in range(2)]
            for acc, s in zip(accs, segs):
                ops.msc_seg_accum_(acc, s, mode=(0 if i == 0 else 2))
    return accs[0], accs[1]


def validate_coco(model, data_loader, args, num_classes=81, cat_list=None, process_group=None, keep_logits=None):
    """eval_seg_coco_ddp._validate on this rank's shard of the loader (the reference splits the val set round-robin over
    the ranks, tools/eval_seg_coco_ddp.py:239-245, and every rank scores its own shard).  With `process_group` (or an
    initialised default group) the per-rank confusion matrices are additionally summed over the ranks -- nc^2 int64
    counters, the only exchange of the evaluation path -- so that every rank returns the scores of the WHOLE set."""
    import torch.distributed as dist
    from ..utils.train_helper import _device_of, _fetch
    dev = _device_of(model)
    cms = [evaluate.ConfusionMatrix(num_classes, dev) for _ in range(2)]
    model.eval()
    scales = getattr(args, "scales", (1.0, 1.25, 1.5))
    for data in data_loader:
        inputs, labels, _ = _fetch(data, dev)
        seg = msc_seg_logits_coco(model, inputs, scales, getattr(args, "crop_size", 448))
        H, W = labels.shape[1:]
        for k in range(2):
            cms[k].update(labels, ops.upsample_argmax(seg[k], H, W))
            if keep_logits is not None:          # eval_seg_coco_ddp.py:127-128 saves the summed low-resolution logits
                keep_logits(data[0], k + 1, seg[k])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(process_group) > 1:
        for c in cms:
            dist.all_reduce(c.hist, op=dist.ReduceOp.SUM, group=process_group)
    sc = [c.scores() for c in cms]
    if cat_list is not None:
        print(format_tabs(sc, ["Seg_1", "Seg_2"], cat_list=cat_list))
    return sc[0], sc[1]


def validate(model, data_loader, args, num_classes=21, cat_list=None, keep_logits=None):
    """eval_seg_voc._validate: multi-scale seg predictions of both students over a loader of
    (name, inputs (1,3,h,w), labels (1,H,W), cls_label) -> (seg_score_1, seg_score_2).
    keep_logits(name, branch, logits) is called with the (1,C1,H,W) device logits (the reference np.save()s them for the
    CRF stage)."""
    from ..utils.train_helper import _device_of, _fetch
    dev = _device_of(model)
    cms = [evaluate.ConfusionMatrix(num_classes, dev) for _ in range(2)]
    model.eval()
    for data in data_loader:
        inputs, labels, _ = _fetch(data, dev)
        seg = msc_seg_logits(model, inputs, labels.shape[1:], getattr(args, "scales", (1.0, 1.5, 1.25)))
        for k in range(2):
            cms[k].update(labels, ops.argmax_channels(seg[k]))
            if keep_logits is not None:
                keep_logits(data[0], k + 1, seg[k])
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        # main() shards the split over the ranks for both datasets: the score is of the summed confusion matrices
        for c in cms:
            dist.all_reduce(c.hist, op=dist.ReduceOp.SUM)
    sc = [c.scores() for c in cms]
    if cat_list is not None and (not dist.is_initialized() or dist.get_rank() == 0):
        print(format_tabs(sc, ["Seg_1", "Seg_2"], cat_list=cat_list))
    return sc[0], sc[1]


def build_parser(dataset: str = "voc"):
    """The flags of tools/eval_seg_voc.py:26-36 (dataset "voc") or tools/eval_seg_coco_ddp.py:31-46 ("coco"): same names and
    defaults (tests/golden/cli_flags.json), plus --dataset and --save_logits."""
    import argparse
    voc_ = dataset == "voc"
    p = argparse.ArgumentParser()
    p.add_argument("--dataset", default=dataset, choices=("voc", "coco"))
    p.add_argument("--infer_set", default="val" if voc_ else "val_part", type=str)
    p.add_argument("--pooling", default="gmp", type=str)
    p.add_argument("--model_path", default="your_model_dir/checkpoints.pth", type=str)
    p.add_argument("--backbone", default="deit_base_patch16_224" if voc_ else "vit_base_patch16_224", type=str)
    if voc_:
        p.add_argument("--data_folder", default="your_voc_dir", type=str, help="dataset folder")
    else:
        p.add_argument("--img_folder", default="your_coco_dir", type=str, help="image folder")
        p.add_argument("--label_folder", default="your_coco_seg_dir", type=str, help="label folder")
        p.add_argument("--backend", default="nccl")
        p.add_argument("--crop_size", default=448, type=int)
    p.add_argument("--list_folder", default="datasets/voc" if voc_ else "datasets/coco", type=str)
    p.add_argument("--num_classes", default=21 if voc_ else 81, type=int)
    p.add_argument("--ignore_index", default=255, type=int)
    p.add_argument("--scales", default=[1.0, 1.5, 1.25] if voc_ else [1.0, 1.25, 1.5], help="multi-scale list")
    p.add_argument("--save_logits", default=1, type=int,
                   help="write <run>/segs/logits/<infer_set>/branch{1,2}/<name>.npy = {'msc_seg': ...} like the reference does for "
                        "its DenseCRF stage (utils/dcrf.py, CPU, outside this package)")
    return p


def main(argv=None):
    """tools/eval_seg_voc.py:154-193 (`validate`) / tools/eval_seg_coco_ddp.py:142-248 up to the CRF stage: build the val
    loader, load the reference-format checkpoint strictly, run the multi-scale + flip inference of both students, print the
    score tables; under torch.distributed.run the COCO split is sharded round-robin over the ranks and the confusion
    matrices are summed."""
    import os
    import numpy as np
    import torch.distributed as dist
    from torch.utils.data import DataLoader, Subset
    from ..datasets import voc, coco
    from ..datasets.device_loader import DeviceValLoader, raw_collate
    from ..model.model_dupl import siamese_network
    import argparse
    pre = argparse.ArgumentParser(add_help=False)
    pre.add_argument("--dataset", default="voc", choices=("voc", "coco"))
    is_voc = pre.parse_known_args(argv)[0].dataset == "voc"
    args = build_parser("voc" if is_voc else "coco").parse_args(argv)
    if isinstance(args.scales, str):
        args.scales = [float(v) for v in args.scales.strip("()[] ").split(",")]
    args.scales = tuple(args.scales)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(getattr(args, "backend", "nccl"))
    if is_voc:
        ds = voc.VOC12SegDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.infer_set, stage="val",
                                 aug=False, ignore_index=args.ignore_index, num_classes=args.num_classes)
    else:
        ds = coco.CocoSegDataset(img_dir=args.img_folder, label_dir=args.label_folder, name_list_dir=args.list_folder,
                                 split=args.infer_set, stage="val", aug=False, ignore_index=args.ignore_index)
    if world > 1:
        ds = Subset(ds, list(range(dist.get_rank(), len(ds), world)))        # eval_seg_coco_ddp.py:239-245
    loader = DeviceValLoader(DataLoader(ds, batch_size=1, shuffle=False, num_workers=8, pin_memory=False, drop_last=False,
                                        collate_fn=raw_collate), dev)
    model = siamese_network(backbone=args.backbone, num_classes=args.num_classes, pretrained=False, aux_layer=-3)
    load_checkpoint(model, args.model_path)
    model.to(dev)
    model.eval()
    base_dir = args.model_path.split("checkpoints")[0]
    logits_dir = os.path.join(base_dir, "segs/logits", args.infer_set)
    keep = None
    if args.save_logits:
        for b in ("branch1", "branch2"):
            os.makedirs(os.path.join(logits_dir, b), exist_ok=True)

        def keep(name, branch, logits):
            nm = name[0] if isinstance(name, (tuple, list)) else name
            np.save(os.path.join(logits_dir, f"branch{branch}", str(nm) + ".npy"), {"msc_seg": logits.cpu().numpy()})

    with torch.no_grad():
        if is_voc:
            s1, s2 = validate(model, loader, args, args.num_classes, voc.class_list, keep_logits=keep)
        else:
            s1, s2 = validate_coco(model, loader, args, args.num_classes, coco.class_list, keep_logits=keep)
    if int(os.environ.get("RANK", "0")) == 0:
        print({"Seg_1 mIoU": s1["miou"], "Seg_2 mIoU": s2["miou"],
               "next": f"DenseCRF over {logits_dir}/branch{1 if s1['miou'] > s2['miou'] else 2} (reference: crf_proc, CPU)"})
    return s1, s2


if __name__ == "__main__":
    main()
