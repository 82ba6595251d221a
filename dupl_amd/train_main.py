"""Shared body of train_final_voc.py / train_final_coco.py: the reference's launch surface (argparse flag names and
defaults, LOCAL_RANK env, torchrun launch, DDP wrap, PolyWarmupAdamW, PAR, per-iteration loop, periodic checkpoint
with the DDP `module.` key prefix) on the HIP engine.  Data: the reference's datasets in raw-item form + the device input
pipeline (datasets/, csrc/loader.hip) when the dataset folder exists, a DataLoader-like iterable passed to
`train(loader=...)`, or synthetic batches (no dataset is mounted in the authoring / GPU containers)."""
from __future__ import annotations

import argparse
import datetime
import logging
import os
import random
import time

import numpy as np
import torch
import torch.distributed as dist


def build_parser(dataset: str) -> argparse.ArgumentParser:
    voc = dataset == "voc"
    p = argparse.ArgumentParser()
    p.add_argument("--comment", default="_train_voc" if voc else "_train_coco", type=str, help="comment")
    p.add_argument("--num_workers", default=10 if voc else 16, type=int, help="num_workers")
    p.add_argument("--backend", default="nccl")
    p.add_argument("--seed", default=0, type=int, help="fix random seed")
    p.add_argument("--work_dir", default="work_dir_voc" if voc else "work_dir_coco_wseg", type=str)
    if voc:
        p.add_argument("--data_folder", default="/your_voc_dir", type=str, help="dataset folder")
    else:
        p.add_argument("--img_folder", default="/your_voc_dir", type=str, help="dataset folder")
        p.add_argument("--label_folder", default="/home/wyc/Dataset/MSCOCO/SegmentationClass", type=str)
    p.add_argument("--list_folder", default="datasets/voc" if voc else "datasets/coco", type=str)
    p.add_argument("--train_set", default="train_aug" if voc else "train", type=str)
    p.add_argument("--val_set", default="val" if voc else "val_part", type=str)
    p.add_argument("--scales", default=(0.5, 2), help="random rescale in training")
    p.add_argument("--backbone", default="deit_base_patch16_224", type=str, help="backbone")
    p.add_argument("--pooling", default="gmp", type=str)
    p.add_argument("--pretrained", default=False, help="path to a local ImageNet state_dict (no network here)")
    if voc:
        p.add_argument("--aux_layer", default=-3, type=int, help="aux_layer")
    p.add_argument("--samples_per_gpu", default=2 if voc else 1, type=int, help="samples_per_gpu")
    p.add_argument("--optimizer", default="PolyWarmupAdamW", type=str, help="optimizer")
    p.add_argument("--warmup_iters", default=1500, type=int)
    p.add_argument("--lr", default=6e-5, type=float)
    p.add_argument("--warmup_lr", default=1e-6, type=float)
    p.add_argument("--wt_decay", default=1e-2, type=float)
    p.add_argument("--betas", default=(0.9, 0.999))
    p.add_argument("--power", default=0.9, type=float)
    p.add_argument("--save_ckpt", default=True, type=bool)
    p.add_argument("--num_classes", default=21 if voc else 81, type=int)
    p.add_argument("--crop_size", default=448, type=int)
    p.add_argument("--ignore_index", default=255, type=int)
    p.add_argument("--max_iters", default=20000 if voc else 80000, type=int)
    p.add_argument("--log_iters", default=200, type=int)
    p.add_argument("--eval_iters", default=2000 if voc else 4000, type=int)
    p.add_argument("--cam_iters", default=2000 if voc else 8000, type=int)
    p.add_argument("--high_thre", default=0.7 if voc else 0.65, type=float)
    p.add_argument("--low_thre", default=0.25, type=float)
    p.add_argument("--bkg_thre", default=0.5 if voc else 0.45, type=float)
    p.add_argument("--cam_scales", default=(1.0, 0.5, 1.5))
    if voc:
        p.add_argument("--w_ptc", default=0.2, type=float)
        p.add_argument("--w_seg", default=0.2, type=float)
        p.add_argument("--w_seg_diff", default=2.0, type=float)
    p.add_argument("--gmm_iters", default=8000 if voc else 32000, type=int)
    p.add_argument("--gmm_valid_thre", default=1.0, type=float)
    p.add_argument("--gamma", default=0.95, type=float)
    # additions of this build
    p.add_argument("--synthetic", default="auto", type=_tristate,
                   help="auto (default): synthetic batches unless the dataset folder exists; 1 / 0 force it")
    p.add_argument("--start_iter", default=0, type=int,
                   help="first n_iter (lets a short run exercise phase B); the LR schedule starts there too")
    p.add_argument("--resume", default=None, type=str,
                   help="checkpoint directory of an earlier run: loads checkpoint.pth (reference format) and optimizer.pth "
                        "(moments, bias-correction counters, schedule position) and continues at the saved n_iter")
    p.add_argument("--stop_iter", default=None, type=int, help="leave the loop before this n_iter (time-boxed jobs + --resume)")
    p.add_argument("--single_stream", action="store_true")
    p.add_argument("--deterministic", action="store_true",
                   help="bit-reproducible steps (dupl_amd.set_deterministic): what cudnn.deterministic = True asks for in "
                        "the reference's setup_seed (train_final_voc.py:95-102); costs ~20 % throughput")
    return p


def _tristate(v: str):
    v = str(v).lower()
    if v in ("auto", ""):
        return "auto"
    if v in ("1", "true", "yes", "on"):
        return True
    if v in ("0", "false", "no", "off"):
        return False
    raise argparse.ArgumentTypeError(f"expected auto / 1 / 0, got {v!r}")


def build_loaders(args, dataset: str, device, world: int, rank: int):
    """The reference's dataset / sampler / loader construction (train_final_voc.py:120-141, train_final_coco.py:118-141)
    over this build's raw-item datasets + the device pipeline: DistributedSampler(shuffle=True), batch_size =
    samples_per_gpu, drop_last, prefetch_factor 4; val loader batch_size 1.  Returns (train_loader, val_loader)."""
    from torch.utils.data import DataLoader
    from torch.utils.data.distributed import DistributedSampler
    from .datasets import voc as voc_ds, coco as coco_ds
    from .datasets.device_loader import DeviceLoader, DeviceValLoader, raw_collate
    if dataset == "voc":
        train_dataset = voc_ds.VOC12ClsDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.train_set,
                                               stage="train", aug=True, rescale_range=args.scales, crop_size=args.crop_size,
                                               img_fliplr=True, ignore_index=args.ignore_index, num_classes=args.num_classes)
        val_dataset = voc_ds.VOC12SegDataset(root_dir=args.data_folder, name_list_dir=args.list_folder, split=args.val_set,
                                             stage="val", aug=False, ignore_index=args.ignore_index,
                                             num_classes=args.num_classes)
    else:
        train_dataset = coco_ds.CocoClsDataset(img_dir=args.img_folder, label_dir=args.label_folder,
                                               name_list_dir=args.list_folder, split=args.train_set, stage="train", aug=True,
                                               rescale_range=args.scales, crop_size=args.crop_size, img_fliplr=True,
                                               ignore_index=args.ignore_index, num_classes=args.num_classes)
        val_dataset = coco_ds.CocoSegDataset(img_dir=args.img_folder, label_dir=args.label_folder,
                                             name_list_dir=args.list_folder, split=args.val_set, stage="val", aug=False,
                                             ignore_index=args.ignore_index, num_classes=args.num_classes)
    train_sampler = DistributedSampler(train_dataset, num_replicas=world, rank=rank, shuffle=True)
    kw = dict(prefetch_factor=4) if args.num_workers > 0 else {}
    train_loader = DataLoader(train_dataset, batch_size=args.samples_per_gpu, shuffle=False, num_workers=args.num_workers,
                              pin_memory=False, drop_last=True, sampler=train_sampler, collate_fn=raw_collate, **kw)
    val_loader = DataLoader(val_dataset, batch_size=1, shuffle=False, num_workers=args.num_workers, pin_memory=False,
                            drop_last=False, collate_fn=raw_collate)
    return DeviceLoader(train_loader, device), DeviceValLoader(val_loader, device)


class _EpochIterator:
    """The reference's iterator handling (train_final_voc.py:132-133,177-182): `set_epoch(np.random.randint(max_iters))`
    on the sampler before the first pass and again whenever the loader runs dry, then a fresh iterator."""

    def __init__(self, loader, max_iters: int):
        self.loader, self.max_iters = loader, max_iters
        self.epochs = 0
        self._restart()

    def _restart(self):
        sampler = getattr(self.loader, "sampler", None)
        if sampler is not None and hasattr(sampler, "set_epoch"):
            sampler.set_epoch(np.random.randint(self.max_iters))
        self.it = iter(self.loader)
        self.epochs += 1

    def next(self):
        try:
            return next(self.it)
        except StopIteration:
            self._restart()
            try:
                return next(self.it)
            except StopIteration:
                raise RuntimeError("the training loader is empty after a restart (a one-shot generator, or fewer items "
                                   "than samples_per_gpu * world with drop_last)") from None


def setup_seed(seed):
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)
    np.random.seed(seed)
    random.seed(seed)


_RESUME_FILES = ("optimizer.n_iter", "optimizer.pth", "checkpoint.pth", "checkpoint.n_iter")


def _prev(name):
    stem, ext = os.path.splitext(name)
    return f"{stem}.prev{ext}"


def _read_generation(resume_dir, prev=False, cheap=False):
    """(optimiser payload, n_iter) of the current (or previous) generation if it is COMPLETE -- its iteration tag, written last, equals
    the n_iter inside its optimizer file, written first -- else (None, reason)."""
    begun, opt, ck, tag = (os.path.join(resume_dir, _prev(n) if prev else n) for n in _RESUME_FILES)
    if not (os.path.exists(opt) and os.path.exists(ck)):
        return None, "files missing"
    if cheap and os.path.exists(begun) and os.path.exists(tag):
        # the SAVE path only needs "is the current generation complete?" (it used to torch.load both students' moment
        # buffers for that): optimizer.n_iter is written BEFORE optimizer.pth is replaced and the tag after everything, so the two
        # agree exactly when no later save has begun to replace the files
        with open(begun) as f, open(tag) as g:
            a, b = int(f.read().strip()), int(g.read().strip())
        return (True, b) if a == b else (None, f"a save at n_iter {a} was begun after the last completed one at {b}")
    ost = torch.load(opt, map_location="cpu")
    if not os.path.exists(tag):
        return None, f"no iteration tag (a save was interrupted after its optimizer file reached n_iter {int(ost['n_iter'])})"
    with open(tag) as f:
        saved_at = int(f.read().strip())
    if saved_at != int(ost["n_iter"]):
        return None, (f"the last completed save was at n_iter {saved_at} but the optimizer file is from n_iter {int(ost['n_iter'])}: a later "
                      "save was interrupted after replacing it, the weights file is from one of the two")
    return ost, saved_at


def _save_resume_state(ckpt_dir, wrapped, model, optim, n_iter):
    """checkpoint.pth (keys prefixed `module.`, train_final_voc.py:519) + optimizer.pth + checkpoint.n_iter (+ the small
    optimizer.n_iter marker), each written to a temporary file and renamed into place, in THIS order: the marker, optimizer.pth (it
    carries n_iter), then checkpoint.pth, then the iteration tag -- a job killed between any two renames leaves a tag that differs from optimizer.pth's n_iter, which
    `_load_resume_state` recognises as an incomplete generation.  So that one complete generation ALWAYS exists (an
    interrupted save used to leave a directory that could not be resumed at all), the current generation -- if complete -- is
    first kept as *.prev (hard links: no copy; its tag linked last), and the loader falls back to it."""
    os.makedirs(ckpt_dir, exist_ok=True)
    sd = wrapped.state_dict() if wrapped is not None else {"module." + k: v for k, v in model.state_dict().items()}

    if _read_generation(ckpt_dir, cheap=True)[0] is not None:
        for name in _RESUME_FILES:                       # tag last: an interrupted rotation leaves an incomplete .prev, never a mixed one
            dst = os.path.join(ckpt_dir, _prev(name))
            if os.path.exists(dst):
                os.remove(dst)
        for name in _RESUME_FILES:
            src, dst = os.path.join(ckpt_dir, name), os.path.join(ckpt_dir, _prev(name))
            if not os.path.exists(src):                  # (a directory written before the marker existed)
                continue
            try:
                os.link(src, dst)
            except OSError:                              # a filesystem without hard links
                import shutil
                shutil.copy2(src, dst)

    def put(obj, name):
        tmp = os.path.join(ckpt_dir, name + ".tmp")
        torch.save(obj, tmp)
        os.replace(tmp, os.path.join(ckpt_dir, name))

    tmp = os.path.join(ckpt_dir, "optimizer.n_iter.tmp")
    with open(tmp, "w") as f:
        f.write(str(n_iter))
    os.replace(tmp, os.path.join(ckpt_dir, "optimizer.n_iter"))       # "a save at n_iter has begun": before any payload is replaced
    put({"n_iter": n_iter, "optimizer": optim.state_dict()}, "optimizer.pth")
    put(sd, "checkpoint.pth")
    tmp = os.path.join(ckpt_dir, "checkpoint.n_iter.tmp")
    with open(tmp, "w") as f:
        f.write(str(n_iter))
    os.replace(tmp, os.path.join(ckpt_dir, "checkpoint.n_iter"))


def _load_resume_state(resume_dir):
    """(model state_dict without `module.`, optimiser state, n_iter) of a directory written by _save_resume_state: the current
    generation if it is complete, else the previous one (*.prev, kept by every save), else an error naming what is inconsistent --
    never a possibly mixed pair of files."""
    ost, n = _read_generation(resume_dir)
    prev = False
    if ost is None:
        why = n
        ost, n = _read_generation(resume_dir, prev=True)
        if ost is None:
            raise RuntimeError(f"{resume_dir}: {why}; no complete previous generation either ({n}) -- refusing to resume from "
                               "possibly mixed state")
        prev = True
        print(f"[resume] {resume_dir}: the last save is incomplete ({why}); resuming from the previous complete one at n_iter {n}")
    ck = torch.load(os.path.join(resume_dir, _prev("checkpoint.pth") if prev else "checkpoint.pth"), map_location="cpu")
    sd = {k[len("module."):] if k.startswith("module.") else k: v for k, v in ck.items()}
    return sd, ost["optimizer"], int(ost["n_iter"])


def train(args, dataset: str, loader=None, val_loader=None):
    from .ddp import DistributedDataParallel
    from .model.model_dupl import siamese_network
    from .model.PAR import PAR
    from .synthetic import synthetic_batch
    from .utils import train_helper
    from . import trainer

    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    if getattr(args, "deterministic", False):
        import dupl_amd
        dupl_amd.set_deterministic(True)
    distributed = int(os.environ.get("WORLD_SIZE", "1")) > 1
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=args.backend)
    device = torch.device("cuda", local_rank)
    rank = dist.get_rank() if distributed else 0
    voc = dataset == "voc"
    aux_layer = args.aux_layer if voc else 9            # train_final_coco.py:148
    model = siamese_network(backbone=args.backbone, num_classes=args.num_classes, pretrained=args.pretrained,
                            aux_layer=aux_layer)
    param_groups = model.get_param_groups()
    model.to(device)
    if not args.single_stream:
        model.enable_dual_stream(True)
    wrapped = DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True) if distributed else model
    optim = train_helper.get_optimizer(param_groups, args).bind(model.flat_storage)
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(device)
    C = args.num_classes - 1
    if voc:
        sargs = trainer.StepArgs(cam_iters=args.cam_iters, gmm_iters=args.gmm_iters, max_iters=args.max_iters,
                                 bkg_thre=args.bkg_thre, high_thre=args.high_thre, low_thre=args.low_thre,
                                 ignore_index=args.ignore_index, w_ptc=args.w_ptc, w_seg=args.w_seg,
                                 cam_scales=tuple(args.cam_scales), samples_per_gpu=args.samples_per_gpu)
    else:
        sargs = trainer.coco_step_args(cam_iters=args.cam_iters, gmm_iters=args.gmm_iters, max_iters=args.max_iters,
                                       bkg_thre=args.bkg_thre, high_thre=args.high_thre, low_thre=args.low_thre,
                                       ignore_index=args.ignore_index, cam_scales=tuple(args.cam_scales),
                                       samples_per_gpu=args.samples_per_gpu)
    world = dist.get_world_size() if distributed else 1
    synthetic = getattr(args, "synthetic", "auto")
    if loader is None and synthetic is not True:
        folder = args.data_folder if voc else args.img_folder
        if synthetic is False or os.path.isdir(folder):
            loader, built_val = build_loaders(args, dataset, device, world, rank)
            val_loader = val_loader if val_loader is not None else built_val
    it = _EpochIterator(loader, args.max_iters) if loader is not None else None
    optim.global_step = args.start_iter       # a run that starts at n_iter = k is at step k of the LR schedule too
    if getattr(args, "resume", None):
        sd, opt_state, at = _load_resume_state(args.resume)
        model.load_state_dict(sd, strict=True)
        optim.load_state_dict(opt_state)
        args.start_iter = at
        if rank == 0:
            logging.info("resumed from %s at n_iter %d" % (args.resume, args.start_iter))
    last_iter = args.max_iters if getattr(args, "stop_iter", None) is None else min(args.max_iters, args.stop_iter)
    t0 = time.time()
    acc = {}
    for n_iter in range(args.start_iter, last_iter):
        if it is not None:
            _, inputs, cls_label, img_box, _ = it.next()
            cls_label = cls_label.float()
            cls_host = cls_label
            inputs, cls_label = inputs.to(device), cls_label.to(device)
        else:
            inputs, cls_label, img_box = synthetic_batch(args.samples_per_gpu, C, args.crop_size, seed=n_iter * 64 + rank)
            cls_host = cls_label
            inputs, cls_label = inputs.to(device), cls_label.to(device)
        # phase C's strongly augmented view (train_final_voc.py:191) is computed inside the step, on the device
        out = trainer.train_step(wrapped, optim, par, inputs, cls_label, img_box, n_iter, sargs, cls_label_host=cls_host)
        for k in ("cls_loss", "ptc_loss", "seg_loss", "sim_loss"):
            acc[k] = acc.get(k, 0.0) + out[k].detach().reshape(-1)[0]     # device-side accumulation, no host sync
        if (n_iter + 1) % args.log_iters == 0 and rank == 0:
            n = args.log_iters
            el = time.time() - t0
            logging.info("Iter: %d; Elapsed: %.0fs; LR: %.3e; cls_loss: %.4f | ptc_loss: %.4f | seg_loss: %.4f | sim_loss: %.4f"
                         % (n_iter + 1, el, optim.param_groups[0]["lr"], float(acc["cls_loss"]) / n, float(acc["ptc_loss"]) / n,
                            float(acc["seg_loss"]) / n, float(acc["sim_loss"]) / n))
            acc = {}
        leaving = n_iter + 1 == last_iter and last_iter < args.max_iters     # --stop_iter: keep what the job has done
        if ((n_iter + 1) % args.eval_iters == 0 or leaving) and rank == 0 and args.save_ckpt:
            _save_resume_state(args.ckpt_dir, wrapped if distributed else None, model, optim, n_iter + 1)
        if (n_iter + 1) % args.eval_iters == 0 and rank == 0:
            # in-loop validation on rank 0 (train_final_voc.py:521-533); with no dataset mounted a few synthetic
            # native-size samples stand in for the val split so that the path is exercised end to end
            vl = val_loader
            if vl is None:
                from .synthetic_val import synthetic_val_samples
                vl = [((f"synthetic{i}",), x, lab, cls) for i, (x, lab, cls) in enumerate(
                    synthetic_val_samples(sizes=((375, 500), (333, 500), (500, 375)), num_fg=C, seed=n_iter))]
            validate = train_helper.validate_siamase if voc else train_helper.validate_siamase_coco
            tv = time.time()
            s1, s2, tab, items = validate(model=wrapped, data_loader=vl, args=args, return_item=True)
            logging.info("val cls score: %.6f (branch1) %.6f (branch2); %d images in %.2fs" % (s1, s2, len(vl), time.time() - tv))
            logging.info("\n" + tab)
    torch.cuda.synchronize()
    if distributed:
        dist.destroy_process_group()
    return True


def main(dataset: str):
    args = build_parser(dataset).parse_args()
    timestamp = "{0:%Y-%m-%d-%H-%M-%S-%f}".format(datetime.datetime.now()) + args.comment
    args.work_dir = os.path.join(args.work_dir, timestamp)
    args.ckpt_dir = os.path.join(args.work_dir, "checkpoints")
    args.pred_dir = os.path.join(args.work_dir, "predictions")
    logging.basicConfig(level=logging.INFO, format="%(asctime)s %(message)s")
    setup_seed(args.seed)
    train(args, dataset)
