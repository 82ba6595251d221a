#!/usr/bin/env python
"""bench.py -- DuPL training-step throughput on MI355X (BASELINE.json metric: training img/s at 448^2).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched through torch.distributed.run)

A "step" is one full phase-B iteration of the reference loop (train_final_voc.py:174-472) on a synthetic
batch that is already resident in HBM: ms-CAM for both students (3 scales x flip), dual-student
forward/backward, CAM->label, PTC, PAR refinement (high+low) for both students, cross seg loss,
discrepancy loss, gradient all-reduce (N > 1) and the PolyWarmupAdamW update.  Nothing is skipped or cached
across steps.  Workload = BASELINE.json configs[1]: VOC 448^2, dual-student ViT-B/16, 4 images per GPU
(weak scaling: per-GPU batch fixed).  Rank 0 prints ONE JSON line.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_IMG_PHASE_AB = 3.44e12       # BASELINE.md section 2 (algorithmic, both students, fwd+bwd = 3x fwd)
FLOP_SHARED_PASS = 2 * 0.1570e12      # the scale-1.0 un-flipped ms-CAM encoder pass == the training forward's encoder pass
                                      # (same weights, same input): executed once per student when --share-encoder (default)
FLOP_PHASE_C_AUG = 6 * (0.0828e12 + 0.0052e12)   # phase C: fwd+bwd of both students on the 336^2 strong-aug batch
FLOP_PHASE_C_DEAD = 4 * (0.1570e12 + 0.0093e12)  # phase C: the reference's discarded 2b forward (never executed here)
PEAK_F32_MFMA = 157.3e12              # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (f32 in / f32 acc)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=4, help="images per GPU (VOC config: 4)")
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--dataset", default="voc", choices=["voc", "coco"])
    ap.add_argument("--backbone", default="deit_base_patch16_224")
    ap.add_argument("--n-iter", type=int, default=5000, help="iteration index the step pretends to be (5000 = phase B)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "skip"])
    ap.add_argument("--cpu-size", type=int, default=448)
    ap.add_argument("--cpu-threads", type=int, default=32, help="host threads for the CPU baseline (0 = all logical CPUs)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-share-encoder", action="store_true",
                    help="run the training forward's encoder pass separately from ms-CAM's identical scale-1.0 pass, exactly "
                         "like the reference does (default: computed once and shared; outputs are bit-identical)")
    ap.add_argument("--single-stream", action="store_true", help="run the two students back to back on one stream")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only exists so that the "
                         "multi-rank control flow can be exercised by two ranks sharing one GPU in the tests)")
    return ap.parse_args()


def build_world(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("DUPL_BENCH_RANKS_SHARE_GPU0") == "1":    # test hook: every rank on device 0 (needs --backend gloo)
            local = 0
        torch.cuda.set_device(local)
        dist.init_process_group(backend=args.backend)
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def make_batch(args, rank, dev, C):
    from dupl_amd.synthetic import synthetic_batch
    inputs, cls_label, img_box = synthetic_batch(args.batch, C, args.size, seed=100 + rank)
    return inputs.to(dev), cls_label.to(dev), img_box, cls_label


def cpu_baseline(args, C):
    """The oracle (torch-CPU restatement of the reference, kind 'port') timed on this box's host cores on a BOUNDED
    sample: one phase-B step at b=1 (same 448^2 dual-student workload, 1/`batch` of a GPU step)."""
    from oracle import dupl_oracle as O
    cores = min(os.cpu_count() or 1, args.cpu_threads) if args.cpu_threads > 0 else (os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = O.VIT_BASE
    NC = C + 1
    pp = O.make_siamese_params(cfg, NC, seed=3, randomize_affine=False)
    leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] not in ("encoder.pos_embed",)) for k, v in pp.items()}
    inputs, cls_label, img_box = O.synthetic_batch(1, C, args.cpu_size, seed=100)
    sargs = O.StepArgs() if args.dataset == "voc" else O.StepArgs(cam_iters=8000, gmm_iters=32000, max_iters=80000,
                                                                   bkg_thre=0.45, high_thre=0.65,
                                                                   high_target=tuple([0.55] * 80))
    t0 = time.perf_counter()
    loss, _ = O.train_step_losses(leaf, inputs, cls_label, img_box, 5000, cfg, sargs)   # always the phase-B headline step
    loss.backward()
    mom = {}
    for k, p in leaf.items():
        if p.grad is None:
            continue
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        with torch.no_grad():
            O.adamw_update(p, p.grad, m, v, 1, 6e-5 if O.param_group_index(k) < 2 else 6e-4)
    dt = time.perf_counter() - t0
    return {"value": round(1.0 / dt, 5), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"1 phase-B step of oracle/dupl_oracle.py (dual ViT-B/16, {args.cpu_size}^2, b=1, fp32, "
                      f"torch {torch.__version__} CPU, {cores} threads): {dt:.1f} s"}


def pmc_traffic_per_launch(kernel_prefix="gemm_f32_kernel<false, false"):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC passes (profiles/r01_final_pmc_hbm.txt:
    separate rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same workload, summarised by
    tools/rocpd_pmc.py).  FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md 'HBM');
    it counts L2->fabric requests, i.e. Infinity-Cache hits are included.  None if the file is absent."""
    path = os.path.join(ROOT, "profiles", "r01_final_pmc_hbm.txt")
    try:
        calls = fetch_kib = write_kib = 0.0
        for line in open(path):
            if line.startswith(kernel_prefix):      # both column-tile instantiations (<..., 64, 1, 2> and <..., 64, 1, 1>)
                parts = line.split()
                calls += float(parts[-4])
                fetch_kib += float(parts[-2])
                write_kib += float(parts[-1])
        if calls:
            return round((2.0 * fetch_kib + write_kib) * 1024.0 / calls)
    except OSError:
        pass
    return None


class GemmTimer:
    """Event-pairs around every launch of the dominant kernel (the k-contiguous x k-contiguous 'NT' GEMM that every
    forward Linear maps to) on the stream it is launched on, plus its algorithmic FLOPs (2*M*N*K per launch)."""

    def __init__(self):
        self.pairs = []
        self.flops = 0.0
        self.bytes = 0.0
        self.n = 0

    def install(self):
        from dupl_amd import ops
        self._orig = ops.gemm_raw
        timer = self

        def timed(A, B, C, M, N, K, lda, ldb, ldc, **kw):
            fl = kw.get("flags", 0)
            if fl & 3:    # other operand layouts = other kernel instantiations
                return timer._orig(A, B, C, M, N, K, lda, ldb, ldc, **kw)
            s = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s)
            r = timer._orig(A, B, C, M, N, K, lda, ldb, ldc, **kw)
            e1.record(s)
            timer.pairs.append((e0, e1))
            timer.flops += 2.0 * M * N * K * kw.get("batch", 1)
            timer.bytes += 4.0 * (M * K + N * K + M * N) * kw.get("batch", 1)
            timer.n += 1
            return r

        ops.gemm_raw = timed

    def remove(self):
        from dupl_amd import ops
        ops.gemm_raw = self._orig

    def result(self):
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self.pairs)
        return ms, self.flops, self.n, self.bytes


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    args = parse()
    world, rank, local = build_world(args)
    dev = torch.device("cuda", local)
    from dupl_amd.model.model_dupl import siamese_network
    from dupl_amd.model.PAR import PAR
    from dupl_amd.utils.optimizer import PolyWarmupAdamW
    from dupl_amd.ddp import DistributedDataParallel
    from dupl_amd import trainer

    C = 20 if args.dataset == "voc" else 80
    sargs = trainer.StepArgs() if args.dataset == "voc" else trainer.coco_step_args()
    sargs.share_encoder_pass = not args.no_share_encoder
    torch.manual_seed(0)
    model = siamese_network(args.backbone, num_classes=C + 1, pretrained=False, aux_layer=-3)
    groups = model.get_param_groups()
    model.to(dev)
    if not args.single_stream:
        model.enable_dual_stream(True)
    ddp = DistributedDataParallel(model) if world > 1 else model
    optim = PolyWarmupAdamW(params=[{"params": groups[i], "lr": 6e-5 * (1 if i < 2 else 10), "weight_decay": 1e-2}
                                    for i in range(4)], lr=6e-5, weight_decay=1e-2, betas=(0.9, 0.999),
                            warmup_iter=1500, max_iter=sargs.max_iters, warmup_ratio=1e-6, power=0.9).bind(model.flat_storage)
    par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
    inputs, cls_label, img_box, cls_host = make_batch(args, rank, dev, C)
    phase = "A" if args.n_iter + args.warmup + args.steps < sargs.cam_iters else ("B" if args.n_iter + args.warmup + args.steps < sargs.gmm_iters else "C")
    # phase C: the strongly augmented view (train_final_voc.py:191, RandAugment(5, 10) + flip) is computed inside the step

    def step(i):
        return trainer.train_step(ddp, optim, par, inputs, cls_label, img_box, args.n_iter + i, sargs, cls_label_host=cls_host)

    def barrier():
        if world > 1:
            import torch.distributed as dist
            dist.barrier(device_ids=[local]) if args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    log(f"model on {dev}; starting {args.warmup} warm-up step(s)")
    for i in range(args.warmup):
        tw = time.perf_counter()
        step(i)
        torch.cuda.synchronize()
        log(f"warm-up step {i}: {time.perf_counter() - tw:.3f} s")
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss_val = float(out["loss"].sum().item())
    ms = dt / args.steps * 1e3
    imgs_per_s = world * args.batch * args.steps / dt
    log(f"timed region: {args.steps} steps in {dt:.3f} s -> {imgs_per_s:.2f} img/s")

    flop_ref = FLOP_PER_IMG_PHASE_AB + (FLOP_PHASE_C_AUG + FLOP_PHASE_C_DEAD if phase == "C" else 0.0)
    flop_exec = FLOP_PER_IMG_PHASE_AB - (0.0 if args.no_share_encoder else FLOP_SHARED_PASS) + \
        (FLOP_PHASE_C_AUG if phase == "C" else 0.0)
    roof = None
    if not args.no_roofline:
        model.enable_dual_stream(False)   # per-kernel durations are only meaningful without a co-running stream
        timer = GemmTimer()
        timer.install()
        step(args.warmup + args.steps)
        gms, gflops, gn, gbytes = timer.result()
        timer.remove()
        ach = gflops / (gms * 1e-3)
        roof = {"bound": "mfma", "kernel": "gemm_f32_kernel<false,false,64,1,*> (v_mfma_f32_32x32x2_f32; 128- and 64-column tile instantiations)", "achieved": round(ach / 1e12, 2),
                "peak": round(PEAK_F32_MFMA / 1e12, 1), "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA, 4),
                "traffic": pmc_traffic_per_launch(), "traffic_unit": "bytes/launch (PMC FETCH_SIZE*2 + WRITE_SIZE, "
                "profiles/r01_final_pmc_hbm.txt)", "algorithmic_bytes_per_launch": round(gbytes / max(gn, 1)),
                "launches_per_step": gn, "avg_launch_us": round(gms * 1e3 / max(gn, 1), 1),
                "kernel_share_of_step": round(gms / ms, 3),
                "step_frac_of_peak": round(imgs_per_s / world * flop_exec / PEAK_F32_MFMA, 4),
                "step_flop_per_img": {"reference_algorithm": flop_ref, "executed": flop_exec},
                "note": "HIP event pairs around every launch of the kernel during one extra step run right after the timed "
                        "region (same stream, same workload); algorithmic flops = 2*M*N*K per launch"}
    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline == "auto":
        log("timing the CPU oracle on the host cores (bounded sample)")
        cpu = cpu_baseline(args, C)
        log(f"cpu baseline: {cpu}")
    if rank == 0:
        rec = {"metric": f"training img/s at 448^2, {'VOC' if args.dataset == 'voc' else 'COCO'} dual-student ViT-B/16, phase {phase} step",
               "value": round(imgs_per_s, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"{'VOC2012' if args.dataset == 'voc' else 'MSCOCO2014'} {args.size}^2 dual-student "
                                      f"{args.backbone} + ms-CAM(1.0,0.5,1.5) + PAR + cross seg loss, phase {phase}, "
                                      f"{args.batch} img/GPU, DDP world_size={world}",
                          "global_batch": world * args.batch, "img_per_gpu": args.batch, "num_classes": C + 1,
                          "n_iter": args.n_iter, "parallelism": f"dp{world}", "student_streams": 1 if args.single_stream else 2,
                          "shared_scale1_encoder_pass": not args.no_share_encoder,
                          "loss": round(loss_val, 5)},
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(rec))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
