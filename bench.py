#!/usr/bin/env python
"""bench.py -- DuPL training-step throughput on MI355X (BASELINE.json metric: training img/s at 448^2).

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched through torch.distributed.run)

A "step" is one full iteration of the reference loop (train_final_voc.py:174-472 / train_final_coco.py:170-462) on a
synthetic batch that is already resident in HBM: ms-CAM for both students (3 scales x flip), dual-student
forward/backward, CAM->label, PTC, PAR refinement (high+low) for both students, cross seg loss, discrepancy loss,
gradient all-reduce (N > 1) and the PolyWarmupAdamW update.  Nothing is skipped or cached across steps.

Workload: BASELINE.json configs[1] -- VOC2012 448^2, deit_base_patch16_224, 4 images per GPU -- on EVERY rank, whatever N
(the path shards by image: each rank steps its own 4 images, the only exchange is the gradient all-reduce), so that `value`
(whole-job img/s) over N = 1, 2, 4, 8 is a weak-scaling curve: per-GPU work is the same at every N.
For N > 1 the line also carries `listed_config`: the configuration BASELINE.json lists for that N, measured right after
(the reference's "bs" is the GLOBAL batch; these shrink the per-GPU batch to 2, so they are not points of the
weak-scaling curve):
    N = 2   configs[2]  VOC2012  448^2, deit_base_patch16_224, 2 img/GPU (global 4)
    N = 4   configs[3]  MSCOCO14 448^2, 81 classes,            2 img/GPU (global 8)
    N = 8   configs[4]  MSCOCO14 448^2, vit_base_patch16_224,  2 img/GPU (global 16)
(other N: the COCO 2 img/GPU workload).  --dataset / --batch / --backbone override the main workload.  Rank 0 prints
ONE JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
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
PEAK_F16_MFMA = 2500e12               # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
KERNEL_F16X3 = ("gemm_f16x3_kernel (csrc/gemm_split.hip): every Linear forward as an fp32-equivalent split product -- operands as "
                "fp16 hi / lo planes, 3 v_mfma_f32_32x32x16_f16 per 32x32x16 block, fp32 accumulate; peak = dense f16 MFMA peak / 3")
KERNEL_F32 = "gemm_f32_kernel<false,false,...> (v_mfma_f32_32x32x2_f32; every Linear forward, all tile instantiations)"

DTYPE = {"f16x3": "f32 storage / accumulation / results; Linear GEMMs (forward, dgrad, wgrad), attention (forward, backward) and "
                  "the decoder convs as fp32-equivalent f16x3 split products on the f16 MFMA (operands as fp16 hi / lo planes, "
                  "3 products per block, fp32 accumulate; error vs fp64 <= the f32 MFMA kernels'); CAM / classifier heads, conv8, "
                  "Gram, patch-embedding weight gradient, norms, losses, PAR, optimiser in f32",
         "f32": "f32"}
CONFIG_BY_N = {1: ("voc", 4, "deit_base_patch16_224", "configs[1]"),
               2: ("voc", 2, "deit_base_patch16_224", "configs[2]"),
               4: ("coco", 2, "deit_base_patch16_224", "configs[3]"),
               8: ("coco", 2, "vit_base_patch16_224", "configs[4]")}
DEFAULT_N_ITER = {"voc": 5000, "coco": 20000}    # phase B of either schedule (PTC + PAR refinement + cross seg loss)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=None, help="images per GPU (default: BASELINE's configuration for this N)")
    ap.add_argument("--size", type=int, default=448)
    ap.add_argument("--dataset", default=None, choices=["voc", "coco"])
    ap.add_argument("--backbone", default=None)
    ap.add_argument("--n-iter", type=int, default=None, help="iteration index the step pretends to be (default: phase B)")
    ap.add_argument("--no-listed", "--no-weak4", dest="no_listed", action="store_true",
                    help="N > 1: skip the second measurement (the configuration BASELINE.json lists for this N)")
    ap.add_argument("--no-second", action="store_true",
                    help="N = 1: skip the second measurement (the metric's other configuration, COCO 8 img/GPU)")
    ap.add_argument("--cpu-baseline", default="auto", choices=["auto", "skip"])
    ap.add_argument("--cpu-size", type=int, default=448)
    ap.add_argument("--cpu-batch", type=int, default=1, help="images per CPU-oracle step (stated in the JSON)")
    ap.add_argument("--cpu-threads", type=int, default=0,
                    help="host threads for the timed CPU steps (0 = pick the fastest of the recorded sweep over "
                         "{physical cores, 64, 32})")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed CPU-oracle steps after one warm-up step")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-exact-f32", action="store_true",
                    help="skip the second measurement of the same workload on the exact-f32 MFMA kernels (f16x3 runs only)")
    ap.add_argument("--pmc-profile", default=os.environ.get("DUPL_PMC_PROFILE", "profiles/r06_final_pmc_hbm.txt"),
                    help="PMC summary (tools/profile_round.sh) roofline.traffic is read from; ignored (traffic = null) "
                         "unless its '# csrc_sha256:' header matches the kernel sources of THIS build")
    ap.add_argument("--no-share-encoder", action="store_true",
                    help="run the training forward's encoder pass separately from ms-CAM's identical scale-1.0 pass, exactly "
                         "like the reference does (default: computed once and shared; outputs are bit-identical)")
    ap.add_argument("--single-stream", action="store_true", help="run the two students back to back on one stream")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend for --gpus > 1 (nccl = RCCL over xGMI; gloo only exists so that the "
                         "multi-rank control flow can be exercised by two ranks sharing one GPU in the tests)")
    return ap.parse_args()


def self_launch(args):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (no WORLD_SIZE in the environment): start the N ranks here,
    the way the reference is started (README.md:83,93: torch.distributed.run --nproc_per_node=N), and pass the ranks' exit
    code on.  Refuses to run when the node has fewer than N GPUs -- N ranks never silently share a device."""
    import socket
    import subprocess
    share = os.environ.get("DUPL_BENCH_RANKS_SHARE_GPU0") == "1"      # test hook (gloo, every rank on device 0)
    have = torch.cuda.device_count()
    if have < args.gpus and not share:
        raise SystemExit(f"bench.py --gpus {args.gpus}: this node exposes {have} GPU(s); refusing to run {args.gpus} ranks on fewer devices")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"--gpus {args.gpus} without a launcher: starting {args.gpus} ranks through torch.distributed.run (port {port})")
    raise SystemExit(subprocess.call(cmd, env=env))


def build_world(args):
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was started with WORLD_SIZE={world}: the line would report the wrong n_gpus")
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if os.environ.get("DUPL_BENCH_RANKS_SHARE_GPU0") == "1":    # test hook: every rank on device 0 (needs --backend gloo)
            local = 0
        elif torch.cuda.device_count() < world:
            raise SystemExit(f"bench.py --gpus {world}: this node exposes {torch.cuda.device_count()} GPU(s)")
        torch.cuda.set_device(local)
        dist.init_process_group(backend=args.backend)
        assert dist.get_world_size() == world == args.gpus, (dist.get_world_size(), world, args.gpus)
    else:
        torch.cuda.set_device(0)
    return world, rank, local


def physical_cores() -> int:
    """Physical cores of the host (unique (socket, core) pairs of /proc/cpuinfo), limited to this process's affinity."""
    try:
        pairs, phys = set(), None
        for line in open("/proc/cpuinfo"):
            if line.startswith("physical id"):
                phys = line.split(":")[1].strip()
            elif line.startswith("core id"):
                pairs.add((phys, line.split(":")[1].strip()))
        n = len(pairs)
    except OSError:
        n = 0
    logical = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n or logical, logical))


def cpu_baseline(args, dataset, C):
    """The oracle (torch-CPU restatement of the reference, kind 'port') timed on this box's host cores on a BOUNDED
    sample, SURVEY 8(d) protocol: one warm-up step, a recorded thread-count sweep (one step each at {physical cores,
    64, 32}), then `--cpu-steps` (>= 3) timed steps at the fastest count; one process, fp32, same 448^2 dual-student
    phase-B step incl. backward and the AdamW update, `--cpu-batch` images per step (stated)."""
    from oracle import dupl_oracle as O
    cfg = O.VIT_BASE
    NC = C + 1
    b = max(1, args.cpu_batch)
    pp = O.make_siamese_params(cfg, NC, seed=3, randomize_affine=False)
    inputs, cls_label, img_box = O.synthetic_batch(b, C, args.cpu_size, seed=100)
    sargs = O.StepArgs() if dataset == "voc" else O.coco_step_args()
    n_iter = DEFAULT_N_ITER[dataset]

    def one_step(threads):
        torch.set_num_threads(threads)
        leaf = {k: v.clone().requires_grad_(k.split(".", 1)[1] not in ("encoder.pos_embed",)) for k, v in pp.items()}
        t0 = time.perf_counter()
        loss, _ = O.train_step_losses(leaf, inputs, cls_label, img_box, n_iter, cfg, sargs)
        loss.backward()
        for k, p in leaf.items():
            if p.grad is None:
                continue
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            with torch.no_grad():
                O.adamw_update(p, p.grad, m, v, 1, 6e-5 if O.param_group_index(k) < 2 else 6e-4)
        return time.perf_counter() - t0

    phys = physical_cores()
    cand = [args.cpu_threads] if args.cpu_threads > 0 else sorted({phys, min(64, phys), min(32, phys), min(16, phys)}, reverse=True)
    one_step(cand[0])                                   # warm-up (allocator, thread pool, page-in)
    sweep = {}
    if len(cand) > 1:
        for t in cand:
            sweep[str(t)] = round(b / one_step(t), 5)
        best = int(max(sweep, key=sweep.get))
    else:
        best = cand[0]
    times = [one_step(best) for _ in range(max(1, args.cpu_steps))]
    dt = sum(times) / len(times)
    return {"value": round(b / dt, 5), "unit": "img/s", "cores": best, "kind": "port", "physical_cores": phys,
            "logical_cpus": os.cpu_count(), "batch": b, "warmup_steps": 1, "timed_steps": len(times),
            "step_seconds": [round(t, 2) for t in times], "threads_sweep_img_per_s": sweep,
            "sample": f"{len(times)} timed phase-B steps (after 1 warm-up) of oracle/dupl_oracle.py: dual ViT-B/16, "
                      f"{args.cpu_size}^2, {dataset.upper()} {NC} classes, b={b} image(s)/step incl. backward + AdamW, fp32, "
                      f"torch {torch.__version__} CPU, {best} threads (fastest of the sweep; {phys} physical cores): "
                      f"{dt:.1f} s/step"}


def csrc_digest() -> str:
    from dupl_amd.build import source_digest
    return source_digest()


def pmc_traffic_per_launch(path, kernel_prefix="gemm_f32_kernel<false, false", gemm_mode=None):
    """HBM-side bytes per launch of the dominant kernel from a PMC summary written by tools/profile_round.sh (separate
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this same workload, tools/rocpd_pmc.py).  FETCH_SIZE is
    doubled (gfx950 tallies 128-B requests at 64 B, MI355X_MICROARCH.md 'HBM'); it counts L2->fabric requests, i.e.
    Infinity-Cache hits are included.  The summary is only trusted if its '# csrc_sha256:' header equals the digest of
    the kernel sources of THIS build; otherwise (or if absent) -> (None, reason)."""
    full = path if os.path.isabs(path) else os.path.join(ROOT, path)
    try:
        lines = open(full).read().splitlines()
    except OSError:
        return None, f"no PMC summary at {path}"
    tag = {}
    for line in lines:
        if line.startswith("# ") and ":" in line:
            k, v = line[2:].split(":", 1)
            tag[k.strip()] = v.strip()
    have = csrc_digest()
    if tag.get("csrc_sha256") != have:
        return None, (f"{path} is stale or untagged (its csrc_sha256 {tag.get('csrc_sha256', 'missing')[:12]} != "
                      f"{have[:12]} of this build): re-run tools/profile_round.sh")
    if gemm_mode is not None and tag.get("gemm_mode", "f16x3") != gemm_mode:
        return None, (f"{path} was measured in {tag.get('gemm_mode', 'f16x3')} mode, this run is in {gemm_mode} mode: its per-kernel "
                      "traffic belongs to another launch mix")
    calls = fetch_kib = write_kib = 0.0
    for line in lines:
        if line.startswith(kernel_prefix):      # every column-tile instantiation of the NT kernel
            parts = line.split()
            calls += float(parts[-4])
            fetch_kib += float(parts[-2])
            write_kib += float(parts[-1])
    if not calls:
        return None, f"{path}: no row for {kernel_prefix}"
    return round((2.0 * fetch_kib + write_kib) * 1024.0 / calls), \
        f"PMC FETCH_SIZE*2 + WRITE_SIZE per launch, {path} (tag {tag.get('tag', '?')}, git {tag.get('git_head', '?')[:10]})"


# kernels of the PMC summary that make up a family of the roofline section (prefix match on the kernel name)
FAMILY_KERNELS = {"fwd_f1": ("gemm_f16x3_ring_kernel<4, 2, 2, 4, 2, 2, true>", "gemm_f16x3_ring_kernel<2, 2, 4, 2, 2, 3, true>",
                             "gemm_f16x3_pring_kernel<2, 2, 4, 2, 2, false, true"),
                  "dgrad": ("gemm_f16x3_km_kernel<2, 2, 4, 2, 2, true, 1, false, true", "gemm_f16x3_km_kernel<2, 2, 4, 2, 2, false, 0, false, true"),
                  "wgrad": ("gemm_f16x3_km_group_kernel", "gemm_f16x3_km_kernel<2, 2, 4, 2, 2, true, 1, true, true",
                            "gemm_f16x3_km_kernel<2, 2, 4, 2, 2, false, 2, true, true")}


def pmc_family_traffic(path, steps_profiled=None):
    """{family: HBM-side MB per step} from a digest-matching PMC summary (FETCH_SIZE * 2 + WRITE_SIZE of the family's kernels, divided
    by the steps the profiled command ran: `# steps:` header, default 4 = --steps 3 --warmup 1); {} if the summary is not usable."""
    full = path if os.path.isabs(path) else os.path.join(ROOT, path)
    try:
        lines = open(full).read().splitlines()
    except OSError:
        return {}
    tag = {}
    for line in lines:
        if line.startswith("# ") and ":" in line:
            k, v = line[2:].split(":", 1)
            tag[k.strip()] = v.strip()
    if tag.get("csrc_sha256") != csrc_digest():
        return {}
    steps = steps_profiled or int(tag.get("steps", "4"))
    out = {}
    for fam, prefixes in FAMILY_KERNELS.items():
        kib = 0.0
        for line in lines:
            if any(line.startswith(pf) for pf in prefixes):
                parts = line.split()
                kib += 2.0 * float(parts[-2]) + float(parts[-1])
        if kib:
            out[fam] = kib * 1024.0 / 1e6 / steps
    return out


class GemmTimer:
    """Event-pairs around every launch of the dominant kernel -- the k-contiguous x k-contiguous 'NT' GEMM that every
    forward Linear maps to: dupl_gemm_f16x3 (f16x3 mode) or the NT instantiation of dupl_gemm_f32 (f32 mode) -- on the
    stream it is launched on, plus its algorithmic FLOPs (2*M*N*K per launch)."""

    FAMILIES = ("fwd_f1", "fwd_f0", "dgrad", "wgrad", "attention_fwd", "attention_bwd")

    def __init__(self):
        self.pairs = {"f16x3": [], "f32": []}
        self.flops = {"f16x3": 0.0, "f32": 0.0}
        self.bytes = {"f16x3": 0.0, "f32": 0.0}
        # per family of split (f16x3) launches: [(event, event)], algorithmic flops -- the GEMM families are subsets of
        # pairs["f16x3"]; the two attention families are the split attention kernels (own launches, same MFMA, same peak)
        self.fam_pairs = {k: [] for k in self.FAMILIES}
        self.fam_flops = {k: 0.0 for k in self.FAMILIES}
        self.fam_bytes = {k: 0.0 for k in self.FAMILIES}      # algorithmic operand + result bytes (GEMM families)

    def _timed_family(self, fam, fn, flops):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        r = fn()
        e1.record(s)
        self.fam_pairs[fam].append((e0, e1))
        self.fam_flops[fam] += flops
        return r

    def _timed(self, kind, fn, M, N, K, batch=1, mn_tensors=1, family=None):
        """mn_tensors: how many [M, N] 4-byte-per-element tensors the launch reads or writes (fp32 result, result planes,
        residual, stored pre-activation, activation-gradient operand, accumulated-into result)."""
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        r = fn()
        e1.record(s)
        self.pairs[kind].append((e0, e1))
        self.flops[kind] += 2.0 * M * N * K * batch
        self.bytes[kind] += 4.0 * (M * K + N * K + mn_tensors * M * N) * batch     # two fp16 planes = 4 bytes per element
        if family is not None:
            self.fam_pairs[family].append((e0, e1))
            self.fam_flops[family] += 2.0 * M * N * K * batch
            self.fam_bytes[family] += 4.0 * (M * K + N * K + mn_tensors * M * N) * batch
        return r

    def install(self):
        from dupl_amd import ops
        self._orig, self._orig16 = ops.gemm_raw, ops.linear16
        self._orig_af, self._orig_ab = ops.attention_fwd16, ops.attention_bwd16
        self._orig_afs = ops.attention_fwd16_segs
        self._orig_wg = ops.wgrad16_group
        timer = self

        def timed(A, B, C, M, N, K, lda, ldb, ldc, **kw):
            if kw.get("flags", 0) & 3:    # other operand layouts = other kernel instantiations
                return timer._orig(A, B, C, M, N, K, lda, ldb, ldc, **kw)
            return timer._timed("f32", lambda: timer._orig(A, B, C, M, N, K, lda, ldb, ldc, **kw), M, N, K, kw.get("batch", 1))

        def timed16(x, W, *a, **kw):
            f32_out = kw.get("want_f32", True) or kw.get("out") is not None
            planes_out = kw.get("want16", False) or kw.get("out16") is not None
            aux = sum(kw.get(k) is not None for k in ("res", "store_pre", "dgelu_of", "relumask_of"))
            mn = int(f32_out) * (2 if kw.get("accumulate", False) else 1) + int(planes_out) + aux
            # family: weight gradients accumulate; data gradients carry the inverse scale of their scaled gradient planes;
            # everything else is a forward Linear / decoder conv, on format 1 (single accumulator) or format 0 planes
            # (a stream-K data gradient accumulates into a zero-filled dx: k-major B only; a weight gradient has a k-major A too)
            acc_, akm_ = kw.get("accumulate", False), kw.get("a_kmajor", False)
            wg = acc_ and (akm_ or not kw.get("b_kmajor", False))
            fam = "wgrad" if wg else ("dgrad" if kw.get("alpha") is not None else ("fwd_f1" if getattr(x, "exp", 0) else "fwd_f0"))
            # k-major operands (the backward GEMMs on the forward's planes) are stored [K, rows]; the algorithmic contraction
            # length of a weight gradient is the token count, not its zero-padded k_pad
            a_km, b_km = kw.get("a_kmajor", False), kw.get("b_kmajor", False)
            M_ = x.cols if a_km else x.rows
            N_ = W.cols if b_km else W.rows
            K_ = getattr(x, "valid_rows", x.rows) if a_km else x.cols
            return timer._timed("f16x3", lambda: timer._orig16(x, W, *a, **kw), M_, N_, K_, mn_tensors=mn, family=fam)

        def timed_af(qkv16, B, N, H, hd, scale, *a, **kw):      # S = Q K^T and O = P V: 4 N^2 hd per (image, head)
            return timer._timed_family("attention_fwd", lambda: timer._orig_af(qkv16, B, N, H, hd, scale, *a, **kw),
                                       4.0 * B * H * N * N * hd)

        def timed_ab(qkv16, out, dout, lse, B, N, H, hd, scale, *a, **kw):   # S recomputed, dP, dV, dQ, dK: 10 N^2 hd
            return timer._timed_family("attention_bwd", lambda: timer._orig_ab(qkv16, out, dout, lse, B, N, H, hd, scale, *a, **kw),
                                       10.0 * B * H * N * N * hd)

        def timed_wg(items, **kw):        # the grouped weight gradients of a transformer block: ONE launch of the same kernel family
            fl = by = 0.0
            for dy16, x16, out, _ in items:
                m_, n_, k_ = dy16.cols, x16.cols, getattr(dy16, "valid_rows", dy16.rows)
                fl += 2.0 * m_ * n_ * k_
                by += 4.0 * (m_ * k_ + n_ * k_ + 2 * m_ * n_)
            s_ = torch.cuda.current_stream()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(s_)
            r = timer._orig_wg(items, **kw)
            e1.record(s_)
            timer.pairs["f16x3"].append((e0, e1))
            timer.flops["f16x3"] += fl
            timer.bytes["f16x3"] += by
            timer.fam_pairs["wgrad"].append((e0, e1))
            timer.fam_flops["wgrad"] += fl
            timer.fam_bytes["wgrad"] += by
            return r

        def timed_afs(qkv16, segs, H, hd, scale, *a, **kw):     # several batches in one launch: the same 4 N^2 hd per (image, head)
            return timer._timed_family("attention_fwd", lambda: timer._orig_afs(qkv16, segs, H, hd, scale, *a, **kw),
                                       sum(4.0 * sg[1] * H * sg[2] * sg[2] * hd for sg in segs))

        ops.gemm_raw, ops.linear16 = timed, timed16
        ops.attention_fwd16, ops.attention_bwd16 = timed_af, timed_ab
        ops.attention_fwd16_segs = timed_afs
        ops.wgrad16_group = timed_wg

    def remove(self):
        from dupl_amd import ops
        ops.gemm_raw, ops.linear16 = self._orig, self._orig16
        ops.attention_fwd16, ops.attention_bwd16 = self._orig_af, self._orig_ab
        ops.attention_fwd16_segs = self._orig_afs
        ops.wgrad16_group = self._orig_wg

    def reset(self):
        self.pairs = {k: [] for k in self.pairs}
        self.flops = {k: 0.0 for k in self.flops}
        self.bytes = {k: 0.0 for k in self.bytes}
        self.fam_pairs = {k: [] for k in self.FAMILIES}
        self.fam_flops = {k: 0.0 for k in self.FAMILIES}
        self.fam_bytes = {k: 0.0 for k in self.FAMILIES}

    @staticmethod
    def _union_ms(pairs, base_event):
        iv = sorted((base_event.elapsed_time(a), base_event.elapsed_time(b)) for a, b in pairs)
        busy, cur0, cur1 = 0.0, None, None
        for a, b in iv:
            if cur1 is None or a > cur1:
                if cur1 is not None:
                    busy += cur1 - cur0
                cur0, cur1 = a, b
            else:
                cur1 = max(cur1, b)
        if cur1 is not None:
            busy += cur1 - cur0
        return busy

    def families(self, passes, peak, base_event=None, steps=1):
        """Per family: algorithmic flops per step, launches per step and time per step -- single stream: sum over the launches
        of the per-launch minimum over `passes` repetitions of the step; with base_event (two student streams): the union of
        the family's event intervals / steps."""
        torch.cuda.synchronize()
        out = {}
        for fam in self.FAMILIES:
            v = self.fam_pairs[fam]
            if not v:
                continue
            if base_event is None:
                n = len(v) // passes
                assert n * passes == len(v)
                t = [[a.elapsed_time(b) for a, b in v[p * n:(p + 1) * n]] for p in range(passes)]
                ms = sum(min(col) for col in zip(*t))
                fl = self.fam_flops[fam] / passes
                by = self.fam_bytes[fam] / passes
            else:
                n = len(v) // steps
                ms = self._union_ms(v, base_event) / steps
                fl = self.fam_flops[fam] / steps
                by = self.fam_bytes[fam] / steps
            out[fam] = {"tflop_per_step": round(fl / 1e12, 3), "launches_per_step": n, "ms_per_step": round(ms, 3),
                        "algorithmic_mb_per_step": round(by / 1e6, 1) if by else None,
                        "achieved": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None,
                        "frac": round(fl / (ms * 1e-3) / peak, 4) if ms > 0 else None}
        return out

    def busy_union(self, kind, base_event):
        """(ms during which at least one launch of `kind` was running on ANY stream, flops) since base_event: with the two
        students on two streams their GEMMs overlap, so per-launch durations double-count the chip; the union does not."""
        torch.cuda.synchronize()
        return self._union_ms(self.pairs[kind], base_event), self.flops[kind]

    def result(self, passes=1):
        """(kind, ms, flops, launches, bytes) per pass of the kind with the larger total time.  With passes > 1 (the same
        step repeated, so launch i of every pass is the same GEMM) each launch counts with its MINIMUM over the passes: an
        event pair also spans any moment the stream ran dry because the host fell behind, which is not kernel time."""
        torch.cuda.synchronize()
        out = {}
        for k, v in self.pairs.items():
            n = len(v) // passes
            assert n * passes == len(v), "the repeated step issued a different number of launches"
            t = [[a.elapsed_time(b) for a, b in v[p * n:(p + 1) * n]] for p in range(passes)]
            out[k] = (sum(min(col) for col in zip(*t)) if n else 0.0, n)
        kind = max(out, key=lambda k: out[k][0])
        return kind, out[kind][0], self.flops[kind] / passes, out[kind][1], self.bytes[kind] / passes


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


class Workload:
    """One (dataset, per-GPU batch, backbone) configuration: model, optimiser, resident synthetic batch, step()."""

    def __init__(self, args, world, rank, local, dataset, batch, backbone, n_iter):
        from dupl_amd.model.model_dupl import siamese_network
        from dupl_amd.model.PAR import PAR
        from dupl_amd.utils.optimizer import PolyWarmupAdamW
        from dupl_amd.ddp import DistributedDataParallel
        from dupl_amd.synthetic import synthetic_batch
        from dupl_amd import trainer
        self.args, self.world, self.local = args, world, local
        self.dataset, self.batch, self.backbone, self.n_iter = dataset, batch, backbone, n_iter
        self.dev = dev = torch.device("cuda", local)
        self.C = C = 20 if dataset == "voc" else 80
        self.sargs = trainer.StepArgs() if dataset == "voc" else trainer.coco_step_args()
        self.sargs.share_encoder_pass = not args.no_share_encoder
        torch.manual_seed(0)
        self.model = model = siamese_network(backbone, num_classes=C + 1, pretrained=False, aux_layer=-3)
        groups = model.get_param_groups()
        model.to(dev)
        if not args.single_stream:
            model.enable_dual_stream(True)
        self.ddp = DistributedDataParallel(model) if world > 1 else model
        self.optim = PolyWarmupAdamW(params=[{"params": groups[i], "lr": 6e-5 * (1 if i < 2 else 10), "weight_decay": 1e-2}
                                             for i in range(4)], lr=6e-5, weight_decay=1e-2, betas=(0.9, 0.999),
                                     warmup_iter=1500, max_iter=self.sargs.max_iters, warmup_ratio=1e-6,
                                     power=0.9).bind(model.flat_storage)
        self.par = PAR(num_iter=10, dilations=[1, 2, 4, 8, 12, 24]).to(dev)
        inputs, cls_label, img_box = synthetic_batch(batch, C, args.size, seed=100 + rank)
        self.inputs, self.cls_label, self.img_box, self.cls_host = inputs.to(dev), cls_label.to(dev), img_box, cls_label
        last = n_iter + args.warmup + args.steps
        self.phase = "A" if last < self.sargs.cam_iters else ("B" if last < self.sargs.gmm_iters else "C")
        self._trainer = trainer
        lo, hi = model.flat_storage.trainable_range(0)
        self.grad_bytes = 4 * (hi - lo) * model.flat_storage.n_students

    def step(self, i):
        # phase C: the strongly augmented view (train_final_voc.py:191) is computed inside the step, on the device
        out = self._trainer.train_step(self.ddp, self.optim, self.par, self.inputs, self.cls_label, self.img_box,
                                       self.n_iter + i, self.sargs, cls_label_host=self.cls_host)
        self._armed_last = bool(getattr(self.optim, "_was_armed_exchange", False))
        return out

    def _ranks_seen(self, dist):
        """{"world": dist.get_world_size(), "devices": [...one id per rank...], "distinct_devices": n}."""
        import socket
        props = torch.cuda.get_device_properties(self.dev)
        ident = None
        try:
            ident = str(props.uuid)
        except Exception:
            pass
        if not ident or set(ident.replace("-", "")) <= {"0"}:
            if hasattr(props, "pci_bus_id"):
                ident = "pci-%04x:%02x:%02x" % (getattr(props, "pci_domain_id", 0), props.pci_bus_id, getattr(props, "pci_device_id", 0))
            else:
                ident = f"device-index-{self.dev.index}"
        me = f"{socket.gethostname()}/{ident}"
        got = [None] * self.world
        dist.all_gather_object(got, me)
        return {"world": int(dist.get_world_size()), "devices": got, "distinct_devices": len(set(got))}

    def barrier(self):
        if self.world > 1:
            import torch.distributed as dist
            dist.barrier(device_ids=[self.local]) if self.args.backend == "nccl" else dist.barrier()
        torch.cuda.synchronize()

    def measure(self, label):
        """W warm-up steps, then EXACTLY K steps between barrier + synchronize, max over ranks."""
        args, world = self.args, self.world
        log(f"[{label}] {self.dataset} {self.batch} img/GPU {self.backbone} on {self.dev}; {args.warmup} warm-up step(s)")
        for i in range(args.warmup):
            tw = time.perf_counter()
            self.step(i)
            torch.cuda.synchronize()
            log(f"[{label}] warm-up step {i}: {time.perf_counter() - tw:.3f} s")
        red = self.ddp.reducer if world > 1 else None
        if red is not None:
            red.pop_stats()
        self.barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            out = self.step(args.warmup + i)
        self.barrier()
        dt = time.perf_counter() - t0
        comm = {"world": world, "backend": (args.backend + ("(RCCL)" if args.backend == "nccl" else "")) if world > 1 else "none",
                "grad_bytes_per_rank": self.grad_bytes, "allreduce_bytes": 0, "allreduce_calls": 0, "comm_exposed_ms": 0.0}
        if world > 1:
            import torch.distributed as dist
            t = torch.tensor([dt], device=self.dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
            st = red.pop_stats()
            comm["allreduce_bytes"] = st["allreduce_bytes"] // max(1, args.steps)      # per step, this rank
            comm["allreduce_calls"] = st["allreduce_calls"] // max(1, args.steps)
            # exposed communication: one extra step with the step's compute drained before the waits of finish()
            # (profiling inserts a stream join, so it runs outside the timed region)
            red.profile = True
            self.step(args.warmup + args.steps)
            torch.cuda.synchronize()
            red.profile = False
            stp = red.pop_stats()
            ex = stp["exposed_ms"]
            e = torch.tensor([sum(ex), stp["stall_ms"]], device=self.dev, dtype=torch.float64)
            dist.all_reduce(e, op=dist.ReduceOp.MAX)
            comm["comm_exposed_ms"] = round(float(e[0].item()), 3)
            # how long the student streams stood in the waits of the pieces taken in DURING the backward pass (sum over both streams,
            # max over ranks): an upper bound of what those waits cost -- the other student's kernels run meanwhile
            comm["comm_stall_ms_in_backward"] = round(float(e[1].item()), 3)
            comm["optimizer_in_exchange"] = bool(self._armed_last)     # the AdamW launches rode behind each piece's all-reduce
            # WHO took part: the process group's size and the distinct devices behind it (an all-gather of every rank's device
            # UUID / PCI address + host) -- the first multi-GPU run proves that N ranks sat on N distinct GPUs
            comm["ranks_seen"] = self._ranks_seen(dist)
            # self-validation of the exchange (the first real multi-GPU run must prove itself): every rank started from
            # rank 0's broadcast and applied the same averaged gradients, so the parameter buffers must be bit-identical --
            # two order-sensitive 64-bit checksums of the raw bits, compared over all ranks
            bits = self.model.flat_storage.data.view(torch.int32).to(torch.int64)
            idx = torch.arange(bits.numel(), device=self.dev, dtype=torch.int64)
            cs = torch.stack([bits.sum(), (bits * ((idx % 65521) + 1)).sum()])
            gathered = [torch.zeros_like(cs) for _ in range(world)]
            dist.all_gather(gathered, cs)
            same = all(bool(torch.equal(g, gathered[0])) for g in gathered)
            comm["param_checksum"] = [int(v) for v in cs.tolist()]
            comm["params_identical_on_all_ranks"] = same
            comm["world_matches_gpus"] = (world == args.gpus == dist.get_world_size())
            assert same, f"parameter buffers diverged across ranks after {args.warmup + args.steps + 1} steps: {[g.tolist() for g in gathered]}"
            assert comm["world_matches_gpus"], (world, args.gpus, dist.get_world_size())
        ms = dt / args.steps * 1e3
        value = world * self.batch * args.steps / dt
        log(f"[{label}] timed region: {args.steps} steps in {dt:.3f} s -> {value:.2f} img/s")
        return {"value": value, "ms": ms, "loss": float(out["loss"].sum().item()), "comm": comm}

    def describe(self):
        name = "VOC2012" if self.dataset == "voc" else "MSCOCO2014"
        return (f"{name} {self.args.size}^2 dual-student {self.backbone} + ms-CAM(1.0,0.5,1.5) + PAR + cross seg loss, "
                f"phase {self.phase}, {self.batch} img/GPU, DDP world_size={self.world}")


def main():
    args = parse()
    world, rank, local = build_world(args)
    # main workload: configs[1] per rank at every N (weak scaling); the configuration listed for N > 1 is measured second
    d_ds, d_b, d_bb, d_cfg = CONFIG_BY_N[1]
    dataset = args.dataset or d_ds
    batch = args.batch or d_b
    backbone = args.backbone or d_bb
    listed = (dataset, batch, backbone) == (d_ds, d_b, d_bb)
    if world > 1:
        d_cfg = "configs[1] on every rank (weak scaling of the N = 1 workload)"
    n_iter = args.n_iter if args.n_iter is not None else DEFAULT_N_ITER[dataset]

    from dupl_amd import engine
    gemm_mode = engine.GEMM_MODE
    wl = Workload(args, world, rank, local, dataset, batch, backbone, n_iter)
    res = wl.measure("main")
    ms, imgs_per_s, phase, C = res["ms"], res["value"], wl.phase, wl.C
    guard_main = wl.model.flat_storage.guard          # of the main workload (later legs build other models)

    flop_ref = FLOP_PER_IMG_PHASE_AB + (FLOP_PHASE_C_AUG + FLOP_PHASE_C_DEAD if phase == "C" else 0.0)
    flop_exec = FLOP_PER_IMG_PHASE_AB - (0.0 if args.no_share_encoder else FLOP_SHARED_PASS) + \
        (FLOP_PHASE_C_AUG if phase == "C" else 0.0)
    roof = None
    if not args.no_roofline:
        wl.model.enable_dual_stream(False)   # per-kernel durations are only meaningful without a co-running stream
        timer = GemmTimer()
        timer.install()
        for _ in range(3):
            wl.step(args.warmup + args.steps + 1)
        kind, gms, gflops, gn, gbytes = timer.result(passes=3)
        if kind == "f16x3":
            # one fp32-equivalent multiply-add costs 3 f16 MFMA products (hi*hi, hi*lo, lo*hi): the roofline of the
            # ALGORITHMIC flops is the dense f16 MFMA peak / 3
            peak, kname, prefix = PEAK_F16_MFMA / 3.0, KERNEL_F16X3, "gemm_f16x3"
        else:
            peak, kname, prefix = PEAK_F32_MFMA, KERNEL_F32, "gemm_f32_kernel<false, false"
        fam_single = timer.families(3, PEAK_F16_MFMA / 3.0) if kind == "f16x3" else None
        timer.remove()
        ach1 = gflops / (gms * 1e-3)
        single = {"achieved": round(ach1 / 1e12, 2), "frac": round(ach1 / peak, 4), "avg_launch_us": round(gms * 1e3 / max(gn, 1), 1),
                  "kernel_ms_per_step": round(gms, 2), "families": fam_single,
                  "note": "students back to back on ONE stream: HIP event pairs around every launch during three extra steps, per "
                          "launch the minimum of the three (drops host-side gaps) -- clean per-kernel durations (they agree with "
                          "the rocprofv3 kernel trace), but NOT the configuration the timed region runs in"}
        dual = None
        if not args.single_stream:
            wl.model.enable_dual_stream(True)
            # the same kernels in the regime the timed region runs in: both students' launches in flight on two streams.
            # Per-launch durations are meaningless there (every launch shares the chip with the other student's), so the
            # figure is total algorithmic flops / time during which at least one such launch was running (event union)
            t2 = GemmTimer()
            t2.install()
            wl.step(args.warmup + args.steps + 4)       # settle the tile heuristic's stream count
            t2.reset()
            torch.cuda.synchronize()
            base = torch.cuda.Event(enable_timing=True)
            base.record()
            nrep = 3
            for r in range(nrep):
                wl.step(args.warmup + args.steps + 5 + r)
            busy, fl2 = t2.busy_union(kind, base)
            fam_dual = t2.families(1, PEAK_F16_MFMA / 3.0, base_event=base, steps=nrep) if kind == "f16x3" else None
            t2.remove()
            if busy > 0:
                dual = {"achieved": round(fl2 / (busy * 1e-3) / 1e12, 2), "frac": round(fl2 / (busy * 1e-3) / peak, 4),
                        "busy_ms_per_step": round(busy / nrep, 2), "steps": nrep, "families": fam_dual,
                        "note": "two student streams (the timed configuration): algorithmic flops of every launch of the kernel on "
                                "both streams / union of their [start, end] event intervals; per family the union of that "
                                "family's intervals (a launch's interval also spans what the other student's kernels took from it)"}
        traffic, tnote = pmc_traffic_per_launch(args.pmc_profile, prefix, gemm_mode) if (dataset, batch) == ("voc", 4) else \
            (None, "the committed PMC passes are of the VOC 4 img/GPU workload")
        # per family: fabric traffic of its kernels (PMC summary, per step) over its algorithmic bytes
        if traffic and fam_single:
            for fam, mb in pmc_family_traffic(args.pmc_profile).items():
                if fam in fam_single and fam_single[fam].get("algorithmic_mb_per_step"):
                    fam_single[fam]["traffic_mb_per_step"] = round(mb, 1)
                    fam_single[fam]["traffic_over_algorithmic"] = round(mb / fam_single[fam]["algorithmic_mb_per_step"], 2)
        # top level = the configuration the timed region runs in: two student streams unless --single-stream
        top = dual if dual is not None else single
        roof = {"bound": "mfma", "kernel": kname, "achieved": top["achieved"],
                "peak": round(peak / 1e12, 1), "unit": "TFLOP/s", "frac": round(top["achieved"] / (peak / 1e12), 4),
                "configuration": "two student streams (as timed)" if dual is not None else "one stream",
                "traffic": traffic, "traffic_source": tnote, "csrc_sha256": csrc_digest()[:16],
                "peak_definition": ("dense f16 MFMA peak 2500 TFLOP/s / 3 MFMA products per fp32-equivalent multiply-add; "
                                    f"executed MFMA rate = {3 * top['achieved']:.1f} of 2500 TFLOP/s (the same fraction)")
                if kind == "f16x3" else "f32 MFMA peak (v_mfma_f32_32x32x2_f32) 157.3 TFLOP/s",
                "algorithmic_bytes_per_launch": round(gbytes / max(gn, 1)),
                "traffic_over_algorithmic": (round(traffic * gn / gbytes, 2) if traffic else None),
                "families": top["families"],
                "dual_stream": dual, "single_stream": single,
                "launches_per_step": gn, "avg_launch_us": single["avg_launch_us"],
                "kernel_share_of_step": round((dual["busy_ms_per_step"] if dual is not None else gms) / ms, 3),
                "step_flop_per_img": {"reference_algorithm": flop_ref, "executed": flop_exec},
                "step_algorithmic_tflops": round(imgs_per_s / world * flop_exec / 1e12, 1),
                "note": "achieved / frac: every launch of the split GEMM (forward, data and weight gradients) in the configuration "
                        "the timed region runs in, HIP events on the launching streams during extra steps right after the timed "
                        "region; algorithmic flops = 2*M*N*K per launch (fp32-equivalent); `families` splits them (and adds the "
                        "split attention kernels, same peak); `single_stream` holds the per-launch figures rocprofv3 agrees with"}

    # the same workload on the exact-f32 MFMA kernels (DUPL_GEMM=f32): the number to read if the f16x3 split products are
    # not accepted as the reference's fp32 arithmetic
    exact = None
    if gemm_mode == "f16x3" and not args.no_exact_f32:
        engine.set_gemm_mode("f32")
        rx = wl.measure("exact-f32")
        engine.set_gemm_mode("f16x3")
        exact = {"value": round(rx["value"], 3), "unit": "img/s", "ms_per_step": round(rx["ms"], 2), "dtype": "f32",
                 "loss": round(rx["loss"], 5),
                 "note": "same workload, steps and warm-up with every GEMM / attention on v_mfma_f32_32x32x2_f32 (DUPL_GEMM=f32)"}

    listed_cfg = None
    l_ds, l_b, l_bb, l_name = CONFIG_BY_N.get(world, ("coco", 2, "deit_base_patch16_224", "COCO 2 img/GPU (no BASELINE entry for this N)"))
    if world > 1 and not args.no_listed and (dataset, batch, backbone) != (l_ds, l_b, l_bb):
        del wl
        torch.cuda.empty_cache()
        wL = Workload(args, world, rank, local, l_ds, l_b, l_bb, DEFAULT_N_ITER[l_ds])
        rL = wL.measure("listed-config")
        listed_cfg = {"value": round(rL["value"], 3), "unit": "img/s", "ms_per_step": round(rL["ms"], 2), "workload": wL.describe(),
                      "baseline_config": l_name, "global_batch": world * l_b, "img_per_gpu": l_b, "comm": rL["comm"],
                      "note": "the configuration BASELINE.json lists for this N (global batch fixed by the reference, i.e. "
                              f"{l_b} img/GPU): not a point of the weak-scaling curve that `value` draws"}
        wl = wL

    # the metric's second configuration ("COCO bs=8"): timed in the same run on one GPU, with its own ms_per_step
    second = None
    if world == 1 and not args.no_second and listed and (dataset, batch) == ("voc", 4):
        del wl
        torch.cuda.empty_cache()
        w2 = Workload(args, world, rank, local, "coco", 8, backbone, DEFAULT_N_ITER["coco"])
        r2 = w2.measure("second-config")
        second = {"value": round(r2["value"], 3), "unit": "img/s", "ms_per_step": round(r2["ms"], 2), "steps": args.steps,
                  "warmup": args.warmup, "workload": w2.describe(), "img_per_gpu": 8, "num_classes": 81,
                  "loss": round(r2["loss"], 5),
                  "note": "BASELINE.json metric, second half ('COCO bs=8'): MSCOCO2014 448^2, 81 classes, COCO schedule phase B2, "
                          "8 images on one GPU; same protocol as `value` (warm-up, barrier + synchronize, K timed steps)"}
        wl = w2

    cpu = None
    if rank == 0 and world == 1 and args.cpu_baseline == "auto":
        log("timing the CPU oracle on the host cores (bounded sample: warm-up + thread sweep + timed steps)")
        cpu = cpu_baseline(args, dataset, C)
        log(f"cpu baseline: {cpu}")
    if rank == 0:
        rec = {"metric": f"training img/s at 448^2, {'VOC' if dataset == 'voc' else 'COCO'} dual-student ViT-B/16, phase {phase} step",
               "value": round(imgs_per_s, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms, 2), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": DTYPE[gemm_mode], "data": "synthetic",
               "config": {"workload": (f"VOC2012 {args.size}^2" if dataset == "voc" else f"MSCOCO2014 {args.size}^2") +
                                      f" dual-student {backbone} + ms-CAM(1.0,0.5,1.5) + PAR + cross seg loss, phase {phase}, "
                                      f"{batch} img/GPU, DDP world_size={world}",
                          "baseline_config": d_cfg if listed else "custom (--dataset/--batch/--backbone)",
                          "global_batch": world * batch, "img_per_gpu": batch, "num_classes": C + 1,
                          "n_iter": n_iter, "parallelism": f"dp{world}", "student_streams": 1 if args.single_stream else 2,
                          "shared_scale1_encoder_pass": not args.no_share_encoder, "forward_gemm": gemm_mode,
                          "deterministic": os.environ.get("DUPL_DETERMINISTIC", "0") == "1",
                          "loss": round(res["loss"], 5)},
               # schema 2 (round 4): `value` = configs[1] (VOC, 4 img/GPU) on EVERY rank at every N -- a weak-scaling point, NOT the
               # configuration BASELINE.json lists for N > 1 (that one is `listed_config`); `second_config` = COCO 8 img/GPU at
               # N = 1; roofline.frac = the two-stream (timed) configuration
               "schema": 2, "value_config": "configs[1] per rank (weak scaling)" if listed else "custom",
               "comm": res["comm"], "listed_config": listed_cfg, "second_config": second, "exact_f32_path": exact,
               # f16x3 operand planes have fp16's range: sites whose operands could leave it (rigorous bounds from the
               # parameters, engine.RangeGuard) run on the exact-f32 kernels; 0 = the whole step ran on the split kernels
               "range_guard": (guard_main.summary() if gemm_mode == "f16x3" else None),
               "roofline": roof, "cpu_baseline": cpu}
        print(json.dumps(rec))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
