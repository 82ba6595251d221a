#!/usr/bin/env python
"""Per-shape table of the split-GEMM / split-attention launches of one training step (one stream, HIP events on the launch
stream): which (family, M, N, K, epilogue) a step issues, how often, how long each takes (minimum over the repeated steps) and what
fraction of the f16 / 3 MFMA peak that is.  The bench line's `roofline.families` is the sum of these rows.

    python tools/gemm_shapes.py [--batch 4] [--dataset voc] [--steps 3] [--dual]  > gpurun_out/shapes.txt
"""
import argparse
import os
import sys
from collections import OrderedDict

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

PEAK = 2500e12 / 3.0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dataset", default="voc")
    ap.add_argument("--backbone", default="deit_base_patch16_224")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--dual", action="store_true", help="two student streams (event intervals then span co-running kernels)")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--steps", str(a.steps), "--warmup", "2"] + ([] if a.dual else ["--single-stream"])
    args = bench.parse()
    from dupl_amd import ops
    w = bench.Workload(args, 1, 0, 0, a.dataset, a.batch, a.backbone, 5000)
    for i in range(2):
        w.step(i)
    torch.cuda.synchronize()

    rec = []          # (key, flops, e0, e1, stream) in issue order

    def ev(key, flops, fn):
        s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        r = fn()
        e1.record(s)
        rec.append((key, flops, e0, e1, s.cuda_stream))
        return r

    o16, oaf, oab, oafs, owg = ops.linear16, ops.attention_fwd16, ops.attention_bwd16, ops.attention_fwd16_segs, ops.wgrad16_group

    def t16(x, W, *p, **kw):
        akm, bkm = kw.get("a_kmajor", False), kw.get("b_kmajor", False)
        M = x.cols if akm else x.rows
        N = W.cols if bkm else W.rows
        K = getattr(x, "valid_rows", x.rows) if akm else x.cols
        acc = kw.get("accumulate", False)
        fam = "wgrad" if acc and (akm or not bkm) else ("dgrad" if kw.get("alpha") is not None else ("fwd_f1" if getattr(x, "exp", 0) else "fwd_f0"))
        epi = "+".join(n for n, on in (("bias", (p[0] if p else kw.get("bias")) is not None), ("gelu", kw.get("gelu")), ("relu", kw.get("relu")),
                                       ("res", kw.get("res") is not None), ("pre", kw.get("store_pre") is not None),
                                       ("dgelu", kw.get("dgelu_of") is not None), ("rmask", kw.get("relumask_of") is not None),
                                       ("f32", kw.get("want_f32", True) or kw.get("out") is not None),
                                       ("p16", kw.get("want16", False) or kw.get("out16") is not None), ("acc", acc),
                                       ("amax", kw.get("amax_for_next", False)), ("crows", bool(kw.get("c_rows", 0)))) if on)
        return ev((fam, M, N, K, epi), 2.0 * M * N * K, lambda: o16(x, W, *p, **kw))

    def taf(qkv16, B, N, H, hd, scale, *p, **kw):
        return ev(("attn_fwd", B, N, H, ""), 4.0 * B * H * N * N * hd, lambda: oaf(qkv16, B, N, H, hd, scale, *p, **kw))

    def tab(qkv16, out, dout, lse, B, N, H, hd, scale, *p, **kw):
        return ev(("attn_bwd", B, N, H, ""), 10.0 * B * H * N * N * hd, lambda: oab(qkv16, out, dout, lse, B, N, H, hd, scale, *p, **kw))

    def tafs(qkv16, segs, H, hd, scale, *p, **kw):
        return ev(("attn_fwd_segs", tuple((s[1], s[2]) for s in segs), 0, H, ""), sum(4.0 * s[1] * H * s[2] * s[2] * hd for s in segs),
                  lambda: oafs(qkv16, segs, H, hd, scale, *p, **kw))

    def twg(items, **kw):
        fl = sum(2.0 * d.cols * x.cols * getattr(d, "valid_rows", d.rows) for d, x, _, _ in items)
        return ev(("wgrad_group", len(items), items[0][0].rows, 0, ""), fl, lambda: owg(items, **kw))

    ops.linear16, ops.attention_fwd16, ops.attention_bwd16, ops.attention_fwd16_segs, ops.wgrad16_group = t16, taf, tab, tafs, twg
    per_step = []
    occupancy = []
    for i in range(a.steps):
        rec.clear()
        base = torch.cuda.Event(enable_timing=True)
        base.record(torch.cuda.current_stream())
        t_host = __import__("time").perf_counter()
        w.step(10 + i)
        torch.cuda.synchronize()
        wall = (__import__("time").perf_counter() - t_host) * 1e3
        per_step.append([(k, f, e0.elapsed_time(e1)) for k, f, e0, e1, _ in rec])
        # the step's timeline of the recorded (MFMA-heavy) launches: how long 0 / 1 / 2 of them were in flight
        iv = sorted((base.elapsed_time(e0), base.elapsed_time(e1), st) for _, _, e0, e1, st in rec)
        pts = sorted([(a_, 1) for a_, _, _ in iv] + [(b_, -1) for _, b_, _ in iv])
        depth, last, t_depth = 0, 0.0, {0: 0.0, 1: 0.0, 2: 0.0}
        for t_, d_ in pts:
            t_depth[min(depth, 2)] += t_ - last
            last, depth = t_, depth + d_
        t_depth[0] += max(0.0, wall - last)
        per_stream = {}
        for a_, b_, st in iv:
            per_stream[st] = per_stream.get(st, 0.0) + (b_ - a_)
        occupancy.append((wall, t_depth[0], t_depth[1], t_depth[2], sorted(per_stream.values())))
    n = len(per_step[0])
    assert all(len(p) == n for p in per_step)
    rows = OrderedDict()
    for j in range(n):
        k, f = per_step[0][j][0], per_step[0][j][1]
        ms = min(p[j][2] for p in per_step)
        r = rows.setdefault(k, [0, 0.0, 0.0])
        r[0] += 1
        r[1] += ms
        r[2] += f
    tot = sum(r[1] for r in rows.values())
    print(f"# {a.dataset} {a.batch} img/GPU {a.backbone}, {'two streams' if a.dual else 'one stream'}; {n} launches, {tot:.2f} ms per step")
    for wall, t0, t1, t2, ps in occupancy:
        print(f"# step wall {wall:.2f} ms: no recorded launch in flight {t0:.2f} ms, one {t1:.2f} ms, two or more {t2:.2f} ms; "
              f"per stream sum of intervals {[round(v, 2) for v in ps]}")
    print(f"{'family':14s} {'M':>22s} {'N':>6s} {'K':>6s} {'n':>4s} {'us/launch':>10s} {'ms/step':>8s} {'TF/s-eq':>8s} {'frac':>6s}  epilogue")
    for k, (c, ms, fl) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
        tf = fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        print(f"{k[0]:14s} {str(k[1]):>22s} {k[2]:>6d} {k[3]:>6d} {c:>4d} {ms / c * 1e3:>10.1f} {ms:>8.3f} {tf:>8.1f} {tf * 1e12 / PEAK:>6.3f}  {k[4]}")


if __name__ == "__main__":
    main()
