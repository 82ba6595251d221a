#!/usr/bin/env python
"""What a kernel family costs the TIMED configuration (two student streams): every launch of one family is issued TWICE (the extra one
first, without side effects on the scale ring) and the step is timed again -- the increase is that family's marginal cost in the step
as it is actually scheduled, which the per-launch event intervals of bench.py cannot give (they overlap the other student's kernels)
and a one-stream trace cannot either (it serialises the students).  Results stay finite (accumulating launches double a gradient);
they are not a training run.

    python tools/marginal_cost.py [--batch 4] [--dataset voc] [--steps 20] [--families fwd,attn_fwd,...] > gpurun_out/marginal.txt
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

FAMILIES = ("none", "fwd", "dgrad", "wgrad", "attn_fwd", "attn_bwd", "ln_fwd")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dataset", default="voc")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--families", default=",".join(FAMILIES))
    ap.add_argument("--single-stream", action="store_true")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--steps", str(a.steps), "--warmup", "3"] + (["--single-stream"] if a.single_stream else [])
    args = bench.parse()
    from dupl_amd import ops
    w = bench.Workload(args, 1, 0, 0, a.dataset, a.batch, "deit_base_patch16_224", 5000)
    o16, oaf, oafs, oab, owg, oln = (ops.linear16, ops.attention_fwd16, ops.attention_fwd16_segs, ops.attention_bwd16, ops.wgrad16_group,
                                    ops.layernorm_fwd16)
    state = {"fam": "none", "dup": 0}

    def fam16(x, W, kw):
        akm, bkm, acc = kw.get("a_kmajor", False), kw.get("b_kmajor", False), kw.get("accumulate", False)
        if acc and (akm or not bkm):
            return "wgrad"
        return "dgrad" if kw.get("alpha") is not None else "fwd"

    def t16(x, W, *p, **kw):
        if fam16(x, W, kw) == state["fam"]:
            k2 = dict(kw)
            k2["amax_for_next"] = False
            if not kw.get("accumulate", False):          # the extra launch writes outputs of its own
                for k in ("out", "out16", "store_pre"):
                    if k2.get(k) is not None:
                        v = k2[k]
                        k2[k] = torch.empty_like(v) if torch.is_tensor(v) else ops.split16_empty(v.rows, v.cols, v.planes.device, v.exp)
            o16(x, W, *p, **k2)
            state["dup"] += 1
        return o16(x, W, *p, **kw)

    def taf(*p, **kw):
        if state["fam"] == "attn_fwd":
            oaf(*p, **kw)
            state["dup"] += 1
        return oaf(*p, **kw)

    def tafs(*p, **kw):
        if state["fam"] == "attn_fwd":
            oafs(*p, **kw)
            state["dup"] += 1
        return oafs(*p, **kw)

    def tab(*p, **kw):
        if state["fam"] == "attn_bwd":
            k2 = dict(kw)
            k2["amax_for_next"] = False
            oab(*p, **k2)
            state["dup"] += 1
        return oab(*p, **kw)

    def twg(items, **kw):
        if state["fam"] == "wgrad":
            owg(items, **kw)
            state["dup"] += 1
        return owg(items, **kw)

    def tln(*p, **kw):
        if state["fam"] == "ln_fwd":
            oln(*p, **kw)
            state["dup"] += 1
        return oln(*p, **kw)

    ops.linear16, ops.attention_fwd16, ops.attention_fwd16_segs, ops.attention_bwd16, ops.wgrad16_group, ops.layernorm_fwd16 = t16, taf, tafs, tab, twg, tln

    def timed(fam):
        state["fam"], state["dup"] = fam, 0
        for i in range(3):
            w.step(i)
        torch.cuda.synchronize()
        state["dup"] = 0
        t0 = time.perf_counter()
        for i in range(a.steps):
            out = w.step(3 + i)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / a.steps * 1e3
        return ms, state["dup"] / a.steps, float(out["loss"].sum().item())

    fams = [f for f in a.families.split(",") if f]
    base = None
    print(f"# {a.dataset} {a.batch} img/GPU, {'one stream' if a.single_stream else 'two streams'}, {a.steps} steps per row; marginal = ms per step with every launch "
          f"of the family issued twice - baseline")
    print(f"{'family':10s} {'ms/step':>9s} {'marginal':>9s} {'extra launches/step':>20s}  loss")
    for f in fams + ["none"]:
        ms, dup, loss = timed(f)
        if f == "none" and base is None:
            base = ms
        print(f"{f:10s} {ms:9.2f} {ms - (base if base is not None else ms):9.2f} {dup:20.0f}  {loss:.4f}", flush=True)


if __name__ == "__main__":
    main()
