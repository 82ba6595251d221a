#!/usr/bin/env python
"""How long the HOST needs to issue one training step (Python + ctypes + torch allocator, no waiting for the GPU): the step is
launch-bound wherever this approaches the GPU time of the step.  Issues a few steps back to back without synchronising and times
each call; then the same with the GPU drained before every step (the GPU time is then the synchronised wall time).

    python tools/host_issue_time.py [--batch 2]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dataset", default="voc")
    a = ap.parse_args()
    sys.argv = [sys.argv[0], "--steps", "3", "--warmup", "2"]
    args = bench.parse()
    w = bench.Workload(args, 1, 0, 0, a.dataset, a.batch, "deit_base_patch16_224", 5000)
    for i in range(4):
        w.step(i)
    torch.cuda.synchronize()
    issue, total = [], []
    for i in range(6):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        w.step(10 + i)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        issue.append((t1 - t0) * 1e3)
        total.append((t2 - t0) * 1e3)
    t0 = time.perf_counter()
    for i in range(10):
        w.step(20 + i)
    torch.cuda.synchronize()
    steady = (time.perf_counter() - t0) / 10 * 1e3
    print(f"{a.dataset} {a.batch} img/GPU: host issue time per step {min(issue):.2f} ms (median {sorted(issue)[len(issue) // 2]:.2f}); "
          f"drained step wall {min(total):.2f} ms; steady state {steady:.2f} ms per step")


if __name__ == "__main__":
    main()
