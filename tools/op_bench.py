"""Micro-timings of the small (non-GEMM) kernels of the step at their step shapes, through dupl_amd.ops (torch events on the
current stream, 200 launches each).  Usage: python tools/op_bench.py [ln_bwd] [ln_fwd] [split] [attn_bwd]"""
import gc
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from dupl_amd import ops  # noqa: E402


def timeit(fn, n=200, warm=20):
    gc.collect()            # a generation-2 collection inside the timed loop shows up as a 40 ms outlier
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / n


def main():
    which = set(sys.argv[1:]) or {"ln_bwd", "ln_fwd", "split", "attn_bwd", "multi", "cam"}
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    rows, D = 3140, 768
    x = torch.randn(rows, D, generator=g).to(dev)
    dy = (torch.randn(rows, D, generator=g) * 1e-5).to(dev)
    dres = (torch.randn(rows, D, generator=g) * 1e-5).to(dev)
    gamma = torch.ones(D, device=dev)
    beta = torch.zeros(D, device=dev)
    mean, rstd = x.mean(1), (x.var(1, unbiased=False) + 1e-6).rsqrt()
    dg, db = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    if "ln_bwd" in which:
        for rpw in (8, 4, 2, 1):
            ops.LNB_ROWS_PER_WAVE = rpw
            t = timeit(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, two_stage=False))
            t2 = timeit(lambda: ops.layernorm_bwd(dy, x, gamma, mean, rstd, dg, db, dres=dres, two_stage=True))
            print(f"layernorm_bwd {rows}x{D} rows/wave {rpw}: atomics {t:.1f} us ({4 * rows * D * 4 / t / 1e6:.2f} TB/s), "
                  f"two-stage {t2:.1f} us ({4 * rows * D * 4 / t2 / 1e6:.2f} TB/s)")
        ops.LNB_ROWS_PER_WAVE = 0
    if "ln_fwd" in which:
        for r in (3140, 6280, 15696):
            xx = torch.randn(r, D, generator=g).to(dev)
            t = timeit(lambda: ops.layernorm_fwd16(xx, gamma, beta, 1e-6))
            print(f"layernorm_fwd16 {r}x{D}: {t:.1f} us")
    if "split" in which:
        for (r, c) in ((3140, 768), (3140, 3072), (3140, 2304)):
            xx = (torch.randn(r, c, generator=g) * 1e-5).to(dev)
            t = timeit(lambda: ops.split_prepare(xx, scaled=True, want_rm=True, want_T=True, rows_pad=3168))
            t2 = timeit(lambda: ops.split_prepare(xx, scaled=False, want_rm=False, want_T=True, rows_pad=3168))
            print(f"split_prepare {r}x{c}: scaled rm+T {t:.1f} us (amax + split), T only {t2:.1f} us")
    if "multi" in which:
        xs = [torch.randn(3140, c, generator=g).to(dev) for c in (3072, 768, 768, 768)]
        ws = [torch.randn(r, c, generator=g).to(dev) for r, c in ((768, 3072), (3072, 768), (768, 768), (2304, 768))]
        items = [(x, False, True, 3168) for x in xs] + [(w, False, True, w.shape[0]) for w in ws]

        def singles():
            for x, rm, T, rp in items:
                ops.split_prepare(x, scaled=False, want_rm=rm, want_T=T, rows_pad=rp)
        t1 = timeit(singles, n=100)
        t2 = timeit(lambda: ops.split_prepare_multi(items), n=100)
        t3 = timeit(lambda: ops.split_prepare_multi(items[:4]), n=100)
        print(f"block operands (4 x^T + 4 W^T): 8 launches {t1:.1f} us, one multi launch {t2:.1f} us; x^T only, one launch {t3:.1f} us")
        t4 = timeit(lambda: ops.split_prepare_multi(items[4:]), n=100)
        print(f"  W^T only, one launch {t4:.1f} us")
        for k in range(8):
            tk = timeit(lambda: ops.split_prepare_multi(items[k:k + 1]), n=100)
            ts = timeit(lambda: ops.split_prepare(items[k][0], scaled=False, want_rm=False, want_T=True, rows_pad=items[k][3]), n=100)
            print(f"  item {k} {tuple(items[k][0].shape)}: multi(1) {tk:.1f} us, single {ts:.1f} us")
        for m in (5, 6, 7):
            tm = timeit(lambda: ops.split_prepare_multi(items[:m]), n=100)
            print(f"  first {m} items: {tm:.1f} us")
    if "cam" in which:
        for C in (20, 80):
            B, H, W = 4, 448, 448
            sizes = [(28, 28), (14, 14), (42, 42)]
            lows = [torch.randn(2 * B * (1 + h * w), C, generator=g).to(dev) for h, w in sizes]
            for nb in (384, 512, 768, 1024, 1536, 2048, 4096):
                t = timeit(lambda: ops.cam_fuse(lows, sizes, B, C, H, W, 1, C, band_blocks=nb), n=100)
                print(f"cam_fuse C={C} band kernel, {nb} blocks aimed at: {t:.1f} us = {B * C * H * W * 4 / t / 1e6:.2f} TB/s of output (incl. the min/max init launch)")
    if "attn_bwd" in which:
        B, N, H, hd = 4, 785, 12, 64
        qkv = torch.randn(B * N, 3 * H * hd, generator=g).to(dev)
        dout = (torch.randn(B * N, H * hd, generator=g) * 1e-5).to(dev)
        qkv16 = ops.split16(qkv)
        out = torch.empty(B * N, H * hd, device=dev)
        lse = ops.attention_fwd16(qkv16, B, N, H, hd, hd ** -0.5, need_lse=True, out=out)
        t = timeit(lambda: ops.attention_bwd16(qkv16, out, dout, lse, B, N, H, hd, hd ** -0.5), n=100)
        print(f"attention_bwd16 B{B} N{N}: {t:.1f} us (delta + dq + dv + dk + the dout split)")


if __name__ == "__main__":
    main()
