"""CPU, world_size = 2, gloo: the gradient exchange of the flat-buffer engine (dupl_amd/ddp.py).
Checks: parameter broadcast from rank 0, bucketed all-reduce == mean over ranks on the trainable range only
(frozen segment untouched), bucket partition covers the range exactly, and the DDP wrapper's hook protocol."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dupl_amd.model.model_dupl import siamese_network
        from dupl_amd.ddp import DistributedDataParallel
        from dupl_amd.synthetic import hash_normal
        torch.manual_seed(rank)      # different init per rank -> the broadcast must equalise
        m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
        ddp = DistributedDataParallel(m, bucket_mb=0.25)
        st = m.flat_storage
        gathered = [torch.empty_like(st.data) for _ in range(world)]
        dist.all_gather(gathered, st.data)
        same_params = all(torch.equal(gathered[0], t) for t in gathered)
        # buckets partition each student's trainable range exactly
        ok_buckets = True
        for s in (0, 1):
            lo, hi = st.trainable_range(s)
            bk = ddp.reducer.student_buckets(s)
            ok_buckets &= bk[0][0] == lo and bk[-1][1] == hi and all(a[1] == b[0] for a, b in zip(bk, bk[1:])) and len(bk) > 1
        # layer-granular plan (issued during the backward): disjoint, covers the trainable range exactly, ordered as
        # the backward finalises it (heads, blocks downwards, stem + norm)
        for s in (0, 1):
            lo, hi = st.trainable_range(s)
            plan = sorted(ddp.reducer.plan[s])
            ok_buckets &= plan[0][0] == lo and plan[-1][1] == hi and all(a[1] == b[0] for a, b in zip(plan, plan[1:]))
            trig = [p[2] for p in ddp.reducer.plan[s]]
            ok_buckets &= trig[0] == "heads" and trig[-2:] == ["stem", "stem"] and trig[1:-2] == sorted(trig[1:-2], reverse=True)
        # fake per-rank gradients; frozen segment gets a sentinel that must survive
        g_local = hash_normal(f"grad_rank{rank}", (st.grad.numel(),), std=1.0, seed=1)
        st.grad.copy_(g_local)
        for s in (0, 1):
            flo = s * st.student_numel
            st.grad[flo: flo + st.seg_bounds[0][1]] = 7.0 + rank
        # drive the hook protocol by hand: student 0's backward ends, then student 1's, then the engine callback
        # student 0: the per-layer events of a backward (heads, blocks 3..0 of the tiny backbone, stem), then the
        # post-backward hook; student 1: only the post-backward hook (e.g. a backward that reported nothing)
        for ev in ["heads", 3, 2, 1, 0, "stem"]:
            ddp.reducer.grad_ready(0, ev)
        n_layerwise = len(ddp.reducer._pending)
        for net in (m.branch1, m.branch2):
            ddp.reducer.reduce_student_async(net._student)
        ok_buckets &= n_layerwise >= len(ddp.reducer.plan[0]) and len(ddp.reducer._pending) > n_layerwise
        ddp.reducer.finish()
        ok_buckets &= not any(ddp.reducer._issued)
        expect = sum(hash_normal(f"grad_rank{r}", (st.grad.numel(),), std=1.0, seed=1) for r in range(world)) / world
        ok_mean, ok_frozen = True, True
        for s in (0, 1):
            lo, hi = st.trainable_range(s)
            ok_mean &= bool(torch.allclose(st.grad[lo:hi], expect[lo:hi], atol=1e-6))
            flo = s * st.student_numel
            ok_frozen &= bool((st.grad[flo: flo + st.seg_bounds[0][1]] == 7.0 + rank).all())
        q.put((rank, same_params, ok_buckets, ok_mean, ok_frozen))
    finally:
        dist.destroy_process_group()


def test_grad_allreduce_world2_gloo():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same_params, ok_buckets, ok_mean, ok_frozen in res:
        assert same_params, "parameter broadcast from rank 0 failed"
        assert ok_buckets, "buckets do not partition the trainable range"
        assert ok_mean, "all-reduce result != mean over ranks"
        assert ok_frozen, "frozen segment (pos_embed / head) was touched by the exchange"


def _hist_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dupl_amd.utils import evaluate
        rng = np.random.RandomState(5)
        gts = [rng.randint(0, 23, size=(7, 9)) for _ in range(6)]
        preds = [rng.randint(0, 21, size=(7, 9)) for _ in range(6)]
        # round-robin shard of the "val set" per rank (tools/eval_seg_coco_ddp.py:239-245), per-rank histogram, SUM
        h = torch.zeros((21, 21), dtype=torch.int64)
        for i in range(rank, 6, world):
            h += torch.from_numpy(evaluate._fast_hist(gts[i].flatten(), preds[i].flatten(), 21))
        dist.all_reduce(h, op=dist.ReduceOp.SUM)
        whole = evaluate.scores(gts, preds, 21)
        mine = evaluate.scores_from_hist(h.numpy())
        q.put((rank, abs(mine["miou"] - whole["miou"]) < 1e-12 and abs(mine["pAcc"] - whole["pAcc"]) < 1e-12))
    finally:
        dist.destroy_process_group()


def test_eval_hist_exchange_world2_gloo():
    """The evaluation path's only exchange: per-rank confusion matrices summed over ranks == the whole-set scores."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_hist_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(ok for _, ok in res)
