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


class _Replay(torch.autograd.Function):
    """An autograd node whose backward runs a closure: stands in for a student's _NetworkFn so that the DDP wrapper's hook
    protocol (grad-ready events, post-backward hook, the engine's final callback) is driven by a REAL autograd pass."""

    @staticmethod
    def forward(ctx, x, fn):
        ctx.fn = fn
        return x * 1.0

    @staticmethod
    def backward(ctx, g):
        ctx.fn()
        return g, None


def _worker8(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dupl_amd.model.model_dupl import siamese_network
        from dupl_amd.ddp import DistributedDataParallel
        from dupl_amd.synthetic import hash_normal
        torch.manual_seed(100 + rank)
        m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
        ddp = DistributedDataParallel(m, bucket_mb=0.05, blocks_per_bucket=1)
        st, red = m.flat_storage, ddp.reducer
        ok = {"world": red.world == world}
        # the plan tiles each student's trainable range exactly once, at this world size too
        cover = torch.zeros(st.grad.numel(), dtype=torch.int32)
        for s in (0, 1):
            for lo, hi, _ in red.plan[s]:
                cover[lo:hi] += 1
        want = torch.zeros_like(cover)
        for s in (0, 1):
            lo, hi = st.trainable_range(s)
            want[lo:hi] = 1
        ok["plan_tiles_once"] = bool(torch.equal(cover, want))
        events = ["heads"] + list(reversed(range(st.cfg.depth))) + ["stem"]
        nets = (m.branch1, m.branch2)

        def fill(tag):
            g = hash_normal(f"{tag}_rank{rank}", (st.grad.numel(),), std=1.0, seed=3)
            st.grad.copy_(g)
            for s in (0, 1):
                flo = s * st.student_numel
                st.grad[flo: flo + st.seg_bounds[0][1]] = 5.0 + rank

        def expect(tag):
            return sum(hash_normal(f"{tag}_rank{r}", (st.grad.numel(),), std=1.0, seed=3) for r in range(world)) / world

        def check(tag):
            e = expect(tag)
            good = True
            for s in (0, 1):
                lo, hi = st.trainable_range(s)
                good &= bool(torch.allclose(st.grad[lo:hi], e[lo:hi], atol=2e-6))
                flo = s * st.student_numel
                good &= bool((st.grad[flo: flo + st.seg_bounds[0][1]] == 5.0 + rank).all())
            return good

        seen = {}

        def student_backward(net, last: bool):
            """What _NetworkFn.backward does around engine.network_backward (model_dupl.py): events only during the last
            pending backward of the student, then the post-backward hooks."""
            def run():
                if net._grad_ready_hooks and net._live_graphs <= 1:
                    for ev in events:
                        for hook in net._grad_ready_hooks:
                            hook(net, ev)
                else:
                    seen.setdefault("issued_early", 0)
                    seen["issued_early"] += len(red._issued[net._student])      # a first-of-two backward issues nothing of ITS student
                net._live_graphs = max(0, net._live_graphs - 1)
                for hook in net._post_backward_hooks:
                    hook(net)
            return run

        # ---- phase B: one forward per student
        fill("B")
        red.pop_stats()
        a = torch.zeros(1, requires_grad=True)
        for net in nets:
            net._live_graphs = 1
        loss = sum(_Replay.apply(a, student_backward(net, True)) for net in nets)
        loss.sum().backward()
        stB = red.pop_stats()
        ok["B_mean"] = check("B")
        n_trainable = sum(st.trainable_range(s)[1] - st.trainable_range(s)[0] for s in (0, 1))
        ok["B_bytes_once"] = stB["allreduce_bytes"] == 4 * n_trainable       # every element exchanged exactly once
        ok["B_clean"] = not red._pending and not any(red._issued) and not ddp._touched and not ddp._reduced
        # ---- phase C: TWO forwards per student (train_final_voc.py:291-295); the backward of the first one only accumulates
        fill("C")
        for net in nets:
            net._live_graphs = 2
        parts = []
        for net in nets:
            parts.append(_Replay.apply(a, student_backward(net, False)))
            parts.append(_Replay.apply(a, student_backward(net, True)))
        sum(parts).sum().backward()
        stC = red.pop_stats()
        ok["C_mean"] = check("C")
        ok["C_bytes_once"] = stC["allreduce_bytes"] == 4 * n_trainable
        ok["C_nothing_issued_by_first_backward"] = seen.get("issued_early", 0) == 0
        ok["C_clean"] = not red._pending and not any(red._issued) and all(n._live_graphs == 0 for n in nets)
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_grad_bucket_bookkeeping_world8_gloo():
    """the bucket plan and the DDP wrapper's hook protocol at world 8 (the node size of configs[4]), driven
    by a real autograd pass per step: phase B (one forward per student) and phase C (two forwards per student: the first
    backward must not exchange anything).  Every trainable element is all-reduced exactly once per step and ends as the mean
    over the 8 ranks; the frozen segment is never touched."""
    world = 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=600) for _ in range(world)]
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok.values()), (rank, ok)


def _consumer_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from dupl_amd.model.model_dupl import siamese_network
        from dupl_amd.ddp import DistributedDataParallel
        from dupl_amd.synthetic import hash_normal
        m = siamese_network("tiny_test", num_classes=21, pretrained=False, aux_layer=-3)
        ddp = DistributedDataParallel(m)
        red, st = ddp.reducer, m.flat_storage
        ok = {}
        st.grad.copy_(hash_normal(f"grad_rank{rank}", (st.grad.numel(),), std=1.0, seed=3))
        taken = []                       # (student, lo, hi, number of events reported so far)
        n_events = [0]

        def consumer(student, lo, hi, inv):
            taken.append((student, lo, hi, n_events[0]))
            st.grad[lo:hi].mul_(inv)     # what dupl_adamw's grad_scale does: the mean, written back
            return True
        red.consumer = consumer
        issued_at = {}
        events = ["heads", 2, 0, "stem"]      # tiny backbone: depth 4, two blocks per bucket
        order = [(0, e) for e in events[:2]] + [(1, e) for e in events[:3]] + [(0, e) for e in events[2:]] + [(1, events[3])]
        for s, ev in order:
            n_events[0] += 1
            before = {(e[2], e[3], e[4]) for e in red._pending}
            red.grad_ready(s, ev)
            for e in red._pending:
                issued_at.setdefault((e[2], e[3], e[4]), n_events[0])
            # pieces of THIS student issued at earlier events are gone from the pending list, the other student's are untouched
            ok[f"lag_{s}_{ev}"] = all(k[0] != s for k in before & {(e[2], e[3], e[4]) for e in red._pending})
        n_left = len(red._pending)
        ok["last_pieces_wait_for_finish"] = n_left >= 2
        n_events[0] += 1
        red.finish()
        ok["all_taken"] = not red._pending and not any(red._issued)
        # every piece was consumed exactly once, strictly after the event that issued it
        ok["once"] = len({t[:3] for t in taken}) == len(taken) == sum(len(p) for p in red.plan)
        ok["after_issue"] = all(t[3] > issued_at[t[:3]] for t in taken)
        expect = sum(hash_normal(f"grad_rank{r}", (st.grad.numel(),), std=1.0, seed=3) for r in range(world)) / world
        for s in (0, 1):
            lo, hi = st.trainable_range(s)
            ok[f"mean_{s}"] = bool(torch.allclose(st.grad[lo:hi], expect[lo:hi], atol=1e-6))
        # a consumer that declines (returns False) leaves the averaging to the reducer
        st.grad.copy_(hash_normal(f"grad_rank{rank}", (st.grad.numel(),), std=1.0, seed=3))
        red.consumer = lambda *a: False
        for s, ev in order:
            red.grad_ready(s, ev)
        red.finish()
        for s in (0, 1):
            lo, hi = st.trainable_range(s)
            ok[f"declined_mean_{s}"] = bool(torch.allclose(st.grad[lo:hi], expect[lo:hi], atol=1e-6))
        q.put((rank, ok))
    finally:
        dist.destroy_process_group()


def test_reduced_pieces_are_consumed_one_event_later_world2_gloo():
    """ddp.GradReducer.consumer (round 6: the optimiser's update rides in the exchange): a piece is handed over exactly once, only
    after a LATER event of its own student (or at finish()), never touching the other student's pending pieces, and the buffer
    ends as the mean over ranks whether the consumer scales (returns True) or declines."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_consumer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok in res:
        assert all(ok.values()), (rank, ok)
