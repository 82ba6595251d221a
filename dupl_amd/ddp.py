"""Data-parallel gradient synchronisation for the flat-buffer engine.

Reference mechanism (train_final_voc.py:108-110,153-155): torch DistributedDataParallel over NCCL with
find_unused_parameters=True -- a parameter broadcast at construction, a bucketed gradient all-reduce
(sum, then / world) driven by autograd hooks during loss.backward(), plus a per-step "used parameter"
bitmap all-reduce that exists only because `encoder.head` is never used.

Here the parameters of both students live in one flat buffer (engine.FlatStorage) and their gradients in a
second one, so the exchange step is: one RCCL broadcast of the parameter buffer at construction, and per
student ONE contiguous gradient range [backbone|norm|cls|decoder] all-reduced in a few large buckets
(xGMI is point-to-point: few big messages beat many small ones).  The frozen segment (pos_embed, head)
is never reduced, so no unused-parameter detection is needed.  Overlap: the hand-written backward reports when
a piece of a student's gradient is final -- [cls | decoder] after the heads, then the transformer blocks two
at a time as the backward walks down, the stem and the LayerNorm segment at the end (FlatStorage.grad_buckets)
-- and each piece is all-reduced asynchronously right then, on RCCL's stream, ordered after the student stream
that produced it.  Only the last ~60 MB bucket per student is exposed; at 2 GPUs (one xGMI link per pair,
738.8 MB per step) that is the difference between ~15 ms and ~2 ms of un-hidden communication per 129 ms step.
The optimiser waits at the end of backward through an autograd-engine callback, where stock DDP finalises.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .engine import SEG_BACKBONE, SEG_DECODER, FlatStorage


class GradReducer:
    """Bucketed all-reduce (sum then / world) over the trainable gradient range of each student.

    Round 6: a reduced piece can be CONSUMED as soon as its all-reduce has completed -- `consumer(student, lo, hi, 1 / world)`
    (the optimiser, PolyWarmupAdamW.begin_step) waits for the piece's collective on the student's stream, folds the 1 / world
    into its gradient read (dupl_adamw's grad_scale: the same single rounding the scale launch made, written back, so .grad holds
    the mean like torch DDP's) and updates that range right there, under the other student's backward.  A piece is taken in ONE
    event after it was issued (its all-reduce has had a whole bucket's backward to finish, so the wait does not stall the stream);
    only the last bucket's reduce + update and the stem / LayerNorm ranges are left for the end of the pass.  Without a consumer
    the pieces are waited for and scaled at finish(), as before."""

    def __init__(self, store: FlatStorage, process_group=None, bucket_mb: float = 128.0, blocks_per_bucket: int = 2):
        self.store = store
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self._pending: List = []       # [work, tensor, student, lo, hi] in issue order
        self.consumer = None           # callable(student, lo, hi, inv) -> bool (True: it scaled + used grad[lo:hi])
        # per-step exchange accounting (bench.py: comm_exposed_ms / allreduce_bytes): bytes handed to all_reduce since the
        # last pop_stats(), and -- when `profile` is on -- how long the stream that finalises the step is blocked in the
        # waits of finish() AFTER all compute of the step has drained (what the overlap failed to hide), plus how long the student
        # streams stood in the waits of the pieces taken in during the backward (`stall`)
        self.profile = False
        self._bytes = 0
        self._calls = 0
        self._exposed = []       # (event, event) pairs on CUDA, float ms on CPU tensors
        self._stall = []
        # layer-granular plan: per student [(lo, hi, trigger event)], in the order the backward finalises them
        self.plan = [store.grad_buckets(s, blocks_per_bucket) for s in range(store.n_students)]
        self._issued = [set() for _ in range(store.n_students)]

    def student_buckets(self, student: int):
        lo, hi = self.store.trainable_range(student)
        out = []
        while lo < hi:
            n = min(self.bucket_elems, hi - lo)
            out.append((lo, lo + n))
            lo += n
        return out

    def broadcast_parameters(self, src: int = 0):
        if self.world > 1:
            dist.broadcast(self.store.data, src=src, group=self.pg)
            # a c10d collective does not bump the tensor's version counter: tell the fp16 operand-plane cache
            # (FlatStorage.ensure_w16 / w16T) that the parameters changed.  Every non-autograd writer of store.data
            # (raw-pointer kernels, collectives) must do the same.  rewritten=True: a wholesale rewrite, so the range guard
            # re-checks synchronously before the next forward instead of `period` steps later.
            self.store.mark_dirty(rewritten=True)

    def _issue(self, student: int, lo: int, hi: int):
        """all-reduce grad[lo:hi] in pieces of at most bucket_elems (returns immediately)."""
        while lo < hi:
            n = min(self.bucket_elems, hi - lo)
            t = self.store.grad[lo:lo + n]
            work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending.append([work, t, student, lo, lo + n])
            self._bytes += 4 * n
            self._calls += 1
            lo += n

    def _retire(self, entry):
        """The piece's collective is complete in the order of the current stream: hand it to the consumer, or average it."""
        _, t, student, lo, hi = entry
        inv = 1.0 / self.world
        if self.consumer is not None and self.consumer(student, lo, hi, inv):
            return
        if t.is_cuda:
            from . import ops
            ops.scale_(t, inv)
        else:
            t.mul_(inv)

    def _take_in(self, student: int, upto: int):
        """Wait for (stream-level on RCCL, host-level on gloo) and retire this student's pieces among the first `upto` pending."""
        keep, mine = [], []
        for i, e in enumerate(self._pending):
            (mine if (i < upto and e[2] == student) else keep).append(e)
        if not mine:
            return
        self._pending = keep
        for e in mine:
            if self.profile and e[1].is_cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                e[0].wait()
                e1.record()
                self._stall.append((e0, e1))
            else:
                e[0].wait()
            self._retire(e)

    def grad_ready(self, student: int, event):
        """network_backward reports `event` ("heads", a block index, "stem"): issue the buckets it finalises; with a consumer, take
        in the pieces of this student that were issued at EARLIER events (on the current stream = the student's)."""
        if self.world == 1:
            return
        before = len(self._pending)
        for idx, (lo, hi, trig) in enumerate(self.plan[student]):
            if trig == event and idx not in self._issued[student]:
                self._issued[student].add(idx)
                self._issue(student, lo, hi)
        if self.consumer is not None:
            self._take_in(student, before)

    def reduce_student_async(self, student: int):
        """Issue every bucket of one student that has not been issued yet (returns immediately)."""
        if self.world == 1:
            return
        for idx, (lo, hi, _) in enumerate(self.plan[student]):
            if idx not in self._issued[student]:
                self._issued[student].add(idx)
                self._issue(student, lo, hi)

    def finish(self):
        """Wait for the pieces still pending (stream-level on RCCL, host-level on gloo) and retire them."""
        for s in self._issued:
            s.clear()
        if not self._pending:
            return
        cuda = self._pending[0][1].is_cuda
        if cuda and (self.profile or self.consumer is not None):
            # the consumer updates parameters on THIS stream: everything the students still have queued comes first (and in profile
            # mode the interval below is then communication only)
            self.store.wait_streams()
        if self.profile:
            import time
            if cuda:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            else:
                t0 = time.perf_counter()
        for e in self._pending:
            e[0].wait()
        if self.profile:
            if cuda:
                e1.record()
                self._exposed.append((e0, e1))
            else:
                self._exposed.append((time.perf_counter() - t0) * 1e3)
        pend, self._pending = self._pending, []
        for e in pend:
            self._retire(e)

    def pop_stats(self):
        """{allreduce_bytes, allreduce_calls, exposed_ms (list, one per finish() while profile was on), stall_ms (sum of the student
        streams' waits on pieces taken in during the backward)} since the last call; synchronises the device when CUDA events are
        pending."""
        def ms(lst):
            out = []
            for e in lst:
                if isinstance(e, tuple):
                    e[1].synchronize()
                    out.append(e[0].elapsed_time(e[1]))
                else:
                    out.append(e)
            return out
        out = {"allreduce_bytes": self._bytes, "allreduce_calls": self._calls, "exposed_ms": ms(self._exposed),
               "stall_ms": float(sum(ms(self._stall)))}
        self._bytes, self._calls, self._exposed, self._stall = 0, 0, [], []
        return out

    def reduce_all(self):
        for s in range(self.store.n_students):
            self.reduce_student_async(s)
        self.finish()


class DistributedDataParallel(nn.Module):
    """Drop-in for torch.nn.parallel.DistributedDataParallel around a dupl_amd siamese_network / network
    (same constructor keywords accepted; device_ids / find_unused_parameters are irrelevant here)."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, process_group=None,
                 bucket_mb: float = 128.0, blocks_per_bucket: int = 2, **_ignored):
        super().__init__()
        self.module = module
        store = module.flat_storage if hasattr(module, "flat_storage") else module._store
        self.reducer = GradReducer(store, process_group, bucket_mb, blocks_per_bucket)
        self.reducer.broadcast_parameters(0)
        self._callback_queued = False
        self._students = [module.branch1, module.branch2] if hasattr(module, "branch1") else [module]
        self._touched, self._reduced = set(), set()
        for net in self._students:
            net._post_backward_hooks.append(self._on_student_backward)
            net._grad_ready_hooks.append(self._on_grad_ready)

    def _queue_finalize(self):
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

    def _on_grad_ready(self, net, event):
        """During the last pending backward of a student: a piece of its gradient range is final."""
        self._touched.add(net._student)
        self._queue_finalize()
        self.reducer.grad_ready(net._student, event)

    def _on_student_backward(self, net):
        s = net._student
        self._touched.add(s)
        # a student's range is final once every live forward of it has been back-propagated (phase C runs two);
        # whatever the per-layer events have not issued yet goes out now
        if net._live_graphs == 0 and s not in self._reduced:
            self._reduced.add(s)
            self.reducer.reduce_student_async(s)
        self._queue_finalize()

    def _finalize(self):
        """End of the autograd pass: reduce whatever is left (forwards whose outputs never reached the loss keep
        _live_graphs > 0), then wait + average."""
        self._callback_queued = False
        for net in self._students:
            if net._student in self._touched and net._student not in self._reduced:
                self.reducer.reduce_student_async(net._student)
            net._live_graphs = 0
        self._touched.clear()
        self._reduced.clear()
        self.reducer.finish()

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def state_dict(self, *args, **kwargs):
        """Keys carry the `module.` prefix like the checkpoints the reference saves (train_final_voc.py:519)."""
        return super().state_dict(*args, **kwargs)
