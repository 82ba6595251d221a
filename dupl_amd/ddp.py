"""Data-parallel gradient synchronisation for the flat-buffer engine.

Reference mechanism (train_final_voc.py:108-110,153-155): torch DistributedDataParallel over NCCL with
find_unused_parameters=True -- a parameter broadcast at construction, a bucketed gradient all-reduce
(sum, then / world) driven by autograd hooks during loss.backward(), plus a per-step "used parameter"
bitmap all-reduce that exists only because `encoder.head` is never used.

Here the parameters of both students live in one flat buffer (engine.FlatStorage) and their gradients in a
second one, so the exchange step is: one RCCL broadcast of the parameter buffer at construction, and per
student ONE contiguous gradient range [backbone|norm|cls|decoder] all-reduced in a few large buckets
(xGMI is point-to-point: few big messages beat many small ones).  The frozen segment (pos_embed, head)
is never reduced, so no unused-parameter detection is needed.  Overlap: a student's buckets are issued
asynchronously as soon as that student's backward finishes (engine post-backward hook), i.e. student A's
gradients travel while student B's backward still computes; the optimiser waits at the end of backward
through an autograd-engine callback, exactly where stock DDP finalises.
"""
from __future__ import annotations

from typing import List, Optional

import torch
import torch.distributed as dist
import torch.nn as nn

from .engine import SEG_BACKBONE, SEG_DECODER, FlatStorage


class GradReducer:
    """Bucketed all-reduce (sum then / world) over the trainable gradient range of each student."""

    def __init__(self, store: FlatStorage, process_group=None, bucket_mb: float = 128.0):
        self.store = store
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.bucket_elems = max(1, int(bucket_mb * 1024 * 1024 / 4))
        self._pending: List = []

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

    def reduce_student_async(self, student: int):
        """Issue the all-reduces of one student's gradient range (returns immediately)."""
        if self.world == 1:
            return
        for lo, hi in self.student_buckets(student):
            t = self.store.grad[lo:hi]
            work = dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            self._pending.append((work, t))

    def finish(self):
        """Wait for the issued buckets (stream-level on RCCL, host-level on gloo) and average."""
        if not self._pending:
            return
        inv = 1.0 / self.world
        for work, t in self._pending:
            work.wait()
            if t.is_cuda:
                from . import ops
                ops.scale_(t, inv)
            else:
                t.mul_(inv)
        self._pending.clear()

    def reduce_all(self):
        for s in range(self.store.n_students):
            self.reduce_student_async(s)
        self.finish()


class DistributedDataParallel(nn.Module):
    """Drop-in for torch.nn.parallel.DistributedDataParallel around a dupl_amd siamese_network / network
    (same constructor keywords accepted; device_ids / find_unused_parameters are irrelevant here)."""

    def __init__(self, module, device_ids=None, output_device=None, find_unused_parameters=False, process_group=None,
                 bucket_mb: float = 128.0, **_ignored):
        super().__init__()
        self.module = module
        store = module.flat_storage if hasattr(module, "flat_storage") else module._store
        self.reducer = GradReducer(store, process_group, bucket_mb)
        self.reducer.broadcast_parameters(0)
        self._callback_queued = False
        self._students = [module.branch1, module.branch2] if hasattr(module, "branch1") else [module]
        self._touched, self._reduced = set(), set()
        for net in self._students:
            net._post_backward_hooks.append(self._on_student_backward)

    def _on_student_backward(self, net):
        s = net._student
        self._touched.add(s)
        # a student's range is final once every live forward of it has been back-propagated (phase C runs two)
        if net._live_graphs == 0 and s not in self._reduced:
            self._reduced.add(s)
            self.reducer.reduce_student_async(s)
        if not self._callback_queued:
            self._callback_queued = True
            torch.autograd.Variable._execution_engine.queue_callback(self._finalize)

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
