"""PolyWarmupAdamW (reference: utils/optimizer.py:38-68) as a fused flat-buffer optimiser.

Same constructor and schedule as the reference (linear warm-up from `warmup_ratio` over `warmup_iter`
steps, then poly decay; lr applied BEFORE the update, global_step incremented after).  When the
parameters are views of a dupl_amd FlatStorage (the normal case) the update is one HIP launch per
(student, param-group) segment over the flat param / grad / moment buffers; segments that have never
received a gradient are skipped, which is exactly torch's `if p.grad is None: continue`.
"""
import math

import torch

import os

from .. import ops
from ..engine import FlatStorage, SEG_BACKBONE, SEG_NORM

# the AdamW launches of a step ride in its backward pass (PolyWarmupAdamW.begin_step; world 1): DUPL_ADAMW_IN_BWD=0 = all in step()
ADAMW_IN_BACKWARD = os.environ.get("DUPL_ADAMW_IN_BWD", "1") != "0"


def adamw_segment(p, g, m, v, step, lr, beta1, beta2, eps, wd, planes=None, grad_scale: float = 1.0):
    """One fused AdamW update of flat fp32 tensors (torch.optim.AdamW single-tensor semantics).  planes = (hi pointer, lo pointer,
    plane exponent): the updated values are also written as the f16x3 operand planes of the next forward.  grad_scale != 1: the
    gradient is taken as g * grad_scale (one rounding) and written back so (the 1 / world of a data-parallel exchange)."""
    bc1 = 1.0 - beta1 ** step
    bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
    hi, lo, e = planes if planes is not None else (None, None, 0)
    ops.L().dupl_adamw(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(lr), float(beta1),
                       float(beta2), float(eps), float(wd), float(bc1), float(bc2_sqrt), hi, lo, e, float(grad_scale), ops._stream())


class PolyWarmupAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr, weight_decay, betas, warmup_iter=None, max_iter=None, warmup_ratio=None, power=None,
                 **kwargs):
        defaults = dict(lr=lr, betas=betas, eps=1e-8, weight_decay=weight_decay)
        super().__init__(params, defaults)
        self.global_step = 0
        self.warmup_iter = warmup_iter
        self.warmup_ratio = warmup_ratio
        self.max_iter = max_iter
        self.power = power
        self.__init_lr = [group["lr"] for group in self.param_groups]
        self._flat = None          # (FlatStorage, exp_avg, exp_avg_sq, per-segment step counts)
        self._seg_group = None

    # ---- flat storage discovery -----------------------------------------------------------------
    def bind(self, store: FlatStorage):
        """Attach the flat storage the parameters are views of (done automatically by `step` through
        `store_of`, or explicitly by the training script)."""
        dev = store.data.device
        m = torch.empty_like(store.data)
        v = torch.empty_like(store.data)
        ops.fill_(m, 0.0)
        ops.fill_(v, 0.0)
        steps = [[0] * 5 for _ in range(store.n_students)]
        self._flat = (store, m, v, steps)
        # map segment id (1..4) -> optimizer param group index by locating one parameter of each group
        ptr_to_group = {}
        for gi, grp in enumerate(self.param_groups):
            for p in grp["params"]:
                ptr_to_group[p.data_ptr()] = gi
        self._seg_group = {}
        for seg in range(1, 5):
            lo, hi = store.seg_bounds[seg]
            for key, (off, n) in store.layout.items():
                if lo <= off < hi:
                    ptr = store.view(0, key).data_ptr()
                    if ptr in ptr_to_group:
                        self._seg_group[seg] = ptr_to_group[ptr]
                        break
        return self

    # ---- the update rides in the backward pass (world 1) -----------------------------------------------------------------
    def begin_step(self, model) -> bool:
        """Call between the forward and loss.backward() of a step (trainer.train_step does): from then on each piece of a student's
        gradient range is updated AS SOON AS the backward pass reports it final (network_backward's on_ready events: the heads, then
        the transformer blocks two at a time), on that student's stream -- so the HBM-bound AdamW launches run under the other
        student's MFMA-bound backward instead of alone on the chip after it (1.1 ms of a 52 ms step).  step() then updates what is
        left (the stem, the LayerNorm segment) and does the bookkeeping.  The result is bit-identical to a plain step(): the same
        element-wise kernel with the same scalars over a partition of the same ranges.
        Under a gradient exchange (ddp.DistributedDataParallel, world > 1; round 6) a rank's gradient is only final after its
        all-reduce: the reducer hands every piece to `_on_bucket_reduced` once its collective has completed (one event after it
        was issued, on the student's stream), the 1 / world goes into the kernel's gradient read (grad_scale: the rounding of the
        scale launch it replaces, written back) -- only the last bucket's reduce + update stay behind the backward pass.
        EXACTLY ONE backward pass between begin_step and step(): a second one (gradient accumulation) would meet ranges that are
        already updated -- it raises.  zero_grad() disarms (a backward pass that raised, a step that was skipped).
        Not armed (returns False; step() does everything) when DUPL_ADAMW_IN_BWD=0 or the model is not this optimiser's."""
        self._disarm()
        self._was_armed_exchange = False
        if not ADAMW_IN_BACKWARD or self._flat is None:
            return False
        core = model.module if hasattr(model, "module") else model
        reducer = getattr(model, "reducer", None) if core is not model else None
        store = self._flat[0]
        nets = [core.branch1, core.branch2] if hasattr(core, "branch1") else [core]
        if getattr(core, "flat_storage", getattr(core, "_store", None)) is not store:
            return False
        exchange = reducer is not None and getattr(reducer, "world", 1) > 1
        for net in nets:
            if self._on_grad_ready not in net._grad_ready_hooks:
                net._grad_ready_hooks.append(self._on_grad_ready)
        self._set_schedule_lr()
        self._armed = {"with_planes": [store.planes_current(s) for s in range(store.n_students)],
                       "done": [[] for _ in range(store.n_students)],            # (lo, hi) flat ranges already updated this step
                       "stepped": [set() for _ in range(store.n_students)],      # segments whose step count was advanced
                       "plan": [store.grad_buckets(s) for s in range(store.n_students)],
                       "events": set(),                                          # (student, event) seen: one backward pass per step
                       "exchange": exchange, "reducer": reducer if exchange else None}
        if exchange:
            reducer.consumer = self._on_bucket_reduced
        self._was_armed_exchange = exchange      # (bench.py reports whether the update rode in the exchange)
        return True

    def _disarm(self):
        arm = getattr(self, "_armed", None)
        self._armed = None
        if arm is not None and arm.get("reducer") is not None and arm["reducer"].consumer == self._on_bucket_reduced:
            arm["reducer"].consumer = None
        return arm

    def _on_bucket_reduced(self, s: int, lo: int, hi: int, inv: float) -> bool:
        """ddp.GradReducer consumer: grad[lo:hi) of student s holds the SUM over ranks, complete in the order of the current stream.
        Update it with the gradient read as sum * inv (and written back as that mean).  Returns True: nothing left to scale."""
        arm = getattr(self, "_armed", None)
        if arm is None or not arm["exchange"]:
            return False
        store = self._flat[0]
        base = s * store.student_numel
        b0, b1 = base + store.seg_bounds[SEG_BACKBONE][0], base + store.seg_bounds[SEG_BACKBONE][1]
        n0, n1 = base + store.seg_bounds[SEG_NORM][0], base + store.seg_bounds[SEG_NORM][1]
        if lo < b1 and hi > b0:
            store.seg_has_grad[s][SEG_BACKBONE] = True     # as in _on_grad_ready: P.mark_grad comes at the very end of the pass
        if lo < n1 and hi > n0:
            store.seg_has_grad[s][SEG_NORM] = True         # (the norm range is only issued by the "stem" event: final by then)
        with torch.no_grad():
            self._update_range(s, lo, hi, grad_scale=inv)
        return True

    def _update_range(self, s: int, lo: int, hi: int, grad_scale: float = 1.0):
        """AdamW over the flat range [lo, hi) of student s (absolute offsets), segment by segment, on the current stream."""
        store, m, v, steps = self._flat
        arm = self._armed
        base = s * store.student_numel
        for a_, b_ in arm["done"][s]:
            if a_ < hi and lo < b_:
                raise RuntimeError("PolyWarmupAdamW: a gradient range came back final twice between begin_step() and step() -- "
                                   "exactly one backward pass per armed step (call step(), or zero_grad(), in between)")
        for seg in range(1, 5):
            if seg not in self._seg_group or not store.seg_has_grad[s][seg]:
                continue
            slo, shi = store.seg_bounds[seg]
            a, b = max(lo, base + slo), min(hi, base + shi)
            if b <= a:
                continue
            if seg not in arm["stepped"][s]:
                arm["stepped"][s].add(seg)
                steps[s][seg] += 1
            grp = self.param_groups[self._seg_group[seg]]
            b1, b2 = grp["betas"]
            sl = slice(a, b)
            adamw_segment(store.data[sl], store.grad[sl], m[sl], v[sl], steps[s][seg], grp["lr"], b1, b2, grp["eps"],
                          grp["weight_decay"], planes=store.plane_pointers(a, seg) if arm["with_planes"][s] else None,
                          grad_scale=grad_scale)
            arm["done"][s].append((a, b))

    def _on_grad_ready(self, net, event):
        """network_backward's on_ready (last pending backward of this student, on its stream): update the buckets `event` finalises.
        The stem / LayerNorm buckets wait for step(): SEG_BACKBONE / SEG_NORM are marked as having gradients at the very end."""
        arm = getattr(self, "_armed", None)
        if arm is None:
            return
        s = net._student
        if (s, event) in arm["events"]:
            raise RuntimeError("PolyWarmupAdamW: a second backward pass between begin_step() and step() (gradient accumulation is "
                               "not supported with the update in the backward pass: call begin_step() before the LAST backward only)")
        arm["events"].add((s, event))
        if arm["exchange"] or event == "stem":       # under an exchange the reducer hands the reduced pieces to _on_bucket_reduced
            return
        store = self._flat[0]
        if event != "heads":
            store.seg_has_grad[s][SEG_BACKBONE] = True     # a transformer block was back-propagated (P.mark_grad comes at the end)
        with torch.no_grad():
            for lo, hi, trig in arm["plan"][s]:
                if trig == event:
                    self._update_range(s, lo, hi)

    def zero_grad(self, set_to_none: bool = False):
        self._disarm()       # a step that was armed but never taken (backward raised, non-finite loss skipped): nothing stays hooked
        if self._flat is not None:
            self._flat[0].wait_streams()
            ops.fill_(self._flat[0].grad, 0.0)   # keep the .grad views alive: one fused fill
        else:
            super().zero_grad(set_to_none=False)

    def _set_schedule_lr(self):
        # schedule (optimizer.py:51-63)
        if self.global_step < self.warmup_iter:
            lr_mult = 1 - (1 - self.global_step / self.warmup_iter) * (1 - self.warmup_ratio)
            for i in range(len(self.param_groups)):
                self.param_groups[i]["lr"] = self.__init_lr[i] * lr_mult
        elif self.global_step < self.max_iter:
            lr_mult = (1 - self.global_step / self.max_iter) ** self.power
            for i in range(len(self.param_groups)):
                self.param_groups[i]["lr"] = self.__init_lr[i] * lr_mult

    @torch.no_grad()
    def step(self, closure=None):
        self._set_schedule_lr()
        if self._flat is None:
            raise RuntimeError("PolyWarmupAdamW.bind(model.flat_storage) must be called once: dupl_amd updates the flat "
                               "parameter buffer with fused HIP launches (no per-tensor fallback)")
        store, m, v, steps = self._flat
        store.wait_streams()     # the students' backward passes may still be running on their own streams
        # a student whose operand planes are current gets them rewritten by the update itself (no split pass over the weights
        # before the next forward); segments without gradients keep their values, hence their planes
        arm = self._disarm()
        with_planes = arm["with_planes"] if arm is not None else [store.planes_current(s) for s in range(store.n_students)]
        for s in range(store.n_students):
            base = s * store.student_numel
            done = sorted(arm["done"][s]) if arm is not None else []
            for seg in range(1, 5):
                if not store.seg_has_grad[s][seg] or seg not in self._seg_group:
                    continue
                grp = self.param_groups[self._seg_group[seg]]
                lo, hi = store.seg_bounds[seg]
                if hi <= lo:
                    continue
                if arm is None or seg not in arm["stepped"][s]:
                    steps[s][seg] += 1
                b1, b2 = grp["betas"]
                # what the backward pass has not updated already (begin_step): the gaps of `done` inside this segment
                cur, rest = base + lo, []
                for a_, b_ in done:
                    if b_ <= cur or a_ >= base + hi:
                        continue
                    if a_ > cur:
                        rest.append((cur, a_))
                    cur = max(cur, b_)
                if cur < base + hi:
                    rest.append((cur, base + hi))
                for a_, b_ in rest:
                    sl = slice(a_, b_)
                    adamw_segment(store.data[sl], store.grad[sl], m[sl], v[sl], steps[s][seg], grp["lr"], b1, b2, grp["eps"],
                                  grp["weight_decay"], planes=store.plane_pointers(a_, seg) if with_planes[s] else None)
        store.mark_dirty()       # parameters were rewritten through raw pointers: f16x3 operand planes not written above are stale
        for s in range(store.n_students):
            if with_planes[s]:
                store.planes_written(s)
        self.global_step += 1

    # ---- resume (the reference saves no optimiser state; --start_iter with a checkpoint is this build's own path) --------
    def state_dict(self):
        """Everything `step` depends on: schedule position, the flat moment buffers, the per-(student, segment) step
        counts of the bias correction and which segments have received gradients."""
        sd = {"global_step": self.global_step, "lr": [g["lr"] for g in self.param_groups]}
        if self._flat is not None:
            store, m, v, steps = self._flat
            store.wait_streams()
            sd.update(exp_avg=m.detach().clone(), exp_avg_sq=v.detach().clone(), steps=[list(r) for r in steps],
                      seg_has_grad=[list(r) for r in store.seg_has_grad])
        return sd

    def load_state_dict(self, sd):
        self.global_step = int(sd["global_step"])
        for g, lr in zip(self.param_groups, sd["lr"]):
            g["lr"] = lr
        if "exp_avg" in sd:
            if self._flat is None:
                raise RuntimeError("bind(model.flat_storage) before load_state_dict: the moments live in flat buffers")
            store, m, v, steps = self._flat
            if sd["exp_avg"].numel() != m.numel():
                raise ValueError(f"optimiser state of {sd['exp_avg'].numel()} elements for a model of {m.numel()}")
            m.copy_(sd["exp_avg"].to(m.device))
            v.copy_(sd["exp_avg_sq"].to(v.device))
            for s_, row in enumerate(sd["steps"]):
                steps[s_][:] = [int(x) for x in row]
            for s_, row in enumerate(sd["seg_has_grad"]):
                store.seg_has_grad[s_][:] = [bool(x) for x in row]
