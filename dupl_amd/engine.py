"""Per-student forward / backward engine: the ViT-B/16 encoder, the CAM / classifier heads and the
LargeFOV decoder, expressed as explicit sequences of libdupl_hip.so kernel launches.

Python owns memory (torch tensors, flat parameter / gradient storage) and launch order; every dense
op is a HIP kernel.  The backward pass is hand-scheduled (no autograd inside): it consumes the
activations the forward saved and ACCUMULATES parameter gradients straight into the flat gradient
buffer, which is what the optimiser and the gradient all-reduce operate on.

Reference behaviour mirrored (paths relative to the reference):
  forward_features  model/backbone/vit.py:289-326      Block  vit.py:156-160
  Attention         vit.py:120-138                      Mlp    vit.py:97-103
  network.forward   model/model_dupl.py:69-106          LargeFOV model/decoder/conv_head.py:32-41
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor

# Dense contractions of the step: "f16x3" (default) = fp32-equivalent split products on the f16 matrix cores for the encoder
# and decoder-conv GEMMs (forward, data and weight gradients: csrc/gemm_split.hip, split_prep.hip) and the attention (forward
# and backward, head dim 64: csrc/attn_split*.hip) -- 3 f16 MFMAs per block, operands as fp16 hi / lo planes written by the
# producing kernels; closer to an fp64-accumulated reference than the fp32 fmaf chain, every parity test holds in either mode;
# "f32" = the exact-f32 MFMA kernels for all of them (csrc/gemm.hip, attn.hip; DUPL_GEMM=f32 or set_gemm_mode).  The CAM /
# classifier heads, conv8, the PTC Gram and the patch-embedding weight gradient run on the f32 kernels in both modes.
GEMM_MODE = os.environ.get("DUPL_GEMM", "f16x3")


def set_gemm_mode(mode: str):
    global GEMM_MODE
    assert mode in ("f32", "f16x3"), mode
    GEMM_MODE = mode


@dataclass(frozen=True)
class EncoderConfig:
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    mlp_ratio: int = 4
    patch: int = 16
    img_size: int = 224
    aux_layer: int = -3
    ln_eps: float = 1e-6
    head_classes: int = 1000
    decoder_dim: int = 512
    decoder_dilation: int = 5

    @property
    def grid(self) -> int:
        return self.img_size // self.patch

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads


# ------------------------------------------------------------------------------------------------
# flat parameter storage
# ------------------------------------------------------------------------------------------------
def student_param_shapes(cfg: EncoderConfig, num_classes: int) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys / shapes of one `network` in the reference's own order."""
    D, Hd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio
    s: Dict[str, Tuple[int, ...]] = {}
    s["encoder.cls_token"] = (1, 1, D)
    s["encoder.pos_embed"] = (1, cfg.grid * cfg.grid + 1, D)
    s["encoder.patch_embed.proj.weight"] = (D, 3, cfg.patch, cfg.patch)
    s["encoder.patch_embed.proj.bias"] = (D,)
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}."
        s[p + "norm1.weight"] = (D,)
        s[p + "norm1.bias"] = (D,)
        s[p + "attn.qkv.weight"] = (3 * D, D)
        s[p + "attn.qkv.bias"] = (3 * D,)
        s[p + "attn.proj.weight"] = (D, D)
        s[p + "attn.proj.bias"] = (D,)
        s[p + "norm2.weight"] = (D,)
        s[p + "norm2.bias"] = (D,)
        s[p + "mlp.fc1.weight"] = (Hd, D)
        s[p + "mlp.fc1.bias"] = (Hd,)
        s[p + "mlp.fc2.weight"] = (D, Hd)
        s[p + "mlp.fc2.bias"] = (D,)
    s["encoder.norm.weight"] = (D,)
    s["encoder.norm.bias"] = (D,)
    s["encoder.head.weight"] = (cfg.head_classes, D)
    s["encoder.head.bias"] = (cfg.head_classes,)
    s["decoder.conv6.weight"] = (cfg.decoder_dim, D, 3, 3)
    s["decoder.conv7.weight"] = (cfg.decoder_dim, cfg.decoder_dim, 3, 3)
    s["decoder.conv8.weight"] = (num_classes, cfg.decoder_dim, 1, 1)
    s["classifier.weight"] = (num_classes - 1, D, 1, 1)
    s["aux_classifier.weight"] = (num_classes - 1, D, 1, 1)
    return s


# segment ids: 0 = never receives a gradient (pos_embed frozen vit.py:243; `head` unused by forward_features),
# 1..4 = optimiser groups 0..3 of siamese_network.get_param_groups (model_dupl.py:119-154)
SEG_FROZEN, SEG_BACKBONE, SEG_NORM, SEG_CLS, SEG_DECODER = 0, 1, 2, 3, 4


def param_segment(key: str) -> int:
    if key in ("encoder.pos_embed", "encoder.head.weight", "encoder.head.bias"):
        return SEG_FROZEN
    if key.startswith("encoder."):
        return SEG_NORM if "norm" in key[len("encoder."):] else SEG_BACKBONE
    if key.startswith("decoder."):
        return SEG_DECODER
    return SEG_CLS


# operand plane format of the encoder's forward GEMMs: 1 = prescaled planes with an unscaled lo (one accumulator set, 256 x 256
# tiles: +12-15 % on the big shapes), 0 = the x 2048 lo planes everywhere (DUPL_FMT1=0; the backward always uses those)
FMT1 = os.environ.get("DUPL_FMT1", "1") != "0"
# backward GEMMs of the encoder's Linears on the forward's own operand planes, read K-MAJOR (dupl_gemm16_desc.a_layout / b_layout):
# no transposed planes, no fp32 copies of ln1 / ln2 / h1, single-accumulator kernels (round 4; DUPL_KM_BWD=0 restores the
# transposed-planes path, which sites whose planes are not format 1 take in any case)
KM_BWD = os.environ.get("DUPL_KM_BWD", "1") != "0"
# data gradients with a linear epilogue (fc1, qkv: their dx goes to a LayerNorm backward) as stream-K launches into a zero-filled dx
SK_DGRAD = os.environ.get("DUPL_SK_DGRAD", "1") != "0"
# the weight gradients of a transformer block as one grouped launch (whole tiles, no atomics); 0: one stream-K launch each
WGRAD_GROUP = os.environ.get("DUPL_WGRAD_GROUP", "1") != "0"

SK_DGRAD_MAX_COLS = 1024
ZERO_WS = os.environ.get("DUPL_ZERO_WS", "1") != "0"
# AdamW writes the operand planes of the parameters it updates (no split pass over all weights before the next forward)
FUSED_PLANES = os.environ.get("DUPL_ADAMW_PLANES", "1") != "0"
# a training step's ms-CAM scales and its training forward as ONE encoder pass (cam_logits_shared_multi) while the pass has at most
# this many token rows; above (and with 0): scale 1.0 (saved) and the remaining scales as two passes (round 4).  Measured, same box:
# 2 img/GPU (10 988 rows) 29.46 vs 30.29 ms per step merged (one stream: 37.0 vs 39.4); 4 img/GPU (21 976 rows) 53.7 vs 52.6 (63.0 vs
# 61.5) -- at 21 976 rows the widest activation of a block (the GELU output planes, 270 MB) no longer fits the 256 MB Infinity
# Cache between the GEMM that writes it and the one that reads it
MERGED_PASS = int(os.environ.get("DUPL_MERGED_PASS", "16384"))
# the attention forward of all batches of a merged pass as ONE launch (ops.attention_fwd16_segs) while it has at most this many
# 128-query blocks (0: always one launch per batch).  Measured, two student streams, same box: 2 img/GPU (1 104 blocks) 28.87 vs
# 29.27 ms per step with the single launch; 4 img/GPU (2 208 blocks) 52.07 vs 51.80 -- a grid that holds every block slot of the chip
# for several rounds keeps the other student's kernels out longer than its filled tails give back
ATTN_SEGS = int(os.environ.get("DUPL_ATTN_SEGS", "1536"))


class FlatStorage:
    """All parameters of `n_students` students in ONE fp32 buffer (+ a same-shaped gradient buffer):
        [student 0: frozen | backbone | norm | cls | decoder][student 1: ...]
    so that the optimiser is 4 fused launches per student, the gradient all-reduce runs over a few large
    contiguous buckets, and student 2's copy of any tensor sits at a constant offset from student 1's."""

    def __init__(self, cfg: EncoderConfig, num_classes: int, n_students: int, device="cpu"):
        self.cfg, self.num_classes, self.n_students = cfg, num_classes, n_students
        shapes = student_param_shapes(cfg, num_classes)
        self.shapes = shapes
        self.layout: Dict[str, Tuple[int, int]] = {}     # key -> (offset within a student, numel)
        self.seg_bounds: List[Tuple[int, int]] = []      # per segment id: (start, end) within a student
        off = 0
        for seg in range(5):
            start = off
            for k, shp in shapes.items():
                if param_segment(k) != seg:
                    continue
                n = 1
                for d in shp:
                    n *= d
                self.layout[k] = (off, n)
                off += (n + 7) // 8 * 8   # 32-byte alignment of every tensor (16 bytes in the fp16 operand planes)
            self.seg_bounds.append((start, off))
        self.student_numel = off
        self.data = torch.zeros(n_students * off, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n_students * off, dtype=torch.float32, device=device)
        # sticky "this segment has received a gradient at least once" flags == torch's `p.grad is None` skip
        self.seg_has_grad = [[False] * 5 for _ in range(n_students)]
        self.streams: List = []     # side streams the students run on (siamese_network.enable_dual_stream)
        self.version_anchor = None  # a Parameter that shares the version counter of every student Parameter (_param_key)
        # launch tuning of this model's split GEMMs (dupl_gemm16_desc.tile / concurrency / persist_blocks / group / sk_slices): the
        # model's own copy of the caller-side defaults -- enable_dual_stream of one model does not retune another
        self.gemm16_tuning = dict(ops.GEMM16_TUNING)
        # f16x3 operand planes of the parameters (ops.Split16 format, csrc/gemm_split.hip): [2, n_students * numel] fp16,
        # refreshed per student when its key (torch version counter of the flat buffer, `dirty` bumped by raw-pointer
        # writers such as the optimiser kernel, buffer address) changes
        self.data16: Optional[Tensor] = None
        self.dirty = 0
        self.rewrites = 0           # bulk rewrites through raw pointers / collectives (mark_dirty(rewritten=True))
        self._w16_key = [None] * n_students
        self._planes_fresh: Dict = {}      # student -> parameter key whose planes the optimiser has written
        self._w16T: Dict = {}
        self._w16F0: Dict = {}
        self.guard = RangeGuard(self)

    def wait_streams(self):
        """Make the current stream wait for everything queued on the student streams."""
        if self.streams:
            cur = torch.cuda.current_stream()
            for s in self.streams:
                cur.wait_stream(s)

    def view(self, student: int, key: str, grad: bool = False) -> Tensor:
        off, n = self.layout[key]
        base = student * self.student_numel + off
        buf = self.grad if grad else self.data
        return buf[base:base + n].view(self.shapes[key])

    def apply(self, fn):
        self.data = fn(self.data)
        self.grad = fn(self.grad)
        self.data16 = None
        self._w16_key = [None] * self.n_students
        self._w16T = {}

    def mark_dirty(self, rewritten: bool = False):
        """To be called by whoever rewrites parameters through raw pointers (the optimiser kernel).  rewritten=True: the
        writer replaced the parameters wholesale (a broadcast, a checkpoint copied in through a raw pointer) rather than
        moving them by <= lr -- the range guard then re-checks synchronously, as it does for torch-visible rewrites."""
        self.dirty += 1
        if rewritten:
            self.rewrites += 1

    def _param_key(self):
        # torch-visible writes, counted where torch counts them.  The students' Parameters were created as views of the ORIGINAL flat
        # buffer and keep that buffer's version counter for life -- `p.data = view_of_the_moved_buffer` (network._rebind after .to() /
        # .cuda()) re-points their storage, not their counter -- so after a move `self.data._version` no longer sees `p.copy_()`,
        # `p[i] = v` or a load_state_dict (round 6: on the GPU a rewritten parameter left the operand planes AND the range verdicts
        # stale; on the CPU, where the buffer never moves, the two counters are one and the tests never noticed).  version_anchor is
        # one of those Parameters (network._build_modules): its counter covers every Parameter of every student.
        a = self.version_anchor
        return (self.data._version + (a._version if a is not None else 0), self.dirty, self.data.data_ptr(), self.rewrites)

    # ---- the optimiser writes the planes of the parameters it updates (dupl_adamw p_hi / p_lo): utils/optimizer.py
    def planes_current(self, student: int) -> bool:
        """The student's planes match its parameters right now (so that rewriting the updated segments keeps them complete)."""
        return FUSED_PLANES and self.data16 is not None and self._w16_key[student] == self._param_key()

    def plane_pointers(self, offset: int, seg: int):
        """(hi pointer, lo pointer, plane exponent) of the planes at flat parameter `offset`, which lies in segment `seg`."""
        lo, hi = self.seg_bounds[seg]
        assert lo <= offset % self.student_numel < max(hi, lo + 1), \
            f"offset {offset} is not inside segment {seg}: the optimiser's ranges must be the plane-format ranges of ensure_w16"
        return (self.data16.data_ptr() + 2 * offset, self.data16.data_ptr() + 2 * (self.data.numel() + offset),
                ops.EXP_W if (FMT1 and seg == SEG_BACKBONE) else 0)

    def planes_written(self, student: int):
        self._planes_fresh[student] = self._param_key()

    def ensure_w16(self, student: int):
        """Bring the fp16 hi / lo planes of one student's parameters up to date (on the current stream)."""
        key = self._param_key()
        if self._w16_key[student] == key and self.data16 is not None:
            return
        if self.data16 is None or self.data16.device != self.data.device:
            self.data16 = torch.empty((2, self.data.numel()), device=self.data.device, dtype=torch.float16)
            self._w16_key = [None] * self.n_students
            self._planes_fresh = {}
        n, base = self.student_numel, student * self.student_numel
        tot = self.data.numel()
        if self._planes_fresh.get(student) == key:
            pass        # the optimiser step that produced these parameters wrote their planes
        elif FMT1:
            # the backbone segment (every encoder Linear weight) as format 1 planes of w * 2^EXP_W (single-accumulator forward
            # GEMMs), everything else (decoder convs) as format 0
            b0, b1 = self.seg_bounds[SEG_BACKBONE]
            for lo_, hi_, e in ((0, b0, 0), (b0, b1, ops.EXP_W), (b1, n, 0)):
                if hi_ > lo_:
                    o = base + lo_
                    args = (self.data.data_ptr() + 4 * o, self.data16.data_ptr() + 2 * o, self.data16.data_ptr() + 2 * (tot + o), hi_ - lo_)
                    if e:
                        ops.L().dupl_split_f16x2b(*args, e, ops._stream())
                    else:
                        ops.L().dupl_split_f16x2(*args, ops._stream())
        else:
            ops.L().dupl_split_f16x2(self.data.data_ptr() + 4 * base, self.data16.data_ptr() + 2 * base,
                                     self.data16.data_ptr() + 2 * (tot + base), n, ops._stream())
        old = self._w16_key[student]
        self._w16_key[student] = key
        # the operands changed: re-check their range (synchronously unless this was an optimiser step)
        self.guard.params_changed(student, rewritten=(old is None or old[0] != key[0] or old[2] != key[2] or old[3] != key[3]),
                                  key=key)

    def w16T(self, student: int, key: str, rows: int):
        """Operand planes of the TRANSPOSE of parameter `key` ([rows, cols] -> planes [cols, rows]): the B operand of the data
        gradient dx = dy . W as a k-contiguous product.  Built on first use after every parameter change (backward only)."""
        ver = self._param_key()
        hit = self._w16T.get((student, key))
        if hit is not None and hit[0] == ver:
            return hit[1]
        w = self.view(student, key).view(rows, -1)
        _, T, _ = ops.split_prepare(w, scaled=False, want_rm=False, want_T=True, rows_pad=rows)
        self._w16T[(student, key)] = (ver, T)
        return T

    def w16T_missing(self, student: int, keys_rows):
        """[(key, rows, fp32 view [rows, cols])] of the parameters whose transposed planes are stale: the caller splits them
        (together with other operands, ops.split_prepare_multi) and hands the planes back through w16T_put."""
        ver = self._param_key()
        out = []
        for key, rows in keys_rows:
            hit = self._w16T.get((student, key))
            if hit is None or hit[0] != ver:
                out.append((key, rows, self.view(student, key).view(rows, -1)))
        return out

    def w16T_put(self, student: int, key: str, T):
        self._w16T[(student, key)] = (self._param_key(), T)

    def w16(self, student: int, key: str, rows: int, fmt1: Optional[bool] = None) -> "ops.W16":
        """Operand planes of parameter `key` viewed as a [rows, numel / rows] matrix."""
        off, n = self.layout[key]
        base = student * self.student_numel + off
        p = self.data16.data_ptr()
        have = ops.EXP_W if (FMT1 and param_segment(key) == SEG_BACKBONE) else 0
        if fmt1 is None:
            fmt1 = bool(have)
        if bool(have) == bool(fmt1):
            return ops.W16(p + 2 * base, p + 2 * (self.data.numel() + base), rows, n // rows, have)
        # a backbone weight whose site does not fit the format 1 scale (RangeGuard "<site>_f1"): format 0 planes, built on first
        # use after every parameter change
        assert not fmt1
        ver = self._param_key()
        hit = self._w16F0.get((student, key))
        if hit is None or hit[0] != ver:
            hit = (ver, ops.split16(self.view(student, key).view(rows, -1)))
            self._w16F0[(student, key)] = hit
        return hit[1]

    def trainable_range(self, student: int) -> Tuple[int, int]:
        s = student * self.student_numel
        return s + self.seg_bounds[SEG_BACKBONE][0], s + self.seg_bounds[SEG_DECODER][1]

    def block_range(self, i: int) -> Tuple[int, int]:
        """[start, end) within a student of transformer block i's tensors in the backbone segment."""
        pre = f"encoder.blocks.{i}."
        offs = [(o, n) for k, (o, n) in self.layout.items() if k.startswith(pre) and param_segment(k) == SEG_BACKBONE]
        return min(o for o, _ in offs), max((o + n + 7) // 8 * 8 for o, n in offs)

    def grad_buckets(self, student: int, blocks_per_bucket: int = 2) -> List[Tuple[int, int, object]]:
        """Partition of the student's trainable gradient range into (lo, hi, trigger) buckets in the order the backward
        pass finalises them (network_backward's on_ready events): "heads" = [cls | decoder] once the heads and the
        decoder are back-propagated; an int i = the blocks [i, i+k) once block i is done (the backward walks the blocks
        downwards, their tensors are contiguous); "stem" = cls_token / patch embedding plus the lowest blocks, and the
        norm segment (LayerNorm affine parameters of every block, final only at the end)."""
        s = student * self.student_numel
        depth = self.cfg.depth
        k = max(1, int(blocks_per_bucket))
        out = [(s + self.seg_bounds[SEG_CLS][0], s + self.seg_bounds[SEG_DECODER][1], "heads")]
        i = depth
        while i - k > 0:
            out.append((s + self.block_range(i - k)[0], s + self.block_range(i - 1)[1], i - k))
            i -= k
        out.append((s + self.seg_bounds[SEG_BACKBONE][0], s + self.block_range(i - 1)[1], "stem"))
        out.append((s + self.seg_bounds[SEG_NORM][0], s + self.seg_bounds[SEG_NORM][1], "stem"))
        return out


# ------------------------------------------------------------------------------------------------
# range guard of the f16x3 operand planes
# ------------------------------------------------------------------------------------------------
F16_MAX = 65504.0


class RangeGuard:
    """Keeps every tensor that is written as fp16 hi / lo planes inside fp16's exponent range BY CONSTRUCTION.

    The split format x = hi + lo / 2048 (csrc/gemm_split.hip) saturates at |x| = 65504, the reference's fp32 at 3.4e38
    (deit.py:102-108 loads pretrained weights: outlier channels are real).  For every Linear / attention of a student a
    rigorous bound on its operands follows from the parameters alone (csrc/range.hip):
        LayerNorm out  |y_j| <= max|gamma| sqrt(D) + max|beta|,  ||y||_2 <= max|gamma| sqrt(D) + ||beta||_2
        Linear out     |(W y + b)_j| <= ||y||_2 max_j ||W_j||_2 + max|b|;   GELU, ReLU, softmax-weighted means: |f(x)| <= |x|
    A site whose operand bound (activation side or weight max-abs), times `margin`, exceeds 65504 runs on the exact-f32
    MFMA kernels instead -- forward and backward, its producer then hands it fp32 instead of planes -- so an operand
    beyond fp16's range gets the fp32 answer, never a clamp.  Gradients are scaled from their own max-abs on the device
    (split_prepare) and cannot saturate.  The image itself is the one operand without a parameter bound: the patch
    embedding assumes |pixel| <= 65504 / margin (normalised images are < 3).

    Freshness: a wholesale parameter rewrite (load_state_dict, .copy_: torch version counter / address change) is checked
    synchronously before the next forward; optimiser steps (raw-pointer writes, FlatStorage.mark_dirty) move a weight by
    <= lr per step, so they are re-checked every `period` steps from an ASYNCHRONOUS device -> pinned-host copy (no
    host-device synchronisation on the step path) -- what `margin` = 2 is for.  The copy launched at step k * period is
    harvested at step (k + 1) * period exactly, so the step at which a site changes route does not depend on host / device
    timing (bit-reproducible under DUPL_DETERMINISTIC=1, identical on every DDP rank); a backward pass uses the verdicts its
    forward ran with (EncoderSaved.guard / HeadSaved.guard), whatever has been harvested in between."""

    SITES = ("qkv", "attn", "proj", "fc1", "fc2")

    def __init__(self, store: "FlatStorage", margin: float = 2.0, period: int = 8):
        self.store, self.margin, self.period = store, float(margin), int(period)
        cfg = store.cfg
        self.entries: List[Tuple[str, int, int]] = []        # (key, rows, cols) in table order
        D, Hd, dd = cfg.embed_dim, cfg.embed_dim * cfg.mlp_ratio, cfg.decoder_dim

        def add(key, rows):
            n = store.layout[key][1]
            self.entries.append((key, rows, n // rows))

        add("encoder.patch_embed.proj.weight", D)
        for i in range(cfg.depth):
            p = f"encoder.blocks.{i}."
            for k, r in ((p + "norm1.weight", 1), (p + "norm1.bias", 1), (p + "attn.qkv.weight", 3 * D), (p + "attn.qkv.bias", 1),
                         (p + "attn.proj.weight", D), (p + "norm2.weight", 1), (p + "norm2.bias", 1), (p + "mlp.fc1.weight", Hd),
                         (p + "mlp.fc1.bias", 1), (p + "mlp.fc2.weight", D)):
                add(k, r)
        for k, r in (("encoder.norm.weight", 1), ("encoder.norm.bias", 1), ("decoder.conv6.weight", dd), ("decoder.conv7.weight", dd)):
            add(k, r)
        self.index = {k: i for i, (k, _, _) in enumerate(self.entries)}
        self._table = None          # device descriptor table (offsets relative to a student's base)
        self._dev_out = None        # [n_students, n, 2] device floats
        self._host = None           # pinned mirror
        self._event = None
        self._pending = False
        self._steps = [0] * store.n_students
        self.safe: List[Optional[dict]] = [None] * store.n_students      # per student: site flags, None = not computed yet
        self.headroom = float("inf")   # min over sites of 65504 / (margin * bound) at the last check (< 1: some site is on f32)
        self.headroom1 = float("inf")  # the same at the format 1 scales (< 1: some encoder site runs on format 0 planes)
        self.checks = 0
        self._checked_key = None       # parameter state (FlatStorage._param_key) of the last synchronous check

    # ---- device side
    def _ensure_buffers(self):
        dev = self.store.data.device
        if self._table is None or self._table.device != dev:
            import numpy as np
            tab = np.zeros((len(self.entries), 2), dtype=np.int64)     # {int64 offset, int32 rows | int32 cols << 32}
            for i, (k, rows, cols) in enumerate(self.entries):
                tab[i, 0] = self.store.layout[k][0]
                tab[i, 1] = rows | (cols << 32)
            self._table = torch.from_numpy(tab).to(dev)
            self._dev_out = torch.zeros((self.store.n_students, len(self.entries), 2), device=dev, dtype=torch.float32)
            self._host = torch.zeros((self.store.n_students, len(self.entries), 2), dtype=torch.float32).pin_memory()
            self._event = torch.cuda.Event()

    def _launch(self, students):
        self._ensure_buffers()
        st = self.store
        for s in students:
            ops.L().dupl_param_bounds(st.data.data_ptr() + 4 * s * st.student_numel, self._table.data_ptr(), len(self.entries),
                                      self._dev_out[s].data_ptr(), ops._stream())
        self._host.copy_(self._dev_out, non_blocking=True)
        self._event.record()
        self._pending = True

    def _harvest(self, wait: bool):
        if not self._pending:
            return
        if wait:
            self._event.synchronize()
        elif not self._event.query():
            return
        self._pending = False
        self.headroom = self.headroom1 = float("inf")
        for s in range(self.store.n_students):
            self.safe[s] = self._decide(self._host[s].double().numpy())
        self.checks += 1

    # ---- host side
    def _decide(self, v):
        """v[e] = (max-abs, largest row L2 norm) per table entry -> {site: f16-safe?}."""
        cfg = self.store.cfg
        sq = float(cfg.embed_dim) ** 0.5
        lim = F16_MAX / self.margin
        ix = self.index
        worst = [0.0]

        def ok(*bounds):
            good = True
            for b in bounds:                # element-wise: Python's max() drops a NaN that is not its first argument
                b = float(b)
                if not (b <= lim):          # NaN / inf -> False
                    good = False
                if b == b:
                    worst[0] = max(worst[0], b)
            return good

        def amax(k):
            return v[ix[k], 0]

        def rown(k):
            return v[ix[k], 1]

        # format 1 planes hold x * 2^e / w * 2^EXP_W: "<site>_f1" is the largest activation exponent e <= EXP_ACT at which the
        # site's operands still fit (0: none -- a site that is fp16-safe but not at any format 1 scale runs on format 0 planes,
        # two accumulator sets, not on f32).  A smaller e only moves the point below which an element's lo is a subnormal
        # (|x| < 2^(-2 - e)) up -- for operands whose bound is that large, still far below their typical size.
        sw = 2.0 ** ops.EXP_W
        worst1 = [0.0]

        def exp1(ea_, ew_):
            if not FMT1 or not (ew_ * sw <= lim):
                worst1[0] = max(worst1[0], ew_ * sw if FMT1 else 0.0)
                return 0
            for e in range(ops.EXP_ACT, 0, -1):
                if ea_ * 2.0 ** e <= lim:
                    worst1[0] = max(worst1[0], ea_ * 2.0 ** e, ew_ * sw)
                    return e
            worst1[0] = max(worst1[0], ea_ * 2.0)
            return 0

        wp = amax("encoder.patch_embed.proj.weight")
        out = {"patch": ok(wp), "blocks": []}
        out["patch_f1"] = exp1(3.0, wp) if out["patch"] else 0
        for i in range(cfg.depth):
            p = f"encoder.blocks.{i}."
            g1, g2 = amax(p + "norm1.weight"), amax(p + "norm2.weight")
            el1, n1 = g1 * sq + amax(p + "norm1.bias"), g1 * sq + rown(p + "norm1.bias")
            el2, n2 = g2 * sq + amax(p + "norm2.bias"), g2 * sq + rown(p + "norm2.bias")
            e_qkv = n1 * rown(p + "attn.qkv.weight") + amax(p + "attn.qkv.bias")
            e_h = n2 * rown(p + "mlp.fc1.weight") + amax(p + "mlp.fc1.bias")
            ops_ = {"qkv": (el1, amax(p + "attn.qkv.weight")), "proj": (e_qkv, amax(p + "attn.proj.weight")),
                    "fc1": (el2, amax(p + "mlp.fc1.weight")), "fc2": (e_h, amax(p + "mlp.fc2.weight"))}
            blk = {"attn": ok(e_qkv)}
            for site, (ea_, ew_) in ops_.items():
                blk[site] = ok(ea_, ew_)
                blk[site + "_f1"] = exp1(ea_, ew_) if blk[site] else 0
            out["blocks"].append(blk)
        gf = amax("encoder.norm.weight")
        elf, nf = gf * sq + amax("encoder.norm.bias"), gf * sq + rown("encoder.norm.bias")
        e_c6 = 3.0 * nf * rown("decoder.conv6.weight")                   # 9 taps: ||patch||_2 <= 3 max ||token||_2
        out["conv6"] = ok(elf, amax("decoder.conv6.weight"))
        out["conv7"] = ok(e_c6, amax("decoder.conv7.weight"))
        self.headroom = min(self.headroom, float(lim / max(worst[0], 1e-30)))
        self.headroom1 = min(self.headroom1, float(lim / max(worst1[0], 1e-30)))
        return out

    def params_changed(self, student: int, rewritten: bool, key=None):
        """Called by FlatStorage.ensure_w16 when the operand planes of a student are rebuilt.  rewritten: the buffer was
        replaced or written through torch (load_state_dict, copy_) or wholesale through a raw pointer / collective
        (mark_dirty(rewritten=True)) rather than by an optimiser step.  key: the parameter state the planes were built from --
        one synchronous check covers every student of that state (the other student's rebuild does not repeat it)."""
        if rewritten or self.safe[student] is None:
            if key is not None and key == self._checked_key and self.safe[student] is not None:
                return
            self._launch(range(self.store.n_students))
            self._harvest(wait=True)
            self._steps = [0] * self.store.n_students
            self._checked_key = key
            return
        self._steps[student] += 1
        if student == 0 and self._steps[0] % self.period == 0:
            # the bounds launched `period` optimiser steps ago are taken in HERE, at a fixed step of the run (the host is at most
            # a step or two ahead of the device, so this wait does not block in practice) -- not whenever the copy happens to
            # have landed: a site then changes route at the same step in every run and on every DDP rank
            self._harvest(wait=True)
            self._launch(range(self.store.n_students))

    def sites(self, student: int) -> dict:
        if self.safe[student] is None:
            self._launch(range(self.store.n_students))
            self._harvest(wait=True)
        return self.safe[student]

    def summary(self) -> dict:
        """For logs / bench.py: how many sites run on the f32 kernels because their operands could leave fp16's range."""
        n = n0 = 0
        for s in self.safe:
            if s is None:
                continue
            n += sum(not s[k] for k in ("patch", "conv6", "conv7")) + sum(not b[k] for b in s["blocks"] for k in self.SITES)
            if FMT1:
                n0 += int(s["patch"] and not s["patch_f1"]) + sum(b[k] and not b[k + "_f1"] for b in s["blocks"]
                                                                  for k in ("qkv", "proj", "fc1", "fc2"))
        return {"sites_on_f32": n, "sites_on_fmt0": n0, "checks": self.checks, "margin": self.margin,
                "headroom": (None if self.headroom == float("inf") else round(self.headroom, 1)),
                "headroom_fmt1": (None if self.headroom1 == float("inf") else round(self.headroom1, 1))}



def _lin16(P: "StudentParams", *a, **kw):
    """ops.linear16 with the launch tuning of the model that owns P (FlatStorage.gemm16_tuning: per model, not process-wide)."""
    return ops.linear16(*a, tuning=P.store.gemm16_tuning, **kw)


class StudentParams:
    """Read-only bundle of one student's parameter / gradient views used by the engine."""

    def __init__(self, store: FlatStorage, student: int):
        self.store, self.student = store, student
        self.cfg, self.num_classes = store.cfg, store.num_classes
        self.w = {k: store.view(student, k) for k in store.layout}
        self.g = {k: store.view(student, k, grad=True) for k in store.layout}
        self._pos_cache: Dict[Tuple[int, int], Tensor] = {}
        self._pos_version = None

    def pos_embed_for(self, h: int, w: int) -> Tensor:
        """Bicubic-resized pos-embed, cached per resolution: pos_embed is frozen (vit.py:243), so the cache
        is only invalidated when the parameter tensor is rewritten (load_state_dict bumps _version)."""
        pe = self.w["encoder.pos_embed"]
        ver = (pe._version, pe.data_ptr())
        if ver != self._pos_version:
            self._pos_cache.clear()
            self._pos_version = ver
        key = (h, w)
        if key not in self._pos_cache:
            self._pos_cache[key] = ops.pos_embed_resize(pe, self.cfg.grid, h, w)
        return self._pos_cache[key]

    def mark_grad(self, seg: int):
        self.store.seg_has_grad[self.student][seg] = True

    def w16(self, key: str, rows: int, fmt1: Optional[bool] = None):
        return self.store.w16(self.student, key, rows, fmt1)

    def w16T(self, key: str, rows: int):
        return self.store.w16T(self.student, key, rows)


# ------------------------------------------------------------------------------------------------
# forward
# ------------------------------------------------------------------------------------------------
@dataclass
class BlockSaved:
    x_in: Tensor = None
    mean1: Tensor = None
    rstd1: Tensor = None
    ln1: Tensor = None
    qkv: Tensor = None
    lse: Tensor = None
    att: Tensor = None
    x_mid: Tensor = None
    mean2: Tensor = None
    rstd2: Tensor = None
    ln2: Tensor = None
    pre1: Tensor = None
    h1: Tensor = None
    qkv16: object = None      # f16x3 mode, head dim 64: the fp16 hi / lo planes of qkv (operands of the split attention backward)
    # f16x3 mode, k-major backward (KM_BWD): the format 1 planes the forward GEMMs consumed = the B operands of the weight gradients
    ln1_16: object = None
    att16: object = None
    ln2_16: object = None
    h1_16: object = None


@dataclass
class EncoderSaved:
    B: int = 0
    h: int = 0
    w: int = 0
    x_img: Tensor = None
    blocks: List[BlockSaved] = field(default_factory=list)
    x_last: Tensor = None      # input of the final LayerNorm
    mean_f: Tensor = None
    rstd_f: Tensor = None
    guard: dict = None         # f16x3 mode: the RangeGuard verdicts this forward ran with (its backward takes the same routes)


def encoder_forward(P: StudentParams, x: Tensor, save: bool, save_rows: int = 0):
    """forward_features (vit.py:308-326).  Returns (tokens_final [B*(1+n), D], tokens_aux [B*(1+n), D], saved).
    tokens_aux = output of block `aux_layer` (un-normalised unless it is the last block).
    save_rows: only the first save_rows token rows will be back-propagated (f16x3 mode: their fp32 copies only)."""
    if GEMM_MODE == "f16x3":
        return _encoder_forward16(P, [x], save, save_rows)[0]
    cfg = P.cfg
    B, _, Himg, Wimg = x.shape
    h, w = Himg // cfg.patch, Wimg // cfg.patch
    n, D, H, hd = h * w, cfg.embed_dim, cfg.num_heads, cfg.head_dim
    N = n + 1
    W = P.w
    rows = ops.patch_im2row(x, cfg.patch)
    patch = ops.linear(rows, W["encoder.patch_embed.proj.weight"], W["encoder.patch_embed.proj.bias"])
    del rows
    t = ops.assemble_tokens(patch, W["encoder.cls_token"], P.pos_embed_for(h, w), B, n, D)
    del patch
    sv = EncoderSaved(B=B, h=h, w=w, x_img=x if save else None) if save else None
    aux_idx = cfg.aux_layer % cfg.depth
    aux = None
    scale = hd ** -0.5
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}."
        ln1, m1, r1 = ops.layernorm_fwd(t, W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.ln_eps, save)
        qkv = ops.linear(ln1, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"])
        att, lse = ops.attention_fwd(qkv, B, N, H, hd, scale, need_lse=save)
        x_mid = ops.linear(att, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"], res=t)
        ln2, m2, r2 = ops.layernorm_fwd(x_mid, W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.ln_eps, save)
        pre1 = torch.empty((t.shape[0], D * cfg.mlp_ratio), device=t.device, dtype=torch.float32) if save else None
        h1 = ops.linear(ln2, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"], gelu=True, store_pre=pre1)
        x_out = ops.linear(h1, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"], res=x_mid)
        if save:
            sv.blocks.append(BlockSaved(x_in=t, mean1=m1, rstd1=r1, ln1=ln1, qkv=qkv, lse=lse, att=att, x_mid=x_mid,
                                        mean2=m2, rstd2=r2, ln2=ln2, pre1=pre1, h1=h1))
        t = x_out
        if i == aux_idx and i != cfg.depth - 1:
            aux = t
    tf, mf, rf = ops.layernorm_fwd(t, W["encoder.norm.weight"], W["encoder.norm.bias"], cfg.ln_eps, save)
    if save:
        sv.x_last, sv.mean_f, sv.rstd_f = t, mf, rf
    if aux is None:  # aux_layer == last block -> embeds[-1] is the final LN output (vit.py:323-324)
        aux = tf
    return tf, aux, sv


def _encoder_forward16(P: StudentParams, xs, save: bool, save_rows: int = 0, whole: bool = False):
    """forward_features of ONE or SEVERAL batches (different resolutions) with every Linear on the f16x3 split GEMM.
    Token rows of all batches are concatenated (every row-wise kernel runs once over all of them, attention per batch on
    its row slice -- the merged ms-CAM pass of cam_logits_multi); with save=True (single batch only) the fp32 copies
    the hand-written backward consumes are written next to the operand planes.  Data flow per block (fp32 residual
    stream; `16` = hi / lo fp16 planes, the A operand of the next GEMM, written by the producing kernel):
        t -LN-> ln1_16 -GEMM-> qkv16 -split attention (head dim 64)-> att16 -GEMM + t-> x_mid
        x_mid -LN-> ln2_16 -GEMM, GELU-> h1_16 -GEMM + x_mid-> t
    Returns [(tokens_final, tokens_aux, saved)] per batch (row-slice views)."""
    cfg = P.cfg
    D, H, hd = cfg.embed_dim, cfg.num_heads, cfg.head_dim
    W = P.w
    P.store.ensure_w16(P.student)
    guard = P.store.guard.sites(P.student)      # which sites may run on fp16 planes (RangeGuard); the others run on f32
    # save_rows > 0 (shared ms-CAM / training pass: rows of the un-flipped images come first and only they are back-propagated):
    # the fp32 copies the backward needs are produced for those rows only -- planes, which the forward consumes, for all rows.
    # Only while every site runs on planes (an f32-routed consumer needs its fp32 input for all rows).
    if save_rows and not partial_save_ok(P, guard):
        save_rows = 0
    # several batches WITH saving (round 5: the whole ms-CAM of a step -- all scales -- and the training forward as ONE pass): only
    # the first save_rows rows (a prefix of batch 0) are recorded, which needs the partial-save route above
    assert not (save and len(xs) > 1 and not save_rows), "a merged pass can only save a row prefix of its first batch (partial_save_ok)"
    def ea(site_flags, site):             # plane format (format 1 exponent, 0 = format 0) of the activations that feed `site`'s GEMM
        return int(site_flags[site + "_f1"]) if FMT1 else 0
    toks, groups = [], []
    r0 = 0
    for x in xs:
        B, _, Himg, Wimg = x.shape
        h, w = Himg // cfg.patch, Wimg // cfg.patch
        n = h * w
        if guard["patch"]:
            e = ea(guard, "patch")
            rows16 = ops.split16(ops.patch_im2row(x, cfg.patch), exp=e)
            patch, _ = _lin16(P, rows16, P.w16("encoder.patch_embed.proj.weight", D, bool(e)), W["encoder.patch_embed.proj.bias"])
            del rows16
        else:
            patch = ops.linear(ops.patch_im2row(x, cfg.patch), W["encoder.patch_embed.proj.weight"], W["encoder.patch_embed.proj.bias"])
        toks.append(ops.assemble_tokens(patch, W["encoder.cls_token"], P.pos_embed_for(h, w), B, n, D))
        groups.append((r0, B, n + 1, h, w))
        r0 += B * (n + 1)
    t = torch.cat(toks, dim=0) if len(toks) > 1 else toks[0]
    del toks
    R = t.shape[0]
    sv = None
    if save:
        _, B, _, h, w = groups[0]
        sv = EncoderSaved(B=B, h=h, w=w, x_img=xs[0], guard=guard)
    aux_idx = cfg.aux_layer % cfg.depth
    aux = None
    scale = hd ** -0.5
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}."
        g = guard["blocks"][i]
        attn16 = hd == 64 and g["attn"]          # q, k, v as planes into the split attention kernel
        # sites whose backward reads the forward's planes k-major (format 1 planes on both sides): their input needs no fp32 copy
        km = {st_: bool(save and KM_BWD and g[st_] and ea(g, st_) > 0) for st_ in ("qkv", "proj", "fc1", "fc2")}
        ln1, ln1_16, m1, r1 = ops.layernorm_fwd16(t, W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.ln_eps, save,
                                                  want_f32=(save and not km["qkv"]) or not g["qkv"], f32_rows=save_rows,
                                                  exp=ea(g, "qkv"))
        lse = None
        # q, k, v stay fp16 planes end to end where their range allows: the qkv GEMM writes them, the split attention kernel
        # reads them and writes the planes the projection GEMM consumes; fp32 copies only where the backward (save) or an
        # f32-routed consumer needs them
        # the fp32 q / k / v exist only for consumers that cannot take planes: the f32 attention kernels (other head dims, an
        # out-of-range verdict) and the f32 attention backward of sequences beyond 2 048 tokens
        need_qkv32 = (not attn16) or (save and groups[0][2] > 2048)       # (only batch 0 is ever back-propagated)
        if g["qkv"]:
            qkv, qkv16 = _lin16(P, ln1_16, P.w16(p + "attn.qkv.weight", 3 * D, bool(ln1_16.exp)), W[p + "attn.qkv.bias"],
                                      want_f32=need_qkv32, want16=attn16)
        else:
            qkv = ops.linear(ln1, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"])
            qkv16 = ops.split16(qkv) if attn16 else None
        ln1_keep = ln1_16 if km["qkv"] else None
        del ln1_16
        need_att32 = save or not g["proj"]
        att = torch.empty((save_rows or R, D), device=t.device, dtype=torch.float32) if (need_att32 or not attn16) else None
        att16 = ops.split16_empty(R, D, t.device, ea(g, "proj")) if (attn16 and g["proj"]) else None
        lse = None
        use_segs = attn16 and 1 < len(groups) <= ops._lib.ATTN_SEGS_MAX and \
            sum(B_ * H * ((N_ + 127) // 128) for (_, B_, N_, _, _) in groups) <= ATTN_SEGS
        if use_segs:
            # every batch of the pass in one launch, longest first (dupl_attention_fwd16_segs): the short batches fill the tails
            segs = []
            for gi, (g0, B, N, _, _) in enumerate(groups):
                planes_only = bool(save_rows and gi > 0) or att is None
                bf = save_rows // N if (save_rows and gi == 0) else 0
                segs.append((g0, B, N, None if planes_only else att[g0:g0 + (bf or B) * N], bool(save and gi == 0), bf))
            lse = ops.attention_fwd16_segs(qkv16, segs, H, hd, scale, out16=att16)[0]
        for gi, (g0, B, N, _, _) in enumerate(() if use_segs else groups):
            if attn16:
                if save_rows and gi > 0:      # rows beyond the saved prefix: planes only, no fp32 output, no lse
                    ops.attention_fwd16(qkv16.rows_slice(g0, g0 + B * N), B, N, H, hd, scale, need_lse=False, out=None,
                                        out16=att16.rows_slice(g0, g0 + B * N))
                    continue
                bf = save_rows // N if save_rows else 0
                lse_g = ops.attention_fwd16(qkv16.rows_slice(g0, g0 + B * N), B, N, H, hd, scale, need_lse=save,
                                            out=att[g0:g0 + (bf or B) * N] if att is not None else None,
                                            out16=att16.rows_slice(g0, g0 + B * N) if att16 is not None else None, b_f32=bf)
            else:   # other head dims (the 96-dim test backbone) or q / k / v beyond fp16's range: exact-f32 attention kernel
                _, lse_g = ops.attention_fwd(qkv[g0:g0 + B * N], B, N, H, hd, scale, need_lse=save, out=att[g0:g0 + B * N])
            if gi == 0:
                lse = lse_g
        if not attn16 and g["proj"]:
            att16 = ops.split16(att, exp=ea(g, "proj"))
        qkv16_keep = qkv16 if (save and attn16) else None
        del qkv16
        if g["proj"]:
            x_mid, _ = _lin16(P, att16, P.w16(p + "attn.proj.weight", D, bool(att16.exp)), W[p + "attn.proj.bias"], res=t)
        else:
            x_mid = ops.linear(att, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"], res=t)
        att_keep = att16 if (km["proj"] and att16 is not None and att16.exp > 0) else None
        del att16
        ln2, ln2_16, m2, r2 = ops.layernorm_fwd16(x_mid, W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.ln_eps, save,
                                                  want_f32=(save and not km["fc1"]) or not g["fc1"], f32_rows=save_rows,
                                                  exp=ea(g, "fc1"))
        pre1 = torch.empty((save_rows or R, D * cfg.mlp_ratio), device=t.device, dtype=torch.float32) if save else None
        if g["fc1"]:
            # the planes of h1 in the format fc2 takes; format 1 planes can only come out of a format 1 GEMM
            e2 = ea(g, "fc2") if ln2_16.exp else 0
            km["fc2"] = km["fc2"] and e2 > 0
            h1, h1_16 = _lin16(P, ln2_16, P.w16(p + "mlp.fc1.weight", D * cfg.mlp_ratio, bool(ln2_16.exp)), W[p + "mlp.fc1.bias"],
                                     gelu=True, store_pre=pre1, want_f32=(save and not km["fc2"]) or not g["fc2"], want16=g["fc2"],
                                     c_rows=save_rows, out_exp=e2)
        else:
            h1 = ops.linear(ln2, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"], gelu=True, store_pre=pre1)
            h1_16 = ops.split16(h1, exp=ea(g, "fc2")) if g["fc2"] else None
            km["fc2"] = False          # h1 exists in fp32 anyway: its backward takes the transposed-planes path
        ln2_keep = ln2_16 if km["fc1"] else None
        h1_keep = h1_16 if km["fc2"] else None
        del ln2_16
        if g["fc2"]:
            x_out, _ = _lin16(P, h1_16, P.w16(p + "mlp.fc2.weight", D, bool(h1_16.exp)), W[p + "mlp.fc2.bias"], res=x_mid)
        else:
            x_out = ops.linear(h1, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"], res=x_mid)
        del h1_16
        if save:
            sv.blocks.append(BlockSaved(x_in=t, mean1=m1, rstd1=r1, ln1=ln1, qkv=qkv, lse=lse, att=att, x_mid=x_mid,
                                        mean2=m2, rstd2=r2, ln2=ln2, pre1=pre1, h1=h1, qkv16=qkv16_keep,
                                        ln1_16=ln1_keep, att16=att_keep, ln2_16=ln2_keep, h1_16=h1_keep))
        del ln1_keep, att_keep, ln2_keep, h1_keep
        t = x_out
        if i == aux_idx and i != cfg.depth - 1:
            aux = t
    tf, mf, rf = ops.layernorm_fwd(t, W["encoder.norm.weight"], W["encoder.norm.bias"], cfg.ln_eps, save)
    if save:
        sv.x_last, sv.mean_f, sv.rstd_f = t, mf, rf
    if aux is None:
        aux = tf
    outs = [(tf[g0:g0 + B * N], aux[g0:g0 + B * N], sv) for (g0, B, N, _, _) in groups]
    if whole:
        return outs, tf, aux
    return outs


def partial_save_ok(P: StudentParams, guard=None) -> bool:
    """Can a pass record activations for a row PREFIX only (the shared ms-CAM / training pass)?  Only while every site of the
    student runs on operand planes: an f32-routed consumer needs its fp32 input for all rows."""
    if GEMM_MODE != "f16x3" or P.cfg.head_dim != 64:
        return False
    guard = guard if guard is not None else P.store.guard.sites(P.student)
    return all(b[k] for b in guard["blocks"] for k in RangeGuard.SITES)


def cam_logits(P: StudentParams, x: Tensor):
    """cam_only path (model_dupl.py:81-84), token-major: returns (cam_aux_tok, cam_tok), each
    [B*(1+n), C] (row 0 of every image is the cls token and must be skipped), plus (h, w)."""
    tf, aux, _ = encoder_forward(P, x, save=False)
    Wc = P.w["classifier.weight"].view(P.num_classes - 1, -1)
    Wa = P.w["aux_classifier.weight"].view(P.num_classes - 1, -1)
    cam = ops.linear(tf, Wc)
    cam_aux = ops.linear(aux, Wa)
    return cam_aux, cam


def cam_logits_multi(P: StudentParams, xs):
    """cam_only logits of SEVERAL batches of different resolution in one encoder pass (no activation saving): the token
    rows of all batches are concatenated, so every LayerNorm / Linear runs once over sum_i B_i*(1+n_i) rows (the 0.5x
    ms-CAM scale alone is 1 576 rows = 150-600 GEMM blocks on 256 CUs; merged with the 1.5x scale it rides in a
    15 696-row GEMM) and only attention, which mixes tokens of one image, runs per batch on its row slice.  Row-wise
    ops make this bit-identical to separate passes.  Returns [(cam_aux_tok, cam_tok)] per batch (row-slice views)."""
    cfg = P.cfg
    D, H, hd = cfg.embed_dim, cfg.num_heads, cfg.head_dim
    W = P.w
    if GEMM_MODE == "f16x3":
        C = P.num_classes - 1
        outs, tf_all, aux_all = _encoder_forward16(P, list(xs), save=False, whole=True)
        # the CAM heads (N = C columns) stay on the f32 kernel: one launch over the rows of all batches (which ARE one tensor)
        cam = ops.linear(tf_all, W["classifier.weight"].view(C, -1))
        cam_aux = ops.linear(aux_all, W["aux_classifier.weight"].view(C, -1))
        res, g0 = [], 0
        for o in outs:
            r = o[0].shape[0]
            res.append((cam_aux[g0:g0 + r], cam[g0:g0 + r]))
            g0 += r
        return res
    toks, groups = [], []
    r0 = 0
    for x in xs:
        B, _, Himg, Wimg = x.shape
        h, w = Himg // cfg.patch, Wimg // cfg.patch
        n = h * w
        rows = ops.patch_im2row(x, cfg.patch)
        patch = ops.linear(rows, W["encoder.patch_embed.proj.weight"], W["encoder.patch_embed.proj.bias"])
        toks.append(ops.assemble_tokens(patch, W["encoder.cls_token"], P.pos_embed_for(h, w), B, n, D))
        groups.append((r0, B, n + 1))
        r0 += B * (n + 1)
    t = torch.cat(toks, dim=0) if len(toks) > 1 else toks[0]
    del toks
    aux_idx = cfg.aux_layer % cfg.depth
    aux = None
    scale = hd ** -0.5
    for i in range(cfg.depth):
        p = f"encoder.blocks.{i}."
        ln1, _, _ = ops.layernorm_fwd(t, W[p + "norm1.weight"], W[p + "norm1.bias"], cfg.ln_eps, False)
        qkv = ops.linear(ln1, W[p + "attn.qkv.weight"], W[p + "attn.qkv.bias"])
        att = torch.empty((t.shape[0], D), device=t.device, dtype=torch.float32)
        for (g0, B, N) in groups:
            ops.attention_fwd(qkv[g0:g0 + B * N], B, N, H, hd, scale, out=att[g0:g0 + B * N])
        x_mid = ops.linear(att, W[p + "attn.proj.weight"], W[p + "attn.proj.bias"], res=t)
        ln2, _, _ = ops.layernorm_fwd(x_mid, W[p + "norm2.weight"], W[p + "norm2.bias"], cfg.ln_eps, False)
        h1 = ops.linear(ln2, W[p + "mlp.fc1.weight"], W[p + "mlp.fc1.bias"], gelu=True)
        t = ops.linear(h1, W[p + "mlp.fc2.weight"], W[p + "mlp.fc2.bias"], res=x_mid)
        if i == aux_idx and i != cfg.depth - 1:
            aux = t
    tf, _, _ = ops.layernorm_fwd(t, W["encoder.norm.weight"], W["encoder.norm.bias"], cfg.ln_eps, False)
    if aux is None:
        aux = tf
    C = P.num_classes - 1
    cam = ops.linear(tf, W["classifier.weight"].view(C, -1))
    cam_aux = ops.linear(aux, W["aux_classifier.weight"].view(C, -1))
    return [(cam_aux[g0:g0 + B * N], cam[g0:g0 + B * N]) for (g0, B, N) in groups]


@dataclass
class HeadSaved:
    enc: EncoderSaved = None
    tf: Tensor = None
    aux: Tensor = None
    pooled: Tensor = None
    pooled_idx: Tensor = None
    pooled_aux: Tensor = None
    pooled_aux_idx: Tensor = None
    col6: Tensor = None
    h6: Tensor = None
    col7: Tensor = None
    h7: Tensor = None
    guard: dict = None         # f16x3 mode: the RangeGuard verdicts the decoder convs ran with


def _prefix_saved(sv: EncoderSaved, b: int, rows: int) -> EncoderSaved:
    """Activation record of the first `b` images of a larger saved batch (every tensor is image-major, so a row
    prefix is a contiguous view: nothing is copied)."""
    out = EncoderSaved(B=b, h=sv.h, w=sv.w, x_img=sv.x_img[:b], guard=sv.guard)
    for s in sv.blocks:
        def pre(t_):        # fp32 copies exist for the first save_rows rows only -- or not at all (k-major backward sites)
            return t_[:rows] if t_ is not None else None

        # the kept operand planes stay WHOLE (all 2b images' rows): a weight gradient walks its contraction index in 32-row
        # steps past `rows`, where its other operand holds zeros -- real rows there spare the kernel a clamp
        out.blocks.append(BlockSaved(x_in=s.x_in[:rows], mean1=s.mean1[:rows], rstd1=s.rstd1[:rows], ln1=pre(s.ln1),
                                     qkv=s.qkv[:rows] if s.qkv is not None else None, lse=s.lse[:b], att=s.att[:rows],
                                     x_mid=s.x_mid[:rows],
                                     mean2=s.mean2[:rows], rstd2=s.rstd2[:rows], ln2=pre(s.ln2), pre1=s.pre1[:rows],
                                     h1=pre(s.h1), qkv16=s.qkv16.rows_slice(0, rows) if s.qkv16 is not None else None,
                                     ln1_16=s.ln1_16, att16=s.att16, ln2_16=s.ln2_16, h1_16=s.h1_16))
    out.x_last, out.mean_f, out.rstd_f = sv.x_last[:rows], sv.mean_f[:rows], sv.rstd_f[:rows]
    return out


def cam_logits_shared(P: StudentParams, x2: Tensor, b: int):
    """cam_only logits of x2 = [x ; flip(x)] (2b images) run WITH activation saving, so that the training forward of
    the same step (same weights, same x = x2[:b]) reuses the encoder pass instead of repeating it (the reference runs
    it twice: cam_helper.py:171 under no_grad, then train_final_voc.py:204).  The whole 2b batch goes through the
    kernels in one piece (large grids); the cache handed back is the row-prefix view of the un-flipped half.
    Returns (cam_aux_tok, cam_tok) [2b*(1+n), C] and the cache (tf, aux, EncoderSaved)."""
    C = P.num_classes - 1
    Wc = P.w["classifier.weight"].view(C, -1)
    Wa = P.w["aux_classifier.weight"].view(C, -1)
    rows = (x2.shape[0] // 2) * ((x2.shape[2] // P.cfg.patch) * (x2.shape[3] // P.cfg.patch) + 1)
    tf, aux, enc = encoder_forward(P, x2, save=True, save_rows=rows)
    cam = ops.linear(tf, Wc)
    cam_aux = ops.linear(aux, Wa)
    assert rows == tf.shape[0] // 2
    return cam_aux, cam, (tf[:rows], aux[:rows], _prefix_saved(enc, b, rows))


def cam_logits_shared_multi(P: StudentParams, xs, b: int):
    """The whole encoder work of a training step in ONE pass (round 5): xs[0] = [x ; flip(x)] at scale 1.0 (2b images, run WITH
    activation saving for its un-flipped half, as cam_logits_shared) and xs[1:] the other ms-CAM scales' batches (no-grad) -- the
    token rows of all of them concatenated: at 448^2, 4 images, 6 280 + 1 576 + 14 120 = 21 976 rows per Linear instead of a
    6 280-row pass and a 15 696-row pass: half the launches, larger GEMM grids.  Row-wise kernels and per-batch attention make the
    result bit-identical to the separate passes.  Needs partial_save_ok(P).
    Returns [(cam_aux_tok, cam_tok)] per batch and the training forward's cache (tf, aux, EncoderSaved) of x = xs[0][:b]."""
    C = P.num_classes - 1
    n0 = (xs[0].shape[2] // P.cfg.patch) * (xs[0].shape[3] // P.cfg.patch) + 1
    rows = (xs[0].shape[0] // 2) * n0
    P.store.ensure_w16(P.student)         # (takes in the range guard's verdicts: the check below must see what the pass will run with)
    if not partial_save_ok(P):
        # a site is routed to the f32 kernels: they need fp32 inputs for ALL rows -- the two-pass form (scale 1.0 saved whole, the
        # other scales merged) computes the same values
        cam_aux_t, cam_t, cache = cam_logits_shared(P, xs[0], b)
        return [(cam_aux_t, cam_t)] + (cam_logits_multi(P, xs[1:]) if len(xs) > 1 else []), cache
    outs, tf, aux = _encoder_forward16(P, list(xs), save=True, save_rows=rows, whole=True)
    enc = outs[0][2]
    cam = ops.linear(tf, P.w["classifier.weight"].view(C, -1))
    cam_aux = ops.linear(aux, P.w["aux_classifier.weight"].view(C, -1))
    res, g0 = [], 0
    for o in outs:
        r = o[0].shape[0]
        res.append((cam_aux[g0:g0 + r], cam[g0:g0 + r]))
        g0 += r
    return res, (tf[:rows], aux[:rows], _prefix_saved(enc, b, rows))


def network_forward(P: StudentParams, x: Tensor, save: bool, enc_cache=None):
    """network.forward (model_dupl.py:69-106), training / val branch.
    Returns (cls_x4 (B,C), seg (B,C+1,h,w), x4 (B,D,h,w), cls_aux (B,C)), saved.
    enc_cache: (tf, aux, EncoderSaved) of THIS x from cam_logits_shared -> the encoder pass is not repeated."""
    cfg = P.cfg
    if enc_cache is not None:
        tf, aux, enc = enc_cache
    else:
        tf, aux, enc = encoder_forward(P, x, save)
    B = x.shape[0]
    h, w = x.shape[2] // cfg.patch, x.shape[3] // cfg.patch
    n, D, C = h * w, cfg.embed_dim, P.num_classes - 1
    Wc = P.w["classifier.weight"].view(C, D)
    Wa = P.w["aux_classifier.weight"].view(C, D)
    pooled, pidx = ops.gmp_fwd(tf, B, n, D)
    pooled_a, paidx = ops.gmp_fwd(aux, B, n, D)
    cls_x4 = ops.linear(pooled, Wc)
    cls_aux = ops.linear(pooled_a, Wa)
    x4 = ops.tokens_to_nchw(tf, B, n, D, h, w, skip_cls=True)
    # LargeFOV on the patch tokens (token-major, cls row skipped through the image stride)
    dd, dil = cfg.decoder_dim, cfg.decoder_dilation
    col6 = torch.empty((B * n, 9 * D), device=x.device, dtype=torch.float32)
    ops.L().dupl_im2col_dil3(tf.data_ptr() + 4 * D, col6.data_ptr(), B, h, w, D, dil, D, (n + 1) * D, ops._stream())
    f16 = GEMM_MODE == "f16x3"        # the two 3x3 convs (K = 9 * 768 / 9 * 512) as split GEMMs; conv8 (N = classes) stays f32
    guard = None
    if f16:
        P.store.ensure_w16(P.student)
        guard = P.store.guard.sites(P.student)
    if f16 and guard["conv6"]:
        h6, _ = _lin16(P, ops.split16(col6), P.w16("decoder.conv6.weight", dd), relu=True)
    else:
        h6 = ops.linear(col6, P.w["decoder.conv6.weight"].view(dd, -1), relu=True)
    col7 = torch.empty((B * n, 9 * dd), device=x.device, dtype=torch.float32)
    ops.L().dupl_im2col_dil3(h6.data_ptr(), col7.data_ptr(), B, h, w, dd, dil, dd, n * dd, ops._stream())
    if f16 and guard["conv7"]:
        h7, _ = _lin16(P, ops.split16(col7), P.w16("decoder.conv7.weight", dd), relu=True)
    else:
        h7 = ops.linear(col7, P.w["decoder.conv7.weight"].view(dd, -1), relu=True)
    seg_tok = ops.linear(h7, P.w["decoder.conv8.weight"].view(P.num_classes, dd))
    seg = ops.tokens_to_nchw(seg_tok, B, n, P.num_classes, h, w, skip_cls=False)
    sv = None
    if save:
        sv = HeadSaved(enc=enc, tf=tf, aux=aux, pooled=pooled, pooled_idx=pidx, pooled_aux=pooled_a, pooled_aux_idx=paidx,
                       col6=col6, h6=h6, col7=col7, h7=h7, guard=guard)
    return (cls_x4, seg, x4, cls_aux), sv


def large_fov_forward(x: Tensor, W6: Tensor, W7: Tensor, W8: Tensor, dil: int, keep: Optional[list] = None) -> Tensor:
    """LargeFOV.forward (conv_head.py:32-41) on a stand-alone feature map x (B, C, h, w) -> (B, classes, h, w): the decoder section
    of network_forward on the exact-f32 kernels (im2col + GEMM with fused ReLU), for callers that run `model.decoder(x4)` themselves."""
    B, C, h, w = x.shape
    n, dd, NC = h * w, W6.shape[0], W8.shape[0]
    tok = ops.zeros((B * n, C), x.device)
    ops.nchw_to_tokens_add(x, tok, B, n, C, skip_cls=False)
    col6 = torch.empty((B * n, 9 * C), device=x.device, dtype=torch.float32)
    ops.L().dupl_im2col_dil3(tok.data_ptr(), col6.data_ptr(), B, h, w, C, dil, C, n * C, ops._stream())
    h6 = ops.linear(col6, W6.reshape(dd, -1), relu=True)
    col7 = torch.empty((B * n, 9 * dd), device=x.device, dtype=torch.float32)
    ops.L().dupl_im2col_dil3(h6.data_ptr(), col7.data_ptr(), B, h, w, dd, dil, dd, n * dd, ops._stream())
    h7 = ops.linear(col7, W7.reshape(W7.shape[0], -1), relu=True)
    seg_tok = ops.linear(h7, W8.reshape(NC, -1))
    seg = ops.tokens_to_nchw(seg_tok, B, n, NC, h, w, skip_cls=False)
    if keep is not None:
        keep.extend((col6, h6, col7, h7))
    return seg


class LargeFOVFn(torch.autograd.Function):
    """The stand-alone LargeFOV head as an autograd node (round 6; conv_head.py:32-41 is an ordinary nn.Module in the reference, so
    `model.decoder(x4)` must be trainable on its own): forward = large_fov_forward, backward = the decoder section of
    network_backward on the exact-f32 kernels -- dW8 = dseg^T h7, dh7 = (dseg W8) [h7 > 0], dW7 = dh7^T col7, dh6 = col2im(dh7 W7)
    [h6 > 0], dW6 = dh6^T col6, dx = col2im(dh6 W6).  Gradients are RETURNED (torch accumulates them into .grad, which for a student's
    decoder is a view of the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, x, W6, W7, W8, dil):
        keep = []
        with torch.no_grad():
            seg = large_fov_forward(x.contiguous().float(), W6, W7, W8, dil, keep=keep)
        ctx.save_for_backward(W6, W7, W8, *keep)
        ctx.dil, ctx.xshape = dil, tuple(x.shape)
        return seg

    @staticmethod
    def backward(ctx, dseg):
        W6, W7, W8, col6, h6, col7, h7 = ctx.saved_tensors
        B, C, h, w = ctx.xshape
        n, dd, NC, dil = h * w, W6.shape[0], W8.shape[0], ctx.dil
        dev = dseg.device
        dseg_tok = ops.zeros((B * n, NC), dev)
        ops.nchw_to_tokens_add(dseg.contiguous().float(), dseg_tok, B, n, NC, skip_cls=False)
        need = ctx.needs_input_grad
        dW8 = dW7 = dW6 = dx = None
        if need[3]:
            dW8 = torch.empty_like(W8)
            ops.linear_wgrad(dseg_tok, h7, dW8)
        dh7 = ops.linear_dgrad(dseg_tok, W8.reshape(NC, dd), relumask_of=h7)
        if need[2]:
            dW7 = torch.empty_like(W7)
            ops.linear_wgrad(dh7, col7, dW7)
        dcol7 = ops.linear_dgrad(dh7, W7.reshape(dd, -1))
        dh6 = torch.empty((B * n, dd), device=dev, dtype=torch.float32)
        ops.L().dupl_col2im_dil3(dcol7.data_ptr(), dh6.data_ptr(), B, h, w, dd, dil, dd, n * dd, 0, h6.data_ptr(), ops._stream())
        del dcol7
        if need[1]:
            dW6 = torch.empty_like(W6)
            ops.linear_wgrad(dh6, col6, dW6)
        if need[0]:
            dcol6 = ops.linear_dgrad(dh6, W6.reshape(dd, -1))
            dtok = torch.empty((B * n, C), device=dev, dtype=torch.float32)
            ops.L().dupl_col2im_dil3(dcol6.data_ptr(), dtok.data_ptr(), B, h, w, C, dil, C, n * C, 0, None, ops._stream())
            dx = ops.tokens_to_nchw(dtok, B, n, C, h, w, skip_cls=False)
        return dx, dW6, dW7, dW8, None


# ------------------------------------------------------------------------------------------------
# backward
# ------------------------------------------------------------------------------------------------
def _linear_backward32(P: StudentParams, dy: Tensor, x: Tensor, name: str, dgelu_of: Optional[Tensor] = None,
                       has_bias: bool = True) -> Tensor:
    """Backward of y = x W^T + b on the exact-f32 MFMA kernels: accumulates dW, db; returns dx (* gelu'(dgelu_of))."""
    N = dy.shape[1]
    ops.linear_wgrad(dy, x, P.g[name + ".weight"], accumulate=True)
    if has_bias:
        ops.colsum(dy, P.g[name + ".bias"], accumulate=True)
    return ops.linear_dgrad(dy, P.w[name + ".weight"].view(N, -1), dgelu_of=dgelu_of)


def _linear_backward16_km(P: StudentParams, dy: Tensor, x16, name: str, dgelu_of: Optional[Tensor] = None,
                          dx_feeds_split: bool = False, wgrads: Optional[list] = None) -> Tensor:
    """Backward of y = x W^T + b on the k-major single-accumulator kernels (csrc/gemm_split.hip, gemm_f16x3_km_kernel): ONE pass
    over dy writes its scaled format 1 planes (zero rows up to the padded token count) and the bias gradient; the weight
    gradient reads dy and the forward's x planes k-major, the data gradient reads dy row-wise and the forward's W planes
    k-major.  Nothing is transposed, nothing but dy is split."""
    M, N = dy.shape
    Kp = max(96, (M + 31) // 32 * 32)           # contraction length of the weight gradient: whole k-tiles, >= the pipeline depth
    fuse_bias = not ops.deterministic()
    dy16, _, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=False, fmt1=True, rm_rows=Kp,
                                       colsum_into=P.g[name + ".bias"] if fuse_bias else None)
    gw = P.g[name + ".weight"]
    if wgrads is not None:
        # the weight gradient joins the block's grouped launch (network_backward: ops.wgrad16_group); dy16 stays alive until then
        wgrads.append((dy16, x16, gw.view(N, -1), alpha))
    else:
        _lin16(P, dy16, x16, out=gw.view(N, -1), accumulate=True, alpha=alpha, a_kmajor=True, b_kmajor=True, k_pad=Kp)
    if not fuse_bias:
        ops.colsum(dy, P.g[name + ".bias"], accumulate=True)
    W16 = P.w16(name + ".weight", N, True)
    if SK_DGRAD and dgelu_of is None and not dx_feeds_split and not ops.deterministic() and W16.cols <= SK_DGRAD_MAX_COLS:
        # plain dx = alpha dy . W whose consumer reads fp32 (LayerNorm backward): stream-K into a zero-filled dx -- the narrow
        # outputs (768 columns = 78 tiles of 256 x 128 at 4 images, 42 at 2) then occupy every CU
        # cleared again by the LayerNorm backward that consumes it (DUPL_ZERO_WS=0: a fresh tensor and a fill launch per use)
        dx = ops.zero_workspace(M, W16.cols, dy.device) if ZERO_WS else ops.zeros((M, W16.cols), dy.device)
        _lin16(P, dy16.rows_slice(0, M), W16, out=dx, accumulate=True, alpha=alpha, b_kmajor=True)
        return dx
    dx, _ = _lin16(P, dy16.rows_slice(0, M), W16, alpha=alpha, dgelu_of=dgelu_of, amax_for_next=dx_feeds_split, b_kmajor=True)
    return dx


def _linear_backward16(P: StudentParams, dy: Tensor, x: Tensor, name: str, dgelu_of: Optional[Tensor] = None,
                       has_bias: bool = True, dx_feeds_split: bool = False, xT16=None) -> Tensor:
    """The same on the f16x3 split GEMM (fp32-equivalent).  The gradient dy is scaled by a power of two from its own
    max-abs before it is split (its values are far below fp16's normal range); both GEMMs run as k-contiguous products
    through transposed operand planes:  dW += dy^T16 . (x^T16)^T,  dx = dy16 . (W^T16)^T  (csrc/split_prep.hip)."""
    M, N = dy.shape
    Mp = (M + 31) // 32 * 32
    fuse_bias = has_bias and not ops.deterministic()         # the bias gradient rides in the split pass (fp32 atomics)
    dy16, dyT16, alpha = ops.split_prepare(dy, scaled=True, want_rm=True, want_T=True, rows_pad=Mp,
                                           colsum_into=P.g[name + ".bias"] if fuse_bias else None)
    if xT16 is None:         # (the transformer blocks prepare x^T and W^T of all four Linears in one launch: _block_operands16)
        _, xT16, _ = ops.split_prepare(x, scaled=False, want_rm=False, want_T=True, rows_pad=Mp)
    gw = P.g[name + ".weight"]
    _lin16(P, dyT16, xT16, out=gw.view(N, -1), accumulate=True, alpha=alpha)
    if has_bias and not fuse_bias:
        ops.colsum(dy, P.g[name + ".bias"], accumulate=True)
    # dx_feeds_split: dx is the next Linear's dy -> its max-abs comes out of this epilogue (ops.reserve_amax)
    dx, _ = _lin16(P, dy16, P.w16T(name + ".weight", N), alpha=alpha, dgelu_of=dgelu_of, amax_for_next=dx_feeds_split)
    return dx


BLOCK_OPERANDS_MULTI = os.environ.get("DUPL_BLOCK_OPERANDS_MULTI", "1") != "0"


def _block_operands16(P: StudentParams, p: str, s: "BlockSaved", g: dict, D: int, Hd: int) -> dict:
    """The unscaled operands of one transformer block's f16x3 backward in ONE launch (ops.split_prepare_multi): x^T planes of
    the four Linear inputs (weight gradients) and, where stale, the W^T planes (data gradients).  Between the persistent
    GEMMs -- which own every CU's LDS and registers -- each of these short kernels runs alone on the chip, so eight
    launches of ~10 us become one of ~35.  Returns {site: x^T planes}."""
    sites = [("fc2", s.h1, "mlp.fc2", D, s.h1_16), ("fc1", s.ln2, "mlp.fc1", Hd, s.ln2_16), ("proj", s.att, "attn.proj", D, s.att16),
             ("qkv", s.ln1, "attn.qkv", 3 * D, s.ln1_16)]
    sites = [t[:4] for t in sites if g[t[0]] and t[4] is None]      # sites with kept planes run k-major: nothing to prepare
    if not sites:
        return {}
    items = []
    for _, x, _, _ in sites:
        items.append((x, False, True, (x.shape[0] + 31) // 32 * 32))
    stale = P.store.w16T_missing(P.student, [(p + nm + ".weight", N) for _, _, nm, N in sites])
    for _, rows, w in stale:
        items.append((w, False, True, rows))
    out = ops.split_prepare_multi(items)
    for (key, _, _), (_, T) in zip(stale, out[len(sites):]):
        P.store.w16T_put(P.student, key, T)
    return {site: T for (site, _, _, _), (_, T) in zip(sites, out)}


def network_backward(P: StudentParams, sv: HeadSaved, dcls: Optional[Tensor], dseg: Optional[Tensor],
                     dx4: Optional[Tensor], dcls_aux: Optional[Tensor], on_ready=None):
    """Adjoint of network_forward; accumulates into P.g[...] (the flat gradient buffer).
    on_ready(event): called (host side, stream order) when a part of this student's gradient is final: "heads" after
    the classifier / decoder gradients, i after transformer block i, "stem" at the end (FlatStorage.grad_buckets)."""
    cfg = P.cfg
    enc = sv.enc
    B, h, w = enc.B, enc.h, enc.w
    n, D, C, NC = h * w, cfg.embed_dim, P.num_classes - 1, P.num_classes
    N = n + 1
    dev = sv.tf.device
    G, W = P.g, P.w
    dtf = ops.zeros((B * N, D), dev)
    dta = None
    aux_is_final = sv.aux.data_ptr() == sv.tf.data_ptr()   # aux_layer == last block: embeds[-1] is the final LN output
    # ---- heads
    if dx4 is not None:
        ops.nchw_to_tokens_add(dx4.contiguous(), dtf, B, n, D, skip_cls=True)
    if dcls is not None:
        dcls = dcls.contiguous()
        ops.linear_wgrad(dcls, sv.pooled, G["classifier.weight"], accumulate=True)
        dpool = ops.linear_dgrad(dcls, W["classifier.weight"].view(C, D))
        ops.gmp_bwd(dpool, sv.pooled_idx, dtf, B, n, D)
        P.mark_grad(SEG_CLS)
    if dcls_aux is not None:
        dcls_aux = dcls_aux.contiguous()
        ops.linear_wgrad(dcls_aux, sv.pooled_aux, G["aux_classifier.weight"], accumulate=True)
        dpool = ops.linear_dgrad(dcls_aux, W["aux_classifier.weight"].view(C, D))
        if aux_is_final:
            ops.gmp_bwd(dpool, sv.pooled_aux_idx, dtf, B, n, D)
        else:
            dta = ops.zeros((B * N, D), dev)
            ops.gmp_bwd(dpool, sv.pooled_aux_idx, dta, B, n, D)
        P.mark_grad(SEG_CLS)
    # ---- decoder
    if dseg is not None:
        dd, dil = cfg.decoder_dim, cfg.decoder_dilation
        dseg_tok = ops.zeros((B * n, NC), dev)
        ops.nchw_to_tokens_add(dseg.contiguous(), dseg_tok, B, n, NC, skip_cls=False)
        ops.linear_wgrad(dseg_tok, sv.h7, G["decoder.conv8.weight"], accumulate=True)
        dh7 = ops.linear_dgrad(dseg_tok, W["decoder.conv8.weight"].view(NC, dd), relumask_of=sv.h7)
        f16 = GEMM_MODE == "f16x3"
        guard = (sv.guard or P.store.guard.sites(P.student)) if f16 else None
        conv_bwd = _linear_backward16 if (f16 and guard["conv7"]) else _linear_backward32
        dcol7 = conv_bwd(P, dh7, sv.col7, "decoder.conv7", has_bias=False)
        dh6 = torch.empty((B * n, dd), device=dev, dtype=torch.float32)
        ops.L().dupl_col2im_dil3(dcol7.data_ptr(), dh6.data_ptr(), B, h, w, dd, dil, dd, n * dd, 0, sv.h6.data_ptr(), ops._stream())
        del dcol7
        conv_bwd = _linear_backward16 if (f16 and guard["conv6"]) else _linear_backward32
        dcol6 = conv_bwd(P, dh6, sv.col6, "decoder.conv6", has_bias=False)
        ops.L().dupl_col2im_dil3(dcol6.data_ptr(), dtf.data_ptr() + 4 * D, B, h, w, D, dil, D, N * D, 1, None, ops._stream())
        del dcol6
        P.mark_grad(SEG_DECODER)
    if on_ready is not None:
        on_ready("heads")
    encoder_backward(P, enc, dtf, dta, on_ready=on_ready)


def encoder_backward(P: StudentParams, enc: "EncoderSaved", dtf: Tensor, dta: Optional[Tensor], on_ready=None):
    """Adjoint of encoder_forward (autograd of forward_features, vit.py:308-326): dtf [B*(1+n), D] = gradient of the final-LayerNorm
    tokens, dta = gradient of the aux tokens (un-normalised output of block `aux_layer`; None when nothing reached them or when
    aux_layer is the last block -- their gradient is part of dtf then).  Accumulates into the flat gradient buffer."""
    cfg = P.cfg
    B, h, w = enc.B, enc.h, enc.w
    n, D = h * w, cfg.embed_dim
    N = n + 1
    G, W = P.g, P.w
    # ---- final LayerNorm
    f16 = GEMM_MODE == "f16x3"
    gb = (enc.guard or P.store.guard.sites(P.student))["blocks"] if f16 else None
    # a gradient whose next use is the scaled operand split of an f16x3 Linear backward gets its max-abs from the kernel that
    # writes it (LayerNorm backward, dgrad epilogue) instead of a pass of its own (ops.reserve_amax)
    dx = ops.layernorm_bwd(dtf, enc.x_last, W["encoder.norm.weight"], enc.mean_f, enc.rstd_f,
                           G["encoder.norm.weight"], G["encoder.norm.bias"], amax_for_next=f16 and gb[cfg.depth - 1]["fc2"])
    aux_idx = cfg.aux_layer % cfg.depth
    Hh, hd = cfg.num_heads, cfg.head_dim
    scale = hd ** -0.5
    for i in reversed(range(cfg.depth)):
        p = f"encoder.blocks.{i}."
        s = enc.blocks[i]
        if dta is not None and i == aux_idx:
            ops.axpy_(dx, dta, 1.0)
            dx._dupl_amax = None         # modified after its producer took the max
        g = gb[i] if f16 else None

        def lin_bwd(site):      # the site's backward runs where its forward ran: same operands, same range verdict
            return _linear_backward16 if (f16 and g[site]) else _linear_backward32

        # the block's weight gradients (18 .. 72 tiles of 256 x 128 each) are collected and leave as ONE launch after its data
        # gradients: whole tiles over the whole token axis, no split-K, no atomics (ops.wgrad16_group)
        wg = [] if WGRAD_GROUP else None

        def lin(site, dy_, x_, x16_, nm, **kw):
            if f16 and g[site] and x16_ is not None:          # forward planes kept: k-major backward
                return _linear_backward16_km(P, dy_, x16_, nm, dgelu_of=kw.get("dgelu_of"), dx_feeds_split=kw.get("dx_feeds_split", False),
                                             wgrads=wg)
            return lin_bwd(site)(P, dy_, x_, nm, **kw)

        def feeds(site):        # does the tensor go into a scaled split next (= is `site`'s backward an f16x3 one)?
            return {"dx_feeds_split": bool(g[site])} if f16 else {}
        att16 = f16 and s.qkv16 is not None and N <= 2048
        xT = _block_operands16(P, p, s, g, D, D * cfg.mlp_ratio) if (f16 and BLOCK_OPERANDS_MULTI) else {}

        def pre(site):          # x^T planes prepared above (f16x3 sites only)
            return {"xT16": xT[site]} if site in xT else {}
        # MLP
        dpre1 = lin("fc2", dx, s.h1, s.h1_16, p + "mlp.fc2", dgelu_of=s.pre1, **(feeds("fc1") if g and g["fc2"] else {}), **pre("fc2"))
        dln2 = lin("fc1", dpre1, s.ln2, s.ln2_16, p + "mlp.fc1", **pre("fc1"))
        del dpre1
        dx_mid = ops.layernorm_bwd(dln2, s.x_mid, W[p + "norm2.weight"], s.mean2, s.rstd2,
                                   G[p + "norm2.weight"], G[p + "norm2.bias"], dres=dx, amax_for_next=f16 and g["proj"])
        # attention
        datt = lin("proj", dx_mid, s.att, s.att16, p + "attn.proj",
                   **({"dx_feeds_split": bool(att16)} if g and g["proj"] else {}), **pre("proj"))
        if att16:
            dqkv = ops.attention_bwd16(s.qkv16, s.att, datt, s.lse, B, N, Hh, hd, scale, amax_for_next=bool(g["qkv"]))
        else:
            dqkv = ops.attention_bwd(s.qkv, s.att, datt, s.lse, B, N, Hh, hd, scale)
        dln1 = lin("qkv", dqkv, s.ln1, s.ln1_16, p + "attn.qkv", **pre("qkv"))
        del dqkv, datt, xT
        dx = ops.layernorm_bwd(dln1, s.x_in, W[p + "norm1.weight"], s.mean1, s.rstd1,
                               G[p + "norm1.weight"], G[p + "norm1.bias"], dres=dx_mid,
                               amax_for_next=f16 and i > 0 and gb[i - 1]["fc2"])
        if wg:
            # (measured and dropped: this launch on a side stream of its student, to share the chip with the narrow data-gradient
            # launches of the next block -- 55.2 vs 54.4 ms at 4 img/GPU, 33.8 vs 32.6 at 2, same box: DESIGN 6)
            ops.wgrad16_group(wg, tuning=P.store.gemm16_tuning)
        del wg
        enc.blocks[i] = None  # release activations as we go
        if on_ready is not None:
            on_ready(i)
    # ---- token assembly / patch embed
    dpatch = ops.assemble_tokens_bwd(dx, G["encoder.cls_token"], B, n, D)
    rows = ops.patch_im2row(enc.x_img, cfg.patch)
    ops.linear_wgrad(dpatch, rows, G["encoder.patch_embed.proj.weight"], accumulate=True)
    ops.colsum(dpatch, G["encoder.patch_embed.proj.bias"], accumulate=True)
    P.mark_grad(SEG_BACKBONE)
    P.mark_grad(SEG_NORM)
    if on_ready is not None:
        on_ready("stem")
