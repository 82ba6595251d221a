"""Tensor-level wrappers over the C ABI (no autograd here; see engine.py).

Every function takes CUDA(ROCm) fp32 tensors, checks layout, and enqueues the HIP kernel on the
current torch stream.  Outputs are freshly allocated unless an `out=` is given.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional, Sequence

import numpy as np
import torch

from . import _lib
from ._lib import GemmDesc

Tensor = torch.Tensor


def L():
    return _lib.lib()


# The caller-side determinism switch (dupl_amd.set_deterministic / DUPL_DETERMINISTIC=1).  The library keeps no mode (ABI 3): every
# call that could accumulate with fp32 atomics is handed this value as its `deterministic` argument / descriptor field.
_DETERMINISTIC = [os.environ.get("DUPL_DETERMINISTIC", "0") == "1"]


def deterministic() -> bool:
    return _DETERMINISTIC[0]


def set_deterministic(on) -> None:
    _DETERMINISTIC[0] = bool(on)


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: Tensor, dtype=torch.float32):
    assert t.is_cuda, "dupl_amd ops need device tensors (no CPU fallback)"
    assert t.dtype == dtype, f"expected {dtype}, got {t.dtype}"
    assert t.is_contiguous(), "expected a contiguous tensor"
    return t


# ------------------------------------------------------------------------------------------ GEMM
# per-call launch tuning of dupl_gemm_f32 (dupl_gemm_desc.tile_rows / tile_cols / group; 0 = the library's heuristic): the Python
# caller's defaults, not library state
GEMM32_TUNING = {"tile_rows": 0, "tile_cols": 0, "group": 0}


def gemm_raw(A: int, B: int, C: int, M: int, N: int, K: int, lda: int, ldb: int, ldc: int, *, flags: int = 0,
             bias: Optional[int] = None, res: Optional[int] = None, ldr: int = 0, aux: Optional[int] = None,
             ldaux: int = 0, alpha: float = 1.0, batch: int = 1, zdiv: int = 1,
             sA=(0, 0), sB=(0, 0), sC=(0, 0), sR=(0, 0), sX=(0, 0), sBias=(0, 0)):
    """Pointer-level GEMM; strides are in elements.  z -> (z // zdiv, z % zdiv)."""
    d = GemmDesc()
    d.A, d.B, d.C = A, B, C
    d.bias, d.res, d.aux = bias, res, aux
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ldr, d.ldaux = lda, ldb, ldc, ldr, ldaux
    d.batch, d.zdiv = batch, zdiv
    d.sA0, d.sA1 = sA
    d.sB0, d.sB1 = sB
    d.sC0, d.sC1 = sC
    d.sR0, d.sR1 = sR
    d.sX0, d.sX1 = sX
    d.sBias0, d.sBias1 = sBias
    d.alpha, d.flags = alpha, flags
    d.deterministic = int(deterministic())
    d.tile_rows, d.tile_cols, d.group = GEMM32_TUNING["tile_rows"], GEMM32_TUNING["tile_cols"], GEMM32_TUNING["group"]
    L().dupl_gemm_f32(ctypes.byref(d), _stream())


# ------------------------------------------------------------------------------------------ f16x3 split GEMM
class Split16:
    """A matrix as two fp16 planes: the operand format of dupl_gemm_f16x3 (csrc/gemm_split.hip).
    `planes` is one [2, rows, cols] fp16 tensor (plane 0 = hi, plane 1 = lo).
    exp = 0: format 0, x = hi + lo / 2048 (two accumulator sets in the GEMM).  exp = s > 0: format 1, the planes hold
    X = x * 2^s as hi + lo with an UNSCALED lo -- all three products share one accumulator (256 x 256 tiles); both operands
    of a GEMM must be in the same format (csrc/common.h split_f32_u, dupl_gemm16_desc.fmt).
    fmt: 1 = format 1 (implied by exp > 0; scaled gradient planes are format 1 with exp 0: their power-of-two scale travels
    through device memory, split_prepare).  valid_rows: rows that hold data (the rest are zero padding)."""
    __slots__ = ("planes", "rows", "cols", "exp", "fmt", "valid_rows")

    def __init__(self, planes: Tensor, exp: int = 0, fmt: Optional[int] = None, valid_rows: Optional[int] = None):
        assert planes.dtype == torch.float16 and planes.dim() == 3 and planes.shape[0] == 2 and planes.is_contiguous()
        self.planes, self.rows, self.cols, self.exp = planes, planes.shape[1], planes.shape[2], int(exp)
        self.fmt = int(fmt) if fmt is not None else int(exp > 0)
        self.valid_rows = int(valid_rows) if valid_rows is not None else self.rows

    @property
    def hi(self) -> int:
        return self.planes.data_ptr()

    @property
    def lo(self) -> int:
        return self.planes.data_ptr() + 2 * self.rows * self.cols

    def rows_slice(self, r0: int, r1: int) -> "Split16View":
        return Split16View(self, r0, r1)


class Split16View:
    """Row range [r0, r1) of a Split16 (both planes), as an A operand."""
    __slots__ = ("hi", "lo", "rows", "cols", "base", "exp", "fmt")

    def __init__(self, base: Split16, r0: int, r1: int):
        self.base = base
        self.exp = base.exp
        self.fmt = base.fmt
        self.hi = base.hi + 2 * r0 * base.cols
        self.lo = base.lo + 2 * r0 * base.cols
        self.rows, self.cols = r1 - r0, base.cols


class W16:
    """Raw operand planes (pointers) of a [rows, cols] matrix living in someone else's buffer (parameter planes)."""
    __slots__ = ("hi", "lo", "rows", "cols", "exp", "fmt")

    def __init__(self, hi: int, lo: int, rows: int, cols: int, exp: int = 0):
        self.hi, self.lo, self.rows, self.cols, self.exp = hi, lo, rows, cols, int(exp)
        self.fmt = int(exp > 0)


# Launch tuning of the split GEMM (dupl_gemm16_desc.tile / concurrency / persist_blocks / group): per-call fields of the
# descriptor since ABI 2.  This module-level dict holds the DEFAULTS (environment) for direct callers of linear16 (tests, tools); a
# model carries its own copy (engine.FlatStorage.gemm16_tuning, handed to linear16 as `tuning`) -- model code never writes here.
# concurrency: streams that issue split GEMMs at a time (siamese_network.enable_dual_stream sets 2 on ITS model); tile: DUPL_GEMM16_TILE.
import os as _os
GEMM16_TUNING = {"tile": int(_os.environ.get("DUPL_GEMM16_TILE", "0")), "concurrency": 1,
                 "persist_blocks": int(_os.environ.get("DUPL_PERSIST_BLOCKS", "0")), "group": 0,
                 # stream-K data / weight gradients: 0 = aligned k-slices chosen by the library, n > 0 = n slices, -1 = round 4's equal runs
                 "sk_slices": int(_os.environ.get("DUPL_SK_SLICES", "0"))}


# format 1 scales (powers of two): activations * 2^3, weights * 2^9 -- typical |x| ~ 1 and |w| ~ 0.02 both land near 8 .. 10,
# where lo = X - hi (~2^-12 X) is a normal fp16; the range guard (engine.RangeGuard) checks bound * 2^exp <= 65504 / margin
EXP_ACT, EXP_W = 3, 9


def split16_empty(rows: int, cols: int, device, exp: int = 0) -> Split16:
    return Split16(torch.empty((2, rows, cols), device=device, dtype=torch.float16), exp)


def split16(x: Tensor, out: Optional[Split16] = None, exp: int = 0) -> Split16:
    """fp32 [rows, cols] (contiguous, cols % 4 == 0) -> hi / lo planes (format 0, or format 1 of x * 2^exp)."""
    _chk(x)
    rows, cols = x.shape[0], x.numel() // x.shape[0]
    out = out if out is not None else split16_empty(rows, cols, x.device, exp)
    assert out.exp == exp
    if exp:
        L().dupl_split_f16x2b(x.data_ptr(), out.hi, out.lo, x.numel(), exp, _stream())
    else:
        L().dupl_split_f16x2(x.data_ptr(), out.hi, out.lo, x.numel(), _stream())
    return out


_SCALE_RINGS = {}
_RING = 128


def _scale_ring(device):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ring = _SCALE_RINGS.get(key)
    if ring is None:
        # [records, next index, pending producer token, scaled splits so far (slot generation)]
        ring = [torch.zeros((_RING, 4), device=device, dtype=torch.float32), 0, None, 0]
        _SCALE_RINGS[key] = ring
    return ring


def _scale_slot(device):
    """Next slot of this stream's ring of {scale, 1/scale, amax word, -} records (zero-initialised once; every scaled
    split_prepare zeroes the amax word of its successor, so the ring is self-cleaning in stream order)."""
    ring = _scale_ring(device)
    buf, i = ring[0], ring[1]
    ring[1] = (i + 1) % _RING
    ring[3] += 1
    base = buf.data_ptr()
    return base + 16 * i, base + 16 * ((i + 1) % _RING) + 8, buf[i]


class _Alpha(int):
    """Device pointer of a scale record's 1 / scale, with what it takes to use it safely: the record lives in a ring of _RING
    slots of ONE stream and is rewritten _RING scaled splits later, so consumers check() that they run on that stream and
    that the slot has not been handed out again."""

    def bind(self, ring, stream):
        self.ring, self.stream, self.gen = ring, stream, ring[3]
        return self

    def check(self):
        cur = torch.cuda.current_stream().cuda_stream
        assert cur == self.stream, "a scaled operand's alpha is only valid on the stream that prepared it"
        assert self.ring[3] - self.gen < _RING - 1, "scale slot reused: the alpha of a scaled split_prepare expires after ~128 later ones"


class _AmaxToken:
    __slots__ = ("word",)


def reserve_amax(device):
    """For a kernel that is about to PRODUCE a tensor whose next use on this stream is a scaled split_prepare: the amax word
    of the slot that split will take, and a token.  The producer raises the word to max |tensor| (dupl_gemm16_desc.amax_out,
    dupl_layernorm_bwd); tag the tensor with `t._dupl_amax = token` and split_prepare skips its own amax pass.  Returns
    (None, None) while an earlier reservation is still open (one producer per slot).  A tagged tensor that is modified
    in place afterwards must drop the tag (`t._dupl_amax = None`); an unclaimed or stale word is cleared by the next
    scaled split on the stream (amax_mode 2)."""
    ring = _scale_ring(device)
    if ring[2] is not None:
        return None, None
    tok = _AmaxToken()
    tok.word = ring[0].data_ptr() + 16 * ring[1] + 8
    ring[2] = tok
    return tok.word, tok


def split_prepare(x: Tensor, scaled: bool, want_rm: bool, want_T: bool, rows_pad: int = 0, target_exp: int = 15,
                  colsum_into: Optional[Tensor] = None, fmt1: bool = False, rm_rows: int = 0):
    """Backward-path operand preparation (csrc/split_prep.hip): fp32 [R, C] -> row-major planes [R, C] and / or transposed
    planes [C, Rp] (Rp = rows_pad >= R, zero-filled), optionally scaled by the power of two that brings max|x| into
    [2^14, 2^15) (gradients).  fmt1: format 1 planes (unscaled lo: the single-accumulator k-major backward GEMMs).
    rm_rows > R: the row-major planes get rm_rows rows, the extra ones zeros (k-major A operand of a weight gradient).
    Returns (rm Split16 | None, T Split16 | None, alpha: int device pointer of 1 / scale | None;
    valid for the next ~128 scaled calls on this stream)."""
    _chk(x)
    R, C = x.shape
    Rp = rows_pad if rows_pad else (R + 31) // 32 * 32
    rm = None
    if want_rm:
        rm = Split16(torch.empty((2, max(R, rm_rows), C), device=x.device, dtype=torch.float16), 0, fmt=int(fmt1), valid_rows=R)
    T = Split16(torch.empty((2, C, Rp), device=x.device, dtype=torch.float16), 0, fmt=int(fmt1)) if want_T else None
    slot = nxt = rec = None
    amax_mode = 0
    if scaled:
        ring = _scale_ring(x.device)
        pending, ring[2] = ring[2], None
        if pending is not None:      # a producer wrote into this slot's amax word: x's own (1: no amax pass) or not (2: clear)
            amax_mode = 1 if getattr(x, "_dupl_amax", None) is pending else 2
        slot, nxt, rec = _scale_slot(x.device)
        assert pending is None or pending.word == slot + 8
    # colsum_into: [C] fp32 accumulator that receives the column sums of x (a Linear's bias gradient) from the same pass
    d = _lib.SplitDesc()
    d.x, d.ld, d.R, d.C = x.data_ptr(), x.stride(0), R, C
    d.slot, d.next_bits = slot, nxt
    d.hi, d.lo = (rm.hi, rm.lo) if rm else (None, None)
    d.hiT, d.loT = (T.hi, T.lo) if T else (None, None)
    d.Rp, d.target_exp, d.colsum_accum, d.amax_mode = Rp, target_exp, _p(colsum_into), amax_mode
    d.fmt, d.rows_zero_to = int(fmt1), (rm_rows if (rm is not None and rm_rows > R) else 0)
    d.deterministic = int(deterministic())
    L().dupl_split_prepare(ctypes.byref(d), _stream())
    if scaled:
        for o in (rm, T):
            if o is not None:
                o.planes._dupl_scale = rec      # view of the ring record {scale, 1 / scale, ...} (tests read it)
        return rm, T, _Alpha(slot + 4).bind(ring, torch.cuda.current_stream(x.device).cuda_stream)
    return rm, T, None


def split_prepare_multi(items):
    """Unscaled split_prepare of several matrices in as few launches as possible (dupl_split_prepare_multi, 16 per launch).
    items: [(x fp32 [R, C], want_rm, want_T, rows_pad)]; returns [(rm Split16 | None, T Split16 | None)]."""
    out, descs = [], []
    for x, want_rm, want_T, rows_pad in items:
        _chk(x)
        R, C = x.shape
        Rp = rows_pad if rows_pad else (R + 31) // 32 * 32
        rm = split16_empty(R, C, x.device) if want_rm else None
        T = split16_empty(C, Rp, x.device) if want_T else None
        d = _lib.SplitItem()
        d.x, d.ld, d.R, d.C, d.Rp = x.data_ptr(), x.stride(0), R, C, Rp
        d.hi, d.lo = (rm.hi, rm.lo) if rm else (None, None)
        d.hiT, d.loT = (T.hi, T.lo) if T else (None, None)
        descs.append(d)
        out.append((rm, T))
    for i in range(0, len(descs), _lib.SPLIT_MULTI_MAX):
        chunk = descs[i:i + _lib.SPLIT_MULTI_MAX]
        arr = (_lib.SplitItem * len(chunk))(*chunk)
        L().dupl_split_prepare_multi(ctypes.cast(arr, ctypes.c_void_p), len(chunk), _stream())
    return out


def linear16(x, W: Split16, bias: Optional[Tensor] = None, *, gelu: bool = False, relu: bool = False,
             res: Optional[Tensor] = None, out: Optional[Tensor] = None, store_pre: Optional[Tensor] = None,
             want_f32: bool = True, out16: Optional[Split16] = None, want16: bool = False, device=None,
             alpha: Optional[int] = None, accumulate: bool = False, dgelu_of: Optional[Tensor] = None,
             relumask_of: Optional[Tensor] = None, c_rows: int = 0, amax_for_next: bool = False, out_exp: int = 0,
             a_kmajor: bool = False, b_kmajor: bool = False, k_pad: int = 0, post_exp: int = 0, tuning: Optional[dict] = None):
    """y = act(alpha * x W^T + bias) (+ res) on the f16x3 split GEMM.  x: Split16 / Split16View [M, K]; W: Split16 [N, K].
    alpha: device pointer of a float (inverse scale of scaled gradient planes).  accumulate: out += alpha * x W^T (weight
    gradients; split-K).  dgelu_of / relumask_of: multiply by gelu'(pre) / (post > 0) (data gradients through an activation).
    amax_for_next: y's next use on this stream is a scaled split_prepare -- the epilogue leaves max |y| in that split's
    slot (reserve_amax) and y is tagged, so the split needs no amax pass.
    a_kmajor / b_kmajor: the operand is stored [K, rows] (rows contiguous, dupl_gemm16_desc.a_layout / b_layout): the backward
    GEMMs on the forward's own planes.  k_pad: the contraction length the kernel walks (a multiple of 32, >= 96; operand rows
    beyond their own count are clamped, the other operand holds zeros there).  post_exp: the product carries 2^post_exp beyond the
    operands' exp (scaled gradient planes carry theirs in alpha).
    Returns (y fp32 [M, N] or None, y as Split16 or None)."""
    M, Ka = (x.cols, x.rows) if a_kmajor else (x.rows, x.cols)
    N, Kb = (W.cols, W.rows) if b_kmajor else (W.rows, W.cols)
    if a_kmajor or b_kmajor:
        K = k_pad if k_pad else max(Ka, Kb)
        assert K % 32 == 0 and K >= 96 and (a_kmajor or Ka == K) and (b_kmajor or Kb == K), (Ka, Kb, K)
    else:
        K = Ka
        assert Kb == K and K % 32 == 0
    if dgelu_of is not None or relumask_of is not None:
        assert store_pre is None
        store_pre = dgelu_of if dgelu_of is not None else relumask_of
    dev = device if device is not None else (x.planes.device if isinstance(x, Split16) else x.base.planes.device)
    y = None
    # c_rows > 0: the fp32 outputs (y, store_pre) exist for the first c_rows rows only (the planes for all M)
    if want_f32 or out is not None:
        y = out if out is not None else torch.empty((c_rows or M, N), device=dev, dtype=torch.float32)
    y16 = out16 if out16 is not None else (split16_empty(M, N, dev, out_exp) if want16 else None)
    d = _lib.Gemm16Desc()
    d.deterministic = int(deterministic())
    # operand format: both format 0, or both format 1 (then the product carries 2^(xe + we), taken out in the epilogue)
    xe, we = getattr(x, "exp", 0), getattr(W, "exp", 0)
    xf, wf = getattr(x, "fmt", int(xe > 0)), getattr(W, "fmt", int(we > 0))
    assert xf == wf, f"operand planes in different formats (fmt {xf} / {wf}, exp {xe} / {we})"
    if xf:
        assert (a_kmajor or b_kmajor) or (not accumulate and not amax_for_next)
        assert not (accumulate and b_kmajor and not a_kmajor and deterministic()), "stream-K data gradients use fp32 atomics"
        d.fmt, d.post_scale = 1, 2.0 ** -(xe + we + post_exp)
    if a_kmajor or b_kmajor:
        assert xf == 1, "k-major operands need format 1 planes"
        d.a_layout, d.b_layout = int(a_kmajor), int(b_kmajor)
        d.ka_valid = min(Ka, K) if a_kmajor else 0
        d.kb_valid = min(Kb, K) if b_kmajor else 0
    if y16 is not None:
        assert y16.exp == 0 or xe, "format 1 output planes come from format 1 GEMMs"
        d.out_exp = y16.exp
    d.A_hi, d.A_lo, d.B_hi, d.B_lo = x.hi, x.lo, W.hi, W.lo
    d.C = _p(y)
    d.C_hi, d.C_lo = (y16.hi, y16.lo) if y16 is not None else (None, None)
    d.bias, d.res, d.aux = _p(bias), _p(res), _p(store_pre)
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb = x.cols, W.cols
    d.ldc = y.stride(0) if y is not None else 0
    d.ldo = N
    d.ldr = res.stride(0) if res is not None else 0
    d.ldaux = store_pre.stride(0) if store_pre is not None else 0
    aux_flag = _lib.GEMM_MUL_DGELU if dgelu_of is not None else (_lib.GEMM_MUL_RELUMASK if relumask_of is not None else
                                                                  (_lib.GEMM_STORE_PRE if store_pre is not None else 0))
    d.flags = (_lib.GEMM_GELU if gelu else 0) | (_lib.GEMM_RELU if relu else 0) | aux_flag | (_lib.GEMM_ACCUM if accumulate else 0)
    if isinstance(alpha, _Alpha):
        alpha.check()
    d.alpha_dev = int(alpha) if alpha is not None else None
    d.c_rows = int(c_rows)
    tn = tuning if tuning is not None else GEMM16_TUNING     # (engine.py passes its model's FlatStorage.gemm16_tuning)
    d.tile, d.concurrency = tn["tile"], tn["concurrency"]
    d.persist_blocks, d.group = tn["persist_blocks"], tn["group"]
    d.sk_slices = tn["sk_slices"]
    tok = None
    if amax_for_next and y is not None and not accumulate and not c_rows:
        d.amax_out, tok = reserve_amax(dev)
    if c_rows:
        assert y16 is not None and (y is None or y.shape[0] >= c_rows) and (store_pre is None or store_pre.shape[0] >= c_rows)
    L().dupl_gemm_f16x3(ctypes.byref(d), _stream())
    if tok is not None:
        y._dupl_amax = tok
    return y, y16


def wgrad16_group(items, tuning: Optional[dict] = None):
    """Several weight gradients dW_i += alpha_i dy_i^T x_i in ONE launch (dupl_gemm_f16x3_group: a whole 256 x 128 tile per block over
    the whole token axis, no split-K, no atomics -- the same bits in deterministic mode).  items: [(dy16 Split16 [Kp, n_out] scaled
    format 1 planes with zero rows up to Kp, x16 Split16 / view [rows, n_in] format 1 planes, out fp32 [n_out, n_in], alpha)]."""
    descs = []
    for dy16, x16, out, alpha in items:
        assert dy16.fmt == 1 and getattr(x16, "fmt", 0) == 1, "grouped weight gradients read format 1 planes k-major"
        Kp = dy16.rows
        assert Kp % 32 == 0 and Kp >= 96 and out.is_contiguous() and out.shape == (dy16.cols, x16.cols)
        if isinstance(alpha, _Alpha):
            alpha.check()
        d = _lib.Gemm16Desc()
        d.deterministic = int(deterministic())      # (whole tiles over all of K: the grouped form has no atomics in either mode)
        d.A_hi, d.A_lo, d.B_hi, d.B_lo = dy16.hi, dy16.lo, x16.hi, x16.lo
        d.C = out.data_ptr()
        d.M, d.N, d.K = dy16.cols, x16.cols, Kp
        d.lda, d.ldb, d.ldc, d.ldo = dy16.cols, x16.cols, out.stride(0), x16.cols
        d.flags = _lib.GEMM_ACCUM
        d.alpha_dev = int(alpha) if alpha is not None else None
        d.fmt, d.post_scale = 1, 2.0 ** -(getattr(dy16, "exp", 0) + getattr(x16, "exp", 0))
        d.a_layout, d.b_layout = 1, 1
        d.ka_valid, d.kb_valid = Kp, min(x16.rows, Kp)
        d.group = (tuning if tuning is not None else GEMM16_TUNING)["group"]
        descs.append(d)
    for i in range(0, len(descs), _lib.GEMM16_GROUP_MAX):
        chunk = descs[i:i + _lib.GEMM16_GROUP_MAX]
        arr = (_lib.Gemm16Desc * len(chunk))(*chunk)
        L().dupl_gemm_f16x3_group(arr, len(chunk), _stream())


def linear(x: Tensor, W: Tensor, bias: Optional[Tensor] = None, *, gelu: bool = False, relu: bool = False,
           res: Optional[Tensor] = None, out: Optional[Tensor] = None, store_pre: Optional[Tensor] = None) -> Tensor:
    """y[M,N] = act(x[M,K] @ W[N,K]^T + bias) + res     (nn.Linear / 1x1 conv forward).
    store_pre: optional [M,N] tensor that receives the pre-activation (x W^T + bias)."""
    M, K = x.shape
    N = W.shape[0]
    assert W.numel() == N * K
    y = out if out is not None else torch.empty((M, N), device=x.device, dtype=torch.float32)
    fl = (_lib.GEMM_GELU if gelu else 0) | (_lib.GEMM_RELU if relu else 0)
    if store_pre is not None:
        fl |= _lib.GEMM_STORE_PRE
    gemm_raw(x.data_ptr(), W.data_ptr(), y.data_ptr(), M, N, K, x.stride(0), K, y.stride(0), flags=fl,
             bias=_p(bias), res=_p(res), ldr=(res.stride(0) if res is not None else 0),
             aux=_p(store_pre), ldaux=(store_pre.stride(0) if store_pre is not None else 0))
    return y


def linear_dgrad(dy: Tensor, W: Tensor, *, dgelu_of: Optional[Tensor] = None, relumask_of: Optional[Tensor] = None,
                 out: Optional[Tensor] = None, accumulate: bool = False) -> Tensor:
    """dx[M,K] = dy[M,N] @ W[N,K]  (optionally * gelu'(pre) or * (post_relu > 0))."""
    M, N = dy.shape
    K = W.numel() // N
    dx = out if out is not None else torch.empty((M, K), device=dy.device, dtype=torch.float32)
    fl = _lib.GEMM_B_NCONTIG | (_lib.GEMM_ACCUM if accumulate else 0)
    aux = None
    if dgelu_of is not None:
        fl |= _lib.GEMM_MUL_DGELU
        aux = dgelu_of
    if relumask_of is not None:
        fl |= _lib.GEMM_MUL_RELUMASK
        aux = relumask_of
    gemm_raw(dy.data_ptr(), W.data_ptr(), dx.data_ptr(), M, K, N, dy.stride(0), K, dx.stride(0), flags=fl,
             aux=_p(aux), ldaux=(aux.stride(0) if aux is not None else 0))
    return dx


def linear_wgrad(dy: Tensor, x: Tensor, out: Tensor, accumulate: bool = False):
    """dW[N,K] (+)= dy[M,N]^T @ x[M,K]; `out` is any tensor with N*K elements (e.g. a conv weight view)."""
    M, N = dy.shape
    K = x.shape[1]
    assert out.numel() == N * K
    fl = _lib.GEMM_A_MCONTIG | _lib.GEMM_B_NCONTIG | (_lib.GEMM_ACCUM if accumulate else 0)
    gemm_raw(dy.data_ptr(), x.data_ptr(), out.data_ptr(), N, K, M, dy.stride(0), x.stride(0), K, flags=fl)


def colsum(x: Tensor, out: Tensor, accumulate: bool = False):
    M, N = x.shape
    L().dupl_colsum(x.data_ptr(), out.data_ptr(), M, N, x.stride(0), int(accumulate), int(deterministic()), _stream())


# ------------------------------------------------------------------------------------------ LayerNorm
def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, save: bool = False):
    rows, D = x.shape
    y = torch.empty_like(x)
    mean = rstd = None
    if save:
        mean = torch.empty(rows, device=x.device, dtype=torch.float32)
        rstd = torch.empty(rows, device=x.device, dtype=torch.float32)
    L().dupl_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), _p(mean), _p(rstd), rows, D,
                           eps, _stream())
    return y, mean, rstd


def layernorm_fwd16(x: Tensor, gamma: Tensor, beta: Tensor, eps: float, save: bool = False, want_f32: bool = False,
                    f32_rows: int = 0, exp: int = 0):
    """LayerNorm whose output goes out as f16x3 operand planes (and as fp32 too when want_f32, e.g. saved for backward).
    f32_rows > 0: the fp32 copy and mean / rstd are produced for the first f32_rows rows only (and have that many rows).
    Returns (y fp32 or None, y16 Split16, mean, rstd)."""
    rows, D = x.shape
    keep = f32_rows or rows
    y = torch.empty((keep, D), device=x.device, dtype=torch.float32) if want_f32 else None
    y16 = split16_empty(rows, D, x.device, exp)      # exp > 0: format 1 planes
    mean = rstd = None
    if save:
        mean = torch.empty(keep, device=x.device, dtype=torch.float32)
        rstd = torch.empty(keep, device=x.device, dtype=torch.float32)
    L().dupl_layernorm_fwd16(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _p(y), y16.hi, y16.lo, _p(mean), _p(rstd), rows, D,
                             eps, int(f32_rows), int(exp), _stream())
    return y, y16, mean, rstd


LNB_ROWS_PER_WAVE = 0      # dupl_layernorm_bwd's rows_per_wave (0 = the library default; tools/op_bench.py sweeps it)


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dgamma: Tensor, dbeta: Tensor,
                  dres: Optional[Tensor] = None, amax_for_next: bool = False, two_stage: Optional[bool] = None) -> Tensor:
    """Returns dx = dres + LN'(dy); accumulates into dgamma / dbeta -- fp32 atomics from the main kernel, or (two_stage; the
    default in deterministic mode) per-block partial sums + a fixed-order reduce kernel (dupl_layernorm_bwd's partials).
    amax_for_next: as in linear16.  A dy that came from zero_workspace() is handed back zero-filled by the same kernel."""
    rows, D = x.shape
    if two_stage is None:
        two_stage = deterministic()
    dx = torch.empty_like(x)
    word, tok = reserve_amax(x.device) if amax_for_next else (None, None)
    part, nb = None, 0
    if two_stage:
        nb = L().dupl_layernorm_bwd_blocks(rows, LNB_ROWS_PER_WAVE)
        part = torch.empty((nb, 2 * D), device=x.device, dtype=torch.float32)
    ws = getattr(dy, "_dupl_zero_ws", None)
    if ws is not None:
        assert ws.stream == _stream() and ws.dirty == dy.numel(), "a zero_workspace view is produced and consumed on its own stream"
    L().dupl_layernorm_bwd(dy.data_ptr(), x.data_ptr(), gamma.data_ptr(), mean.data_ptr(), rstd.data_ptr(), _p(dres),
                           dx.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(), rows, D, word, _p(part), nb, LNB_ROWS_PER_WAVE,
                           dy.data_ptr() if ws is not None else None, int(deterministic()), _stream())
    if ws is not None:
        ws.dirty = 0            # clean again, in the order of the stream it belongs to
    if tok is not None:
        dx._dupl_amax = tok
    return dx


_ZERO_WS = {}              # (device index, raw stream handle) -> _ZeroWs; at most _ZERO_WS_MAX entries (least recently used out)
_ZERO_WS_MAX = 8


class _ZeroWs:
    """One zero-filled fp32 arena per (device, stream): `buf` holds the largest rows * cols asked for so far, callers get a
    contiguous prefix view.  `clean_upto`: the prefix [0, clean_upto) IS zero in the order of the owning stream; a user dirties
    its view, the consumer that reads it last (layernorm_bwd: its kernel writes zeros behind its reads) cleans exactly that view."""
    __slots__ = ("buf", "dirty", "stream")

    def __init__(self, n, device, stream):
        self.buf = torch.zeros(n, device=device, dtype=torch.float32)
        self.dirty = 0          # elements [0, dirty) may be non-zero
        self.stream = stream


def zero_workspace(rows: int, cols: int, device) -> Tensor:
    """A [rows, cols] fp32 tensor that IS zero in the order of the current stream: a prefix view of the stream's arena (one
    buffer per stream, sized to the largest request; ADVICE r4: no buffer per shape, and the cache is bounded), handed out
    clean, dirtied by its user (a stream-K data gradient accumulates into it) and cleaned again by the consumer that reads it
    last -- layernorm_bwd recognises the view and has its kernel write zeros behind its reads.  If the previous user never
    reached that consumer (another path, an exception), the arena is still marked dirty and gets an explicit fill.
    Every producer and consumer of the view must run on the stream it was requested on (asserted by layernorm_bwd)."""
    stream = torch.cuda.current_stream(device).cuda_stream
    key = (device.index, stream)
    n = rows * cols
    ws = _ZERO_WS.pop(key, None)
    if ws is None or ws.buf.numel() < n:
        ws = _ZeroWs(n, device, stream)          # (the old arena, if any, is released when its last view dies)
    _ZERO_WS[key] = ws                            # most recently used last
    while len(_ZERO_WS) > _ZERO_WS_MAX:
        _ZERO_WS.pop(next(iter(_ZERO_WS)))
    if ws.dirty:
        fill_(ws.buf[:ws.dirty], 0.0)
    ws.dirty = n
    view = ws.buf[:n].view(rows, cols)
    view._dupl_zero_ws = ws
    return view


# ------------------------------------------------------------------------------------------ attention
def attention_fwd(qkv: Tensor, B: int, N: int, H: int, hd: int, scale: float, need_lse: bool = False, out: Tensor = None):
    """out: optional [B*N, H*hd] destination (a contiguous row slice of a larger token buffer)."""
    if out is None:
        out = torch.empty((B * N, H * hd), device=qkv.device, dtype=torch.float32)
    else:
        assert out.shape == (B * N, H * hd) and out.is_contiguous() and qkv.is_contiguous()
    lse = torch.empty((B, H, N), device=qkv.device, dtype=torch.float32) if need_lse else None
    L().dupl_attention_fwd(qkv.data_ptr(), out.data_ptr(), _p(lse), B, N, H, hd, scale, _stream())
    return out, lse


def attention_fwd16(qkv16, B: int, N: int, H: int, hd: int, scale: float, need_lse: bool = False,
                    out: Optional[Tensor] = None, out16=None, b_f32: int = 0):
    """Attention forward on the f16x3 split kernels (head dim 64).  qkv16: Split16 / Split16View [B*N, 3*H*hd] (planes of
    the qkv GEMM output); out: optional fp32 [B*N, H*hd] destination; out16: optional Split16 / Split16View [B*N, H*hd]
    receiving the planes of the output.  Returns lse (B, H, N) or None."""
    assert hd == 64 and qkv16.rows == B * N and qkv16.cols == 3 * H * hd and (out is not None or out16 is not None)
    dev = out.device if out is not None else (out16.planes.device if isinstance(out16, Split16) else out16.base.planes.device)
    bf = b_f32 or B          # fp32 out / lse for the first bf images only (the planes for all B)
    lse = torch.empty((bf, H, N), device=dev, dtype=torch.float32) if need_lse else None
    if out is not None:
        assert out.shape == (bf * N, H * hd) and out.is_contiguous()
    assert getattr(qkv16, "exp", 0) == 0, "the split attention reads format 0 planes"
    L().dupl_attention_fwd16(qkv16.hi, qkv16.lo, _p(out), out16.hi if out16 is not None else None,
                              out16.lo if out16 is not None else None, _p(lse), B, N, H, hd, float(scale), bf,
                              out16.exp if out16 is not None else 0, _stream())
    return lse


def attention_fwd16_segs(qkv16, segs, H: int, hd: int, scale: float, out16=None):
    """Split attention forward of SEVERAL batches that live in one token buffer, as ONE launch (dupl_attention_fwd16_segs).
    qkv16 / out16: planes of ALL rows; segs: [(row0, B, N, out fp32 [bf*N, H*hd] or None, need_lse, b_f32)] with bf = b_f32 or B.
    Returns the list of lse tensors (None where not asked for)."""
    assert hd == 64 and 1 <= len(segs) <= _lib.ATTN_SEGS_MAX and getattr(qkv16, "exp", 0) == 0
    dev = qkv16.planes.device if isinstance(qkv16, Split16) else qkv16.base.planes.device
    arr = (_lib.AttnSeg * len(segs))()
    lses = []
    for i, (row0, B, N, out, need_lse, b_f32) in enumerate(segs):
        bf = b_f32 or B
        assert out is not None or out16 is not None
        if out is not None:
            assert out.shape == (bf * N, H * hd) and out.is_contiguous()
        lse = torch.empty((bf, H, N), device=dev, dtype=torch.float32) if need_lse else None
        lses.append(lse)
        arr[i].row0, arr[i].B, arr[i].N, arr[i].B_f32 = int(row0), int(B), int(N), int(b_f32)
        arr[i].out, arr[i].lse = _p(out), _p(lse)
    L().dupl_attention_fwd16_segs(qkv16.hi, qkv16.lo, out16.hi if out16 is not None else None,
                                   out16.lo if out16 is not None else None, ctypes.cast(arr, ctypes.c_void_p), len(segs), H, hd,
                                   float(scale), out16.exp if out16 is not None else 0, _stream())
    return lses


def attention_bwd16(qkv16, out: Tensor, dout: Tensor, lse: Tensor, B: int, N: int, H: int, hd: int, scale: float,
                    amax_for_next: bool = False) -> Tensor:
    """Attention backward on the f16x3 split kernels (head dim 64, N <= 2048): qkv16 = the planes of the qkv GEMM output the
    forward saved; out / dout fp32 [B*N, H*hd]; returns dqkv fp32 [B*N, 3*H*hd].  amax_for_next: as in linear16."""
    assert hd == 64 and N <= 2048 and qkv16.rows == B * N
    dev = out.device
    dout = dout.contiguous()
    do16, _, alpha = split_prepare(dout, scaled=True, want_rm=True, want_T=False, target_exp=4)
    alpha.check()
    delta = torch.empty((B, H, N), device=dev, dtype=torch.float32)
    dqkv = torch.empty((B * N, 3 * H * hd), device=dev, dtype=torch.float32)
    word, tok = reserve_amax(dev) if amax_for_next else (None, None)       # after dout's split took its slot
    L().dupl_attention_bwd16(qkv16.hi, qkv16.lo, out.data_ptr(), dout.data_ptr(), do16.hi, do16.lo, int(alpha) - 4, lse.data_ptr(),
                             delta.data_ptr(), dqkv.data_ptr(), B, N, H, hd, float(scale), word, _stream())
    if tok is not None:
        dqkv._dupl_amax = tok
    return dqkv


def attention_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, B: int, N: int, H: int, hd: int, scale: float) -> Tensor:
    dqkv = torch.empty_like(qkv)
    delta = torch.empty((B, H, N), device=qkv.device, dtype=torch.float32)
    L().dupl_attention_bwd(qkv.data_ptr(), out.data_ptr(), dout.data_ptr(), lse.data_ptr(), delta.data_ptr(),
                           dqkv.data_ptr(), B, N, H, hd, scale, _stream())
    return dqkv


# ------------------------------------------------------------------------------------------ tokens
def patch_im2row(x: Tensor, P: int) -> Tensor:
    B, C, H, W = x.shape
    assert C == 3
    rows = torch.empty((B * (H // P) * (W // P), 3 * P * P), device=x.device, dtype=torch.float32)
    L().dupl_patch_im2row(x.data_ptr(), rows.data_ptr(), B, H, W, P, _stream())
    return rows


def pos_embed_resize(pos_embed: Tensor, g: int, h: int, w: int) -> Tensor:
    D = pos_embed.shape[-1]
    out = torch.empty((1 + h * w, D), device=pos_embed.device, dtype=torch.float32)
    L().dupl_pos_embed_resize(pos_embed.data_ptr(), out.data_ptr(), g, h, w, D, _stream())
    return out


def assemble_tokens(patch: Tensor, cls: Tensor, pos: Tensor, B: int, n: int, D: int) -> Tensor:
    tok = torch.empty((B * (n + 1), D), device=patch.device, dtype=torch.float32)
    L().dupl_assemble_tokens(patch.data_ptr(), cls.data_ptr(), pos.data_ptr(), tok.data_ptr(), B, n, D, _stream())
    return tok


def assemble_tokens_bwd(dtok: Tensor, dcls: Tensor, B: int, n: int, D: int) -> Tensor:
    dpatch = torch.empty((B * n, D), device=dtok.device, dtype=torch.float32)
    L().dupl_assemble_tokens_bwd(dtok.data_ptr(), dpatch.data_ptr(), dcls.data_ptr(), B, n, D, _stream())
    return dpatch


def gmp_fwd(tokens: Tensor, B: int, n: int, D: int):
    out = torch.empty((B, D), device=tokens.device, dtype=torch.float32)
    idx = torch.empty((B, D), device=tokens.device, dtype=torch.int32)
    L().dupl_gmp_fwd(tokens.data_ptr(), out.data_ptr(), idx.data_ptr(), B, n, D, _stream())
    return out, idx


def gmp_bwd(dout: Tensor, idx: Tensor, dtokens: Tensor, B: int, n: int, D: int):
    L().dupl_gmp_bwd(dout.data_ptr(), idx.data_ptr(), dtokens.data_ptr(), B, n, D, _stream())


def tokens_to_nchw(tokens: Tensor, B: int, n: int, D: int, h: int, w: int, skip_cls: bool = True) -> Tensor:
    out = torch.empty((B, D, h, w), device=tokens.device, dtype=torch.float32)
    L().dupl_tokens_to_nchw(tokens.data_ptr(), out.data_ptr(), B, n, D, int(skip_cls), _stream())
    return out


def nchw_to_tokens_add(dnchw: Tensor, dtokens: Tensor, B: int, n: int, D: int, skip_cls: bool = True):
    L().dupl_nchw_to_tokens_add(dnchw.data_ptr(), dtokens.data_ptr(), B, n, D, int(skip_cls), _stream())


# ------------------------------------------------------------------------------------------ CAM
def resize_bilinear(x: Tensor, Ho: int, Wo: int, flip_cat: bool = False, align_corners: bool = False) -> Tensor:
    B, C, Hi, Wi = x.shape
    x = _chk(x.contiguous())
    out = torch.empty(((2 * B) if flip_cat else B, C, Ho, Wo), device=x.device, dtype=torch.float32)
    L().dupl_resize_bilinear(x.data_ptr(), out.data_ptr(), B, C, Hi, Wi, Ho, Wo, int(flip_cat), int(align_corners), _stream())
    return out


def cam_fuse(lows: Sequence[Tensor], sizes: Sequence[tuple], B: int, C: int, H: int, W: int, row_off: int, ldc: int,
             impl: int = 0, band_blocks: int = 0):
    """lows[i]: [2B*(row_off+hs*ws), ldc] CAM logits of scale i.  Returns (cam (B,C,H,W) un-normalised, mm [B*C,2])."""
    n = len(lows)
    ptrs = (ctypes.c_void_p * n)(*[t.data_ptr() for t in lows])
    hs = (ctypes.c_int32 * n)(*[s[0] for s in sizes])
    ws = (ctypes.c_int32 * n)(*[s[1] for s in sizes])
    cam = torch.empty((B, C, H, W), device=lows[0].device, dtype=torch.float32)
    mm = torch.empty((B * C, 2), device=lows[0].device, dtype=torch.float32)
    L().dupl_cam_fuse(ctypes.cast(ptrs, ctypes.c_void_p), ctypes.cast(hs, ctypes.c_void_p), ctypes.cast(ws, ctypes.c_void_p),
                      n, row_off, ldc, cam.data_ptr(), mm.data_ptr(), B, C, H, W, int(impl), int(band_blocks), _stream())
    return cam, mm


def cam_normalise_(cam: Tensor, mm: Optional[Tensor] = None) -> Tensor:
    _chk(cam)
    planes = cam.shape[0] * cam.shape[1]
    HW = cam.shape[2] * cam.shape[3]
    have = mm is not None
    if mm is None:
        mm = torch.empty((planes, 2), device=cam.device, dtype=torch.float32)
    L().dupl_cam_minmax_normalise(cam.data_ptr(), mm.data_ptr(), planes, HW, int(have), _stream())
    return cam


def cam_to_label(cam: Tensor, cls_label: Tensor, img_box: Optional[Tensor], high_thre: Optional[Tensor], bkg_thre: float,
                 low_thre: float, ignore_mid: bool, ignore_index: int, want_valid: bool = False):
    b, C, h, w = cam.shape
    _chk(cam), _chk(cls_label)
    label = torch.empty((b, h, w), device=cam.device, dtype=torch.int64)
    valid = torch.empty_like(cam) if want_valid else None
    L().dupl_cam_to_label(cam.data_ptr(), cls_label.data_ptr(), _p(img_box), _p(high_thre), float(bkg_thre),
                          float(low_thre if low_thre is not None else 0.0), int(ignore_mid),
                          int(ignore_index if ignore_index is not None else 0), label.data_ptr(), _p(valid), b, C, h, w,
                          _stream())
    return valid, label


def denormalize_img(x: Tensor, mean=None, std=None) -> Tensor:
    B, C, H, W = x.shape
    assert C == 3
    x = _chk(x.contiguous())
    out = torch.empty_like(x)
    ms = None
    if mean is not None or std is not None:
        m = list(mean) if mean is not None else [123.675, 116.28, 103.53]
        sd = list(std) if std is not None else [58.395, 57.12, 57.375]
        assert len(m) == 3 and len(sd) == 3
        ms = (ctypes.c_float * 6)(*m, *sd)
    L().dupl_denormalize_img(x.data_ptr(), out.data_ptr(), B, H * W, ms, _stream())
    return out


# ------------------------------------------------------------------------------------------ PAR
def par_pos_term(dilations: Sequence[int], w1: float = 0.3, w2: float = 0.01) -> np.ndarray:
    """w2 * softmax_k(-(pos_k/(std(pos)+1e-8)/w1)^2): the input-independent part of PAR.forward
    (PAR.py:51-62,78,83-85), evaluated once on the host in float32 like the reference does."""
    ker = np.ones(8, dtype=np.float32)
    ker[[0, 2, 5, 7]] = np.float32(np.sqrt(2))
    pos = np.concatenate([ker * np.float32(d) for d in dilations]).astype(np.float32)
    std = np.float32(np.std(pos.astype(np.float64), ddof=1))
    a = -((pos / (std + np.float32(1e-8)) / np.float32(w1)) ** 2)
    e = np.exp(a - a.max())
    return (np.float32(w2) * (e / e.sum())).astype(np.float32)


def par_affinity(imgs: Tensor, dilations: Sequence[int], pos_term: Tensor) -> Tensor:
    B, C, h, w = imgs.shape
    assert C == 3
    nd = len(dilations)
    aff = torch.empty((B, 8 * nd, h, w), device=imgs.device, dtype=torch.float32)
    dil = (ctypes.c_int32 * nd)(*dilations)
    L().dupl_par_affinity(imgs.data_ptr(), aff.data_ptr(), ctypes.cast(dil, ctypes.c_void_p), nd, pos_term.data_ptr(), B, h, w,
                          _stream())
    return aff


def par_propagate(aff: Tensor, masks: Tensor, job_img: Tensor, job_K: Tensor, dilations: Sequence[int], num_iter: int) -> Tensor:
    """masks [njobs, Kmax, h, w]; returns the propagated masks after num_iter ping-pong iterations."""
    njobs, Kmax, h, w = masks.shape
    nd = len(dilations)
    dil = (ctypes.c_int32 * nd)(*dilations)
    a, b = masks, torch.empty_like(masks)
    for _ in range(num_iter):
        L().dupl_par_propagate(aff.data_ptr(), a.data_ptr(), b.data_ptr(), job_img.data_ptr(), job_K.data_ptr(),
                               ctypes.cast(dil, ctypes.c_void_p), nd, njobs, Kmax, h, w, _stream())
        a, b = b, a
    return a


def refine_pre(cams: Tensor, thr_map: Optional[Tensor], thr: Optional[Tensor], job_img: Tensor, job_K: Tensor, keys: Tensor,
               down_scale: int = 2) -> Tensor:
    b, C, H, W = cams.shape
    _chk(cams)
    njobs, Kmax = keys.shape
    h, w = H // down_scale, W // down_scale
    masks = zeros((njobs, Kmax, h, w), cams.device)
    L().dupl_refine_pre(cams.data_ptr(), _p(thr_map), _p(thr), job_img.data_ptr(), job_K.data_ptr(), keys.data_ptr(), njobs, Kmax,
                        masks.data_ptr(), C, H, W, h, w, _stream())
    return masks


def refine_post(masks: Tensor, job_img: Tensor, job_K: Tensor, keys: Tensor, box: Tensor, ignore_index: float,
                out_size=None) -> Tensor:
    njobs, Kmax, h, w = masks.shape
    H, W = out_size if out_size is not None else (2 * h, 2 * w)
    label = torch.empty((njobs, H, W), device=masks.device, dtype=torch.float32)
    L().dupl_refine_post(masks.data_ptr(), job_img.data_ptr(), job_K.data_ptr(), keys.data_ptr(), njobs, Kmax, box.data_ptr(),
                         float(ignore_index), label.data_ptr(), h, w, H, W, _stream())
    return label


def refine_merge(lab_h: Tensor, lab_l: Tensor, ignore_index: float) -> Tensor:
    out = torch.empty_like(lab_h)
    L().dupl_refine_merge(lab_h.data_ptr(), lab_l.data_ptr(), out.data_ptr(), float(ignore_index), lab_h.numel(), _stream())
    return out


# ------------------------------------------------------------------------------------------ misc
def to_device_async(host, dtype, device) -> Tensor:
    """Small host table -> device without a stream synchronisation: staged through pinned memory and copied with
    non_blocking=True (a plain torch.tensor(..., device=dev) copies from pageable memory and blocks the host until
    every kernel queued before it has finished)."""
    t = torch.as_tensor(np.asarray(host), dtype=dtype)
    if device.type != "cuda":
        return t.to(device)
    return t.pin_memory().to(device, non_blocking=True)


def fill_(t: Tensor, v: float):
    L().dupl_fill(t.data_ptr(), float(v), t.numel(), _stream())
    return t


def axpy_(y: Tensor, x: Tensor, a: float = 1.0):
    assert y.numel() == x.numel()
    L().dupl_axpy(y.data_ptr(), x.data_ptr(), float(a), y.numel(), _stream())
    return y


def scale_(y: Tensor, a: float):
    L().dupl_scale(y.data_ptr(), float(a), y.numel(), _stream())
    return y


LOSS_SUMS_FLOATS = 136     # include/dupl_hip.h DUPL_LOSS_SUMS_FLOATS: the `sums` buffers of dupl_ptc_reduce / dupl_seg_loss_fwd


def zeros(shape, device) -> Tensor:
    t = torch.empty(shape, device=device, dtype=torch.float32)
    return fill_(t, 0.0)


# ------------------------------------------------------------------------------------------ validation / evaluation
def upsample_argmax(logits: Tensor, H: int, W: int) -> Tensor:
    """argmax_c of the bilinear (align_corners=False) up-sampling of logits (B,C,h,w) to (H,W) -> (B,H,W) int64."""
    B, C, h, w = logits.shape
    logits = _chk(logits.contiguous())
    out = torch.empty((B, H, W), device=logits.device, dtype=torch.int64)
    L().dupl_upsample_argmax(logits.data_ptr(), out.data_ptr(), B, C, h, w, int(H), int(W), _stream())
    return out


def msc_seg_accum_(acc: Tensor, segs: Tensor, mode: int) -> Tensor:
    """v = up(segs[0]) + flip(up(segs[1])), segs (2,C,h,w); acc (1,C,H,W) = v (mode 0) | max(acc, v) (1) | acc + v (2):
    one scale of eval_seg_voc.py:58-72 (max) / eval_seg_coco_ddp.py:80-119 (sum)."""
    two, C, h, w = segs.shape
    assert two == 2 and acc.shape[1] == C and acc.is_contiguous()
    segs = _chk(segs.contiguous())
    H, W = acc.shape[-2:]
    L().dupl_msc_seg_accum(segs.data_ptr(), acc.data_ptr(), C, h, w, H, W, int(mode), _stream())
    return acc


def argmax_channels(x: Tensor) -> Tensor:
    B, C = x.shape[:2]
    HW = x[0, 0].numel()
    x = _chk(x.contiguous())
    out = torch.empty((B,) + tuple(x.shape[2:]), device=x.device, dtype=torch.int64)
    L().dupl_argmax_channels(x.data_ptr(), out.data_ptr(), B, C, HW, _stream())
    return out


def confusion_accum(gt: Tensor, pred: Tensor, hist: Tensor) -> Tensor:
    """hist (nc,nc) int64 += confusion counts of the pixels with 0 <= gt < nc (evaluate._fast_hist)."""
    assert gt.dtype == torch.int64 and pred.dtype == torch.int64 and hist.dtype == torch.int64
    assert gt.numel() == pred.numel() and gt.is_cuda and pred.is_cuda and hist.is_cuda and hist.is_contiguous()
    gt, pred = gt.contiguous(), pred.contiguous()
    L().dupl_confusion_accum(gt.data_ptr(), pred.data_ptr(), gt.numel(), hist.shape[0], hist.data_ptr(), _stream())
    return hist


def multilabel_f1_accum(logits: Tensor, label: Tensor, total: Tensor) -> Tensor:
    """total[0] += sum over rows of f1((logits > 0), label)  (evaluate.multilabel_score per image)."""
    B, C = logits.shape
    logits, label = _chk(logits.contiguous()), _chk(label.contiguous().float())
    L().dupl_multilabel_f1_accum(logits.data_ptr(), label.data_ptr(), B, C, total.data_ptr(), _stream())
    return total
