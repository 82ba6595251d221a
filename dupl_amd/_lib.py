"""ctypes binding of libdupl_hip.so -- the reference-side stub INTEGRATION.md describes.

The prototypes are read from include/dupl_hip.h (single source of truth) and turned into ctypes
signatures; every exported function is wrapped so that a non-zero status raises.  There is NO
fallback: if the shared library is missing or a symbol is absent, importing the product path fails.
"""
from __future__ import annotations

import ctypes
import os
import re
from typing import Dict, List, Tuple

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "include", "dupl_hip.h")
LIB_PATH = os.path.join(HERE, "libdupl_hip.so")


class GemmDesc(ctypes.Structure):
    """Mirror of `dupl_gemm_desc` (include/dupl_hip.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("tile_rows", ctypes.c_int32), ("tile_cols", ctypes.c_int32), ("group", ctypes.c_int32),
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p), ("aux", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("lda", ctypes.c_int32), ("ldb", ctypes.c_int32), ("ldc", ctypes.c_int32),
        ("ldr", ctypes.c_int32), ("ldaux", ctypes.c_int32),
        ("batch", ctypes.c_int32), ("zdiv", ctypes.c_int32),
        ("sA0", ctypes.c_int64), ("sA1", ctypes.c_int64), ("sB0", ctypes.c_int64), ("sB1", ctypes.c_int64),
        ("sC0", ctypes.c_int64), ("sC1", ctypes.c_int64), ("sR0", ctypes.c_int64), ("sR1", ctypes.c_int64),
        ("sX0", ctypes.c_int64), ("sX1", ctypes.c_int64), ("sBias0", ctypes.c_int64), ("sBias1", ctypes.c_int64),
        ("alpha", ctypes.c_float), ("flags", ctypes.c_int32), ("deterministic", ctypes.c_int32), ("reserved0", ctypes.c_int32),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(GemmDesc)


class Gemm16Desc(ctypes.Structure):
    """Mirror of `dupl_gemm16_desc` (include/dupl_hip.h): operands as fp16 hi / lo planes."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("deterministic", ctypes.c_int32),
        ("A_hi", ctypes.c_void_p), ("A_lo", ctypes.c_void_p), ("B_hi", ctypes.c_void_p), ("B_lo", ctypes.c_void_p),
        ("C", ctypes.c_void_p), ("C_hi", ctypes.c_void_p), ("C_lo", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("res", ctypes.c_void_p), ("aux", ctypes.c_void_p),
        ("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
        ("lda", ctypes.c_int32), ("ldb", ctypes.c_int32), ("ldc", ctypes.c_int32), ("ldo", ctypes.c_int32),
        ("ldr", ctypes.c_int32), ("ldaux", ctypes.c_int32),
        ("flags", ctypes.c_int32), ("c_rows", ctypes.c_int32),
        ("alpha_dev", ctypes.c_void_p), ("fmt", ctypes.c_int32), ("out_exp", ctypes.c_int32), ("post_scale", ctypes.c_float),
        ("amax_out", ctypes.c_void_p),
        ("a_layout", ctypes.c_int32), ("b_layout", ctypes.c_int32), ("ka_valid", ctypes.c_int32), ("kb_valid", ctypes.c_int32),
        ("tile", ctypes.c_int32), ("concurrency", ctypes.c_int32), ("persist_blocks", ctypes.c_int32), ("group", ctypes.c_int32),
        ("sk_slices", ctypes.c_int32), ("reserved1", ctypes.c_int32),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(Gemm16Desc)


class SplitDesc(ctypes.Structure):
    """Mirror of `dupl_split_desc` (include/dupl_hip.h)."""
    _fields_ = [
        ("struct_size", ctypes.c_uint32), ("ld", ctypes.c_int32), ("R", ctypes.c_int32), ("C", ctypes.c_int32),
        ("x", ctypes.c_void_p), ("slot", ctypes.c_void_p), ("next_bits", ctypes.c_void_p),
        ("hi", ctypes.c_void_p), ("lo", ctypes.c_void_p), ("hiT", ctypes.c_void_p), ("loT", ctypes.c_void_p),
        ("Rp", ctypes.c_int32), ("target_exp", ctypes.c_int32), ("colsum_accum", ctypes.c_void_p),
        ("amax_mode", ctypes.c_int32), ("fmt", ctypes.c_int32), ("rows_zero_to", ctypes.c_int32), ("deterministic", ctypes.c_int32),
    ]

    def __init__(self, *a, **kw):
        super().__init__(*a, **kw)
        self.struct_size = ctypes.sizeof(SplitDesc)


GEMM_A_MCONTIG, GEMM_B_NCONTIG, GEMM_GELU, GEMM_ACCUM = 1, 2, 4, 8
GEMM_MUL_DGELU, GEMM_RELU, GEMM_MUL_RELUMASK, GEMM_ABS, GEMM_STORE_PRE = 16, 32, 64, 128, 256

class SplitItem(ctypes.Structure):
    """dupl_split_item (include/dupl_hip.h)."""
    _fields_ = [("x", ctypes.c_void_p), ("hi", ctypes.c_void_p), ("lo", ctypes.c_void_p), ("hiT", ctypes.c_void_p),
                ("loT", ctypes.c_void_p), ("ld", ctypes.c_int32), ("R", ctypes.c_int32), ("C", ctypes.c_int32),
                ("Rp", ctypes.c_int32)]


class AttnSeg(ctypes.Structure):
    """dupl_attn_seg (include/dupl_hip.h): one batch of a segmented attention-forward launch."""
    _fields_ = [("row0", ctypes.c_int64), ("B", ctypes.c_int32), ("N", ctypes.c_int32), ("B_f32", ctypes.c_int32),
                ("reserved0", ctypes.c_int32), ("out", ctypes.c_void_p), ("lse", ctypes.c_void_p)]


ATTN_SEGS_MAX = 4             # DUPL_ATTN_SEGS_MAX
SPLIT_MULTI_MAX = 16
GEMM16_GROUP_MAX = 8          # DUPL_GEMM16_GROUP_MAX

_PROTO = re.compile(r"^\s*int\s+(dupl_\w+)\s*\(([^;{]*)\)\s*;", re.M | re.S)


def _ctype(decl: str):
    d = decl.strip()
    if d == "void" or not d:
        return None
    if "dupl_gemm16_desc" in d:
        return ctypes.POINTER(Gemm16Desc)
    if "dupl_split_desc" in d:
        return ctypes.POINTER(SplitDesc)
    if "dupl_gemm_desc" in d:
        return ctypes.POINTER(GemmDesc)
    if "*" in d or d.startswith("dupl_stream_t"):
        return ctypes.c_void_p
    base = d.replace("const", "").split()
    ty = base[0]
    return {"int32_t": ctypes.c_int32, "int64_t": ctypes.c_int64, "float": ctypes.c_float, "uint8_t": ctypes.c_uint8,
            "int": ctypes.c_int, "double": ctypes.c_double, "uint32_t": ctypes.c_uint32, "uint64_t": ctypes.c_uint64}[ty]


def parse_header(path: str = HEADER) -> Dict[str, List]:
    """{function name: [ctypes argtypes]} for every `int dupl_*(...)` prototype in the header."""
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    src = re.sub(r"//[^\n]*", "", src)
    out: Dict[str, List] = {}
    for m in _PROTO.finditer(src):
        name, args = m.group(1), m.group(2)
        tys = [_ctype(a) for a in args.split(",")]
        out[name] = [t for t in tys if t is not None]
    return out


class _Lib:
    def __init__(self):
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  dupl_amd has no CPU / eager fallback by design.")
        # The process must hold ONE HIP runtime.  torch ships its own libamdhip64 (SONAME libamdhip64.so.7, found by
        # libtorch_hip under the file name libamdhip64.so): when torch is loaded first our NEEDED libamdhip64.so.7
        # binds to that copy; loaded the other way round the dynamic linker maps /opt/rocm's runtime for us and torch's
        # for torch, and torch stream handles passed across the C ABI are then foreign (launches fail with -2).
        import torch  # noqa: F401  (device memory + streams come from torch: load its runtime first)
        self.cdll = ctypes.CDLL(LIB_PATH)
        runtimes = {ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln}
        if len(runtimes) > 1:
            raise ImportError(f"two HIP runtimes mapped in one process ({sorted(runtimes)}): stream handles and "
                              "device pointers would not be interchangeable")
        self.protos = parse_header()
        for name, argtypes in self.protos.items():
            fn = getattr(self.cdll, name)  # AttributeError if the .so does not export a declared symbol
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
            # getters return a value, everything else a status (0 = ok) that is checked on every call
            setattr(self, name, fn if name in ("dupl_abi_version", "dupl_layernorm_bwd_blocks") else self._checked(name, fn))
        # Build identity (ABI 4): the digest of the kernel sources is baked into the library at build time.  The GPU box runs a
        # PREBUILT, git-ignored .so that travels with the tree: if the sources next to it are not the ones it was built from, refuse
        # it here rather than measure (or test) one set of kernels under the name of another.
        buf = ctypes.create_string_buffer(80)
        if self.cdll.dupl_build_digest(buf, 80) != 0:
            raise ImportError(f"{LIB_PATH}: dupl_build_digest failed")
        self.build_digest = buf.value.decode()
        from .build import source_digest, _sources
        if _sources() and os.environ.get("DUPL_SKIP_DIGEST_CHECK", "0") != "1":
            have = source_digest()
            if have != self.build_digest:
                raise ImportError(
                    f"{LIB_PATH} was built from kernel sources with digest {self.build_digest[:16]}, the sources in this tree have "
                    f"{have[:16]}: rebuild it (`python -c 'import __graft_entry__ as g; g.build()'`)")

    @staticmethod
    def _checked(name, fn):
        def call(*a):
            rc = fn(*a)
            if rc != 0:
                raise RuntimeError(f"{name} failed with status {rc} (-1 bad argument, -2 launch failure)")
            return rc
        call.__name__ = name
        call.raw = fn
        return call


_LIB = None


def lib() -> _Lib:
    global _LIB
    if _LIB is None:
        _LIB = _Lib()
    return _LIB
