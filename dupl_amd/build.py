"""Build libdupl_hip.so (gfx950 only) in-tree with hipcc.  No JIT cache, no cmake: one hipcc per
translation unit (parallel), then one link.  The .so is git-ignored but travels with the tree."""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libdupl_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


def _sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps():
    hdr = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hdr.append(os.path.join(os.path.dirname(HERE), "include", "dupl_hip.h"))
    return hdr


def source_digest() -> str:
    """sha256 over the kernel sources (csrc/*.hip, csrc/*.h, include/dupl_hip.h): the build identity that profile
    summaries are tagged with (tools/profile_round.sh) and that bench.py checks before quoting a PMC number."""
    import hashlib
    h = hashlib.sha256()
    for path in sorted(_sources() + _deps()):
        h.update(os.path.basename(path).encode())
        with open(path, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _stale(target: str, srcs) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in srcs)


def build_library(force: bool = False, verbose: bool = True) -> str:
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    deps = _deps()
    jobs = []
    objs = []
    for src in _sources():
        obj = os.path.join(OBJ, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or _stale(obj, [src] + deps):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stderr}")
        return src

    if jobs:
        if verbose:
            print(f"[dupl_amd.build] compiling {len(jobs)} HIP translation unit(s) for {ARCH}", file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    if force or jobs or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stderr}")
        if verbose:
            print(f"[dupl_amd.build] linked {LIB}", file=sys.stderr)
    return LIB


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
