"""dupl_amd -- MI355X-native (gfx950) engine for DuPL's per-step hot path.

Host code is Python on PyTorch-ROCm (tensors, streams, torch.distributed); every dense op on the
step path is a hand-written HIP kernel in libdupl_hip.so reached through the C ABI declared in
include/dupl_hip.h.  There is no CPU fallback in this package.
"""
__version__ = "0.1.0"
