"""dupl_amd -- MI355X-native (gfx950) engine for DuPL's per-step hot path.

Host code is Python on PyTorch-ROCm (tensors, streams, torch.distributed); every dense op on the
step path is a hand-written HIP kernel in libdupl_hip.so reached through the C ABI declared in
include/dupl_hip.h.  There is no CPU fallback in this package.
"""
__version__ = "0.1.0"


def install_reference_aliases(override: bool = False):
    """Make the reference's own import lines resolve to this package:

        import dupl_amd; dupl_amd.install_reference_aliases()
        from model.model_dupl import siamese_network          # train_final_voc.py:17-30, unchanged
        from model.losses import get_masked_ptc_loss, get_seg_loss
        from model.PAR import PAR
        from utils import cam_helper, train_helper, imutils, evaluate

    by registering `model`, `utils`, `datasets`, `tools` (and their sub-modules) in sys.modules as aliases of
    dupl_amd.model, dupl_amd.utils, ...  (Putting dupl_amd/ itself on sys.path does NOT work: the modules use
    package-relative imports.)  Existing top-level modules of those names are kept unless override=True."""
    import importlib
    import pkgutil
    import sys

    def alias(src: str, dst: str):
        mod = importlib.import_module(src)
        if override or dst not in sys.modules:
            sys.modules[dst] = mod
        if hasattr(mod, "__path__"):
            for info in pkgutil.iter_modules(mod.__path__):
                alias(f"{src}.{info.name}", f"{dst}.{info.name}")

    for pkg in ("model", "utils", "datasets", "tools"):
        alias(f"dupl_amd.{pkg}", pkg)


def set_deterministic(on: bool = True):
    """Bit-reproducible steps (the reference's `torch.backends.cudnn.deterministic = True`, train_final_voc.py:95-102):
    every accumulation that otherwise uses fp32 atomics -- split-K / stream-K gradients, LayerNorm dgamma / dbeta, bias column
    sums, the seg-loss backward scatter -- runs in a fixed order (the loss scalars are order-independent in either mode: 64-bit
    fixed-point sums, csrc/loss.hip).  Costs throughput (the weight-gradient GEMMs lose their k-split); off by default.  Also
    switched on by the environment variable DUPL_DETERMINISTIC=1.  The switch lives on the caller's side (ops.set_deterministic):
    since ABI 3 the library holds no mode, every call is handed the value."""
    from . import ops
    ops.set_deterministic(on)
