import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: minutes of host-side oracle work per case; skipped unless DUPL_RUN_SLOW=1 "
                                       "(tools/gate.sh runs them)")


def pytest_collection_modifyitems(config, items):
    if os.environ.get("DUPL_RUN_SLOW", "0") == "1":
        return
    skip = pytest.mark.skip(reason="slow case: set DUPL_RUN_SLOW=1 (tools/gate.sh does)")
    for it in items:
        if "slow" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="session", autouse=True)
def _report_gemm_mode():
    """DUPL_GEMM=f16x3 runs the WHOLE suite with the encoder's forward Linears on the split-f16 GEMM (the gate for
    making it the default: every parity bar must hold in either mode)."""
    from dupl_amd import engine
    mode = engine.GEMM_MODE
    print(f"\n[dupl_amd] forward GEMM mode for this session: {mode}")
    yield
