import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
for p in (ROOT, ROOT / "enlsip.jl_amd" / "python"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")
    # the oracle's LAPACK calls: at most 16 BLAS threads (the GPU box grants a 16-core share while showing 256 cores;
    # oversubscribed BLAS workers keep spinning after a call and starve the HIP runtime's submission thread)
    try:
        from threadpoolctl import threadpool_limits
        config._blas_limit = threadpool_limits(limits=16, user_api="blas")
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return ROOT / "tests" / "golden"


def pytest_collection_modifyitems(config, items):
    """GPU runs: let PyTorch initialise its HIP runtime BEFORE the library does.  torch ships its own ROCm libraries and the
    library links the system's; in one process the runtime that comes up second after the other has claimed the device reports
    "No HIP GPUs are available" when that second one is torch's (seen with `-k tsqr`, where the first GPU call used to be the
    library's).  bench.py imports torch first for the same reason."""
    if not any(item.get_closest_marker("gpu") for item in items):
        return
    try:
        import torch
        if torch.cuda.is_available():
            torch.zeros(1, device="cuda:0")
    except Exception:
        pass
