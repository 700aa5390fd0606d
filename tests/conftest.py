import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def repo_root():
    return ROOT


@pytest.fixture(params=["f32", "split", "f16x2"])
def gemm_mode(request):
    """Run a GPU parity test in every projection arithmetic: exact fp32 MFMA, the 3 x bf16-split MFMA and the
    2 x fp16-split MFMA with block exponents."""
    from gotennet_amd import engine
    old = engine.GEMM_MODE
    engine.GEMM_MODE = request.param
    yield request.param
    engine.GEMM_MODE = old
