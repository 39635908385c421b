import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    import __graft_entry__ as ge
    try:    # tests that share device buffers with torch: let torch initialise the GPU before liborbx does (seen on the
        import torch   # GPU box: torch's lazy CUDA init failed with "No HIP GPUs are available" after liborbx had run)
        torch.cuda.is_available()
    except Exception:
        pass
    return ge.load_pkg()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle_py
    oracle_py.build()
    return oracle_py
