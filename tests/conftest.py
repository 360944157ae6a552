import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def sift():
    """One library context on GPU 0 for the whole session (no fallback: a missing
    library or device is a failure, not a skip)."""
    import sift_amd

    ctx = sift_amd.SiftHip(0)
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def oracle():
    from tests import oracle_lib

    return oracle_lib.Oracle()
