import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (os.path.join(ROOT, "lammps-spherharm_amd"), ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (checker). Built on demand from oracle/shpair_oracle.c."""
    from oracle import oracle as O
    O.build()
    return O


@pytest.fixture(scope="session")
def gpu_available():
    return _has_gpu()
