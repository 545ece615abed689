import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def native():
    """The built native pieces (built on demand; hipcc cross-compiles on CPU)."""
    from madrona_renderer_amd import build
    build.build_all()
    import madrona_renderer_amd
    return madrona_renderer_amd


@pytest.fixture(scope="session")
def oracle_mod():
    from oracle import oracle
    oracle.build()
    return oracle


def has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False
