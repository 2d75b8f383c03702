import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


def pytest_sessionstart(session):
    """Test setup, not a fallback: make sure the two native libraries exist (hipcc cross-compiles gfx950 without a GPU;
    gcc builds the oracle).  The product itself never builds or falls back implicitly (i-vit_amd/_lib.py)."""
    import importlib
    _lib = importlib.import_module("i-vit_amd._lib")
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    from oracle import oracle as orc
    orc.build()
