import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.dont_write_bytecode = False


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # `-m gpu` on a box without a GPU must fail loudly, not skip silently: leave items untouched.
    # Plain `pytest` (no -m) on a CPU box skips the gpu tests.
    if config.getoption("-m"):
        return
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible (run with -m gpu on the GPU box)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
