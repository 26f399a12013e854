import os
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU visible")
    from splatter360_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    return torch.device("cuda:0")


@pytest.fixture
def parity_lists():
    """Upstream-compatible tile lists (3-sigma rectangles) for the tests that compare tiles_touched / sorted lists / n_contrib
    with the oracle; the product default is the lean lists (rasterizer.LEAN_LISTS), whose images and gradients are bit-identical
    (tests/test_gpu_lean.py), with long unsaturated lists composited segment-parallel (rasterizer.SPLIT_LONG_LISTS: same integers,
    float association differs inside split quadrants — tests/test_gpu_saturating_parity.py puts that mode against the oracle)."""
    from splatter360_amd import rasterizer
    old = (rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS)
    rasterizer.LEAN_LISTS = False
    rasterizer.SPLIT_LONG_LISTS = False     # ... and every list one sequential chain: the oracle's rounding order
    yield
    rasterizer.LEAN_LISTS, rasterizer.SPLIT_LONG_LISTS = old
