"""Host-side pieces of bench.py that run without a GPU: the CPU baselines must stay bounded (the driver's default
`python bench.py` has minutes, not hours) and the JSON contract's static fields must be there."""
import importlib.util
import sys
import time
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parent.parent


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", ROOT / "bench.py")
    m = importlib.util.module_from_spec(spec)
    sys.modules["bench_mod"] = m
    spec.loader.exec_module(m)
    return m


def test_torch_cpu_baseline_is_bounded_and_restores_the_thread_count():
    """BASELINE configs[0] size through oracle/torch_ref.py on at most 8 threads (one intra-op thread per core of a
    256-core host took 19 minutes for 1.6 s of work) — a few seconds here, and torch's thread count is put back."""
    b = _bench()
    before = torch.get_num_threads()
    t0 = time.time()
    r = b.cpu_baseline_torch()
    dt = time.time() - t0
    assert torch.get_num_threads() == before
    assert r["unit"] == "Msplats/s" and r["kind"] == "port" and 1 <= r["cores"] <= 8 and r["value"] > 0
    assert "10000 Gaussians" in r["sample"] and dt < 120


def test_bench_requires_a_gpu_and_has_no_cpu_path():
    src = (ROOT / "bench.py").read_text()
    assert "bench.py needs a GPU" in src
    for key in ('"roofline"', '"cpu_baseline"', "higher_is_better", "vs_baseline", '"scaling"'):
        assert key in src, key
