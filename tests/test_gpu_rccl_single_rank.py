"""RCCL on the one GPU of the box: a world-size-1 `nccl` process group drives the chunked in-backward exchange and
the factored start / finish exchange through their COLLECTIVE branches (distributed.ExchangeConfig(force_collectives=True)) —
gloo, which every other multi-rank test uses, runs its collectives synchronously on the host and can say nothing about the ordering
between ProcessGroupNCCL's streams and the ctypes-launched kernels.  Own process: the process group must not leak into the suite.
Reference behaviour replaced: Lightning DDP's gradient all-reduce (/root/reference/src/main.py:117-130)."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def test_chunked_and_factored_exchange_through_rccl_with_one_rank(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, str(ROOT / "scripts" / "rccl_single_rank.py")], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "rccl single rank ok: backend nccl world 1" in r.stdout, r.stdout[-2000:]


def test_bench_single_rank_rccl_line(gpu):
    """`bench.py --single-rank-rccl 1`: the bench's own training step with the chunked exchange issued through a one-rank RCCL
    communicator — small cloud, the JSON line says which backend ran the collectives and what the exchange cost."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--cpu-baseline", "0",
                        "--pano-h", "64", "--face", "64", "--workloads", "0", "--single-rank-rccl", "1"], env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["backend"] == "nccl" and d["rccl_ranks"] == 1 and d["exchange"]["mode"] == "chunked" and d["value"] > 0
