#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -40 > gpurun_out/tests_call2.log
cat gpurun_out/tests_call2.log
python bench.py --cpu-baseline 0 2>/dev/null | tail -1 > gpurun_out/bench_em.json
python - <<'P'
import json
for n in ("em",):
    d=json.load(open(f"gpurun_out/bench_{n}.json"))
    print(n, round(d["value"],1), round(d["ms_per_step"],3), {k:round(v["avg_us"]) for k,v in d["kernels"].items()})
P
