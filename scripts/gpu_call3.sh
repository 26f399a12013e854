#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/tests_call3.log
cat gpurun_out/tests_call3.log
for b in 1 0; do
S360_BWD_B4=$b python bench.py --cpu-baseline 0 2>/dev/null | tail -1 > gpurun_out/bench_b4_$b.json
python - <<P
import json
d=json.load(open("gpurun_out/bench_b4_$b.json"))
print("B4=$b", round(d["value"],1), round(d["ms_per_step"],3), {k:round(v["avg_us"]) for k,v in d["kernels"].items()})
P
done
