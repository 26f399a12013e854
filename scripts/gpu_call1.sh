#!/bin/bash
# first GPU call of round 2: full GPU test suite + profiles at HEAD
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/tests_call1.log
cat gpurun_out/tests_call1.log
ROUND=r02a timeout 900 bash scripts/collect_profiles.sh > gpurun_out/collect_r02a.log 2>&1
tail -3 gpurun_out/r02a/bench_fwdbwd.json | cut -c1-1500
