#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25
bash scripts/calib/run_calib.sh 2>&1 | tail -14
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 2>gpurun_out/bench_err.txt | tail -1 > gpurun_out/bench_now.json
python - <<'P'
import json
r=json.load(open("gpurun_out/bench_now.json")); k=r['kernels']
print(round(r['value'],1), round(r['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in k.items()})
print(r['roofline']); print(r.get('forward_only')); print(r.get('backend'), r.get('rccl_ranks'), r.get('rank_devices'))
P
tail -3 gpurun_out/bench_err.txt
S360_DIST_BACKEND=gloo S360_FORCE_DEVICE=0 python bench.py --gpus 2 --steps 5 --warmup 2 --cpu-baseline 0 2>gpurun_out/bench2_err.txt | tail -1 > gpurun_out/bench_2rank.json
python - <<'P'
import json
r=json.load(open("gpurun_out/bench_2rank.json"))
print("2 ranks / 1 GPU gloo:", round(r['value'],1), round(r['ms_per_step'],3), r['config']['parallelism'], r.get('exchange'), r.get('backend'), r.get('rccl_ranks'), r.get('rank_devices'))
P
tail -3 gpurun_out/bench2_err.txt
