#!/bin/bash
mkdir -p gpurun_out/r05c
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python scripts/split_diag.py surface_like 5 > gpurun_out/r05c/diag_surface.log 2>&1; echo "diag surf rc $?"
timeout 300 python scripts/split_diag.py encoder_like 5 > gpurun_out/r05c/diag_encoder.log 2>&1; echo "diag enc rc $?"
grep -v amdgpu.ids gpurun_out/r05c/diag_surface.log | tail -16; grep -v amdgpu.ids gpurun_out/r05c/diag_encoder.log | tail -4
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r05c/t_all.log 2>&1; echo "all rc $?"; tail -15 gpurun_out/r05c/t_all.log
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r05c/bench.json 2> gpurun_out/r05c/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05c/bench.json') if l.startswith('{')][0])
print(d['value'], d['ms_per_step'], d['forward_only'], {k:v['avg_us'] for k,v in d['kernels'].items()})
print({k:(v['ms_per_step'], v.get('split_quadrants'), v['kernels_avg_us']) for k,v in d['workloads'].items()})
print(d['config']['workspace_bytes_forward'], d['config']['workspace_bytes_backward'], d.get('dropin_train'), d.get('cpu_baseline'))
PY
S360_HIPCC_EXTRA=-DS360_DBG_TIMING python -c "from splatter360_amd import _lib; _lib.build(force=True)" > gpurun_out/r05c/build.log 2>&1; echo "build rc $?"
timeout 300 python scripts/fwdtiming.py surface_like 1 > gpurun_out/r05c/fwd_surface_like_split.txt 2>&1; echo "fwd rc $?"
grep -v amdgpu.ids gpurun_out/r05c/fwd_surface_like_split.txt | grep -v "^unit"
