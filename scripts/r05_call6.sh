#!/bin/bash
mkdir -p gpurun_out/r05e
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 120 python -m pytest tests/test_gpu_raw_entry.py -x -q -m gpu > gpurun_out/r05e/t_raw.log 2>&1; echo "raw rc $?"; tail -12 gpurun_out/r05e/t_raw.log
timeout 90 python scripts/split_diag.py surface_like 5 > gpurun_out/r05e/diag_surface.log 2>&1; echo "diag surf rc $?"
grep -v amdgpu.ids gpurun_out/r05e/diag_surface.log | tail -3
timeout 300 python -m pytest tests/test_gpu_saturating_parity.py -q -m gpu > gpurun_out/r05e/t_sat.log 2>&1; echo "sat rc $?"; tail -6 gpurun_out/r05e/t_sat.log
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/r05e/bench.json 2> gpurun_out/r05e/bench.err; echo "bench rc $?"; tail -3 gpurun_out/r05e/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r05e/bench.json') if l.startswith('{')][0])
    print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    print({k:(round(v['ms_per_step'],3), v.get('split_quadrants'), v['kernels_avg_us']) for k,v in d['workloads'].items()})
    print(d['config']['workspace_bytes_forward'], d['config']['workspace_bytes_backward'], d['config']['max_instances'], d.get('dropin_train',{}).get('ms_per_step'))
    print(json.dumps(d.get('adapter_plus_render'))[:1500])
    print(d.get('cpu_baseline'))
except Exception as e: print("parse", e)
PY
