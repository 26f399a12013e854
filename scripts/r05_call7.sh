#!/bin/bash
mkdir -p gpurun_out/r05f
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 90 python scripts/split_diag.py surface_like 5 > gpurun_out/r05f/diag_surface.log 2>&1; echo "diag surf rc $?"
grep -v amdgpu.ids gpurun_out/r05f/diag_surface.log | tail -12 | head -9; grep -v amdgpu.ids gpurun_out/r05f/diag_surface.log | tail -2
timeout 300 python -m pytest tests/test_gpu_saturating_parity.py tests/test_gpu_raw_entry.py -q -m gpu > gpurun_out/r05f/t_sat.log 2>&1; echo "sat+raw rc $?"; tail -4 gpurun_out/r05f/t_sat.log
timeout 400 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 > gpurun_out/r05f/bench.json 2> gpurun_out/r05f/bench.err; echo "bench rc $?"; tail -3 gpurun_out/r05f/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r05f/bench.json') if l.startswith('{')][0])
    print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    print({k:(round(v['ms_per_step'],3), v.get('split_quadrants'), v.get('split_errors'), v['kernels_avg_us']) for k,v in d['workloads'].items()})
    print(d['config']['split_flag_set_in_timed_steps'], d['config']['workspace_bytes_forward'], d['config']['workspace_bytes_backward'])
    a=d.get('adapter_plus_render'); print({k:(v if not isinstance(v,dict) else (round(v['ms_per_step'],3), v['kernels_avg_us'].get('sh_eval'), v['kernels_avg_us'].get('sh_bwd'))) for k,v in a.items() if k in ('two_step','fused_raw','fused_over_two_step','error')})
except Exception as e: print("parse", e)
PY
