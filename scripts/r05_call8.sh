#!/bin/bash
mkdir -p gpurun_out/r05h
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 60 python scripts/split_dbg.py 2>&1 | grep -v amdgpu.ids | tail -2
timeout 90 python scripts/split_diag.py surface_like 5 > gpurun_out/r05h/diag_surface.log 2>&1; echo "diag surf rc $?"
grep -v amdgpu.ids gpurun_out/r05h/diag_surface.log | tail -12 | head -9; grep -v amdgpu.ids gpurun_out/r05h/diag_surface.log | tail -2
timeout 200 python -m pytest tests/test_gpu_saturating_parity.py tests/test_gpu_raw_entry.py -q -m gpu --timeout 150 > gpurun_out/r05h/t.log 2>&1; echo "sat+raw rc $?"; tail -3 gpurun_out/r05h/t.log
timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline 0 > gpurun_out/r05h/bench.json 2> gpurun_out/r05h/bench.err; echo "bench rc $?"; tail -2 gpurun_out/r05h/bench.err
python - <<'PY'
import json
try:
    d=json.loads([l for l in open('gpurun_out/r05h/bench.json') if l.startswith('{')][0])
    print(d['value'], d['ms_per_step'], d['forward_only']['ms_per_step'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
    print({k:(round(v['ms_per_step'],3), v.get('split_quadrants'), v.get('split_errors'), v['kernels_avg_us']) for k,v in d['workloads'].items()})
    a=d.get('adapter_plus_render'); print({k:(v if not isinstance(v,dict) else (round(v['ms_per_step'],3), v['kernels_avg_us'].get('sh_eval'), v['kernels_avg_us'].get('sh_bwd'))) for k,v in a.items() if k in ('two_step','fused_raw','fused_over_two_step','error')})
except Exception as e: print("parse", e)
PY
