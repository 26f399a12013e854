#!/bin/bash
# gpurun helper: build with extra hipcc flags ($1), run the default bench without the side workloads, print the headline + kernels
export S360_HIPCC_EXTRA="$1"
R=$GRAFT_REPO_ROOT
python -c "from splatter360_amd import _lib; _lib.build(force=True)" || exit 1
python $R/bench.py --cpu-baseline 0 --workloads ${2:-0} > /tmp/b.json 2>/tmp/b.err || { tail -5 /tmp/b.err; exit 1; }
python - <<'PY'
import json
d = json.load(open('/tmp/b.json'))
print(round(d['value'], 1), round(d['ms_per_step'], 4), 'fwd', round(d['forward_only']['ms_per_step'], 4))
print({k: round(v['avg_us'], 1) for k, v in d['kernels'].items()})
for k, v in (d.get('workloads') or {}).items():
    print(k, round(v['ms_per_step'], 3), v['kernels_avg_us'])
PY
