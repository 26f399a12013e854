#!/bin/bash
# GPU box: rebuild the library with extra hipcc flags and print the headline step + the composites' times (bench.py, HIP events)
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  S360_HIPCC_EXTRA="$cfg" python -c "from splatter360_amd import _lib; _lib.build(force=True)" 2>&1 | tail -1 || continue
  python bench.py --cpu-baseline 0 --workloads 0 --forward-figure 0 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); k=d['kernels']
print('== $cfg ::', round(d['ms_per_step'],4), {n:round(k[n]['avg_us'],1) for n in ('render','render_bwd','preprocess','sort_tiles','emit','gather_slots','preprocess_bwd')})"
done
