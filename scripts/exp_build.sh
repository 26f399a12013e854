#!/bin/bash
# usage (on the GPU box): exp_build.sh "<hipcc -D flags>" <label> [bench args]  -> rebuilds the library with the flags, runs the bench
cd $GRAFT_REPO_ROOT
flags="$1"; label="$2"; shift 2
S360_HIPCC_EXTRA="$flags" python -c "
import sys; sys.path.insert(0,'.')
from splatter360_amd import _lib; _lib.build(force=True)" || exit 1
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 "$@" 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']; print('$label', round(r['value'],1), round(r['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in k.items()})"
