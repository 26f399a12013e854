#!/bin/bash
# usage (on the GPU box): r03_exp.sh <outdir> <label> "<hipcc -D flags>" [bench args] -> rebuild with the flags, bench, one summary line
cd $GRAFT_REPO_ROOT
out=$1; label=$2; flags="$3"; shift 3
mkdir -p $out
S360_HIPCC_EXTRA="$flags" python -c "
import sys; sys.path.insert(0,'.')
from splatter360_amd import _lib; _lib.build(force=True)" || exit 1
python bench.py --steps 20 --warmup 5 --cpu-baseline 0 "$@" 2>$out/$label.err | tail -1 > $out/$label.json
python - <<P
import json
r=json.load(open("$out/$label.json")); k=r['kernels']
print("$label", round(r['value'],1), round(r['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in k.items()}, flush=True)
P
