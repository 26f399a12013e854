#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > $O/tests.log; cat $O/tests.log
(cd .exp/base && python bench.py --steps 20 --warmup 5 --cpu-baseline 0 2>/dev/null | tail -1 > ../../$O/base.json)
python - <<P
import json
r=json.load(open("$O/base.json")); k=r['kernels']
print("base", round(r['value'],1), round(r['ms_per_step'],4), {n:round(v['avg_us'],1) for n,v in k.items()}, flush=True)
P
bash scripts/r03_exp.sh $O new ""
bash scripts/r03_exp.sh $O new_fwd "" --mode fwd
bash scripts/r03_exp.sh $O nt "-DS360_SHBWD_NT -DS360_SHEVAL_NT"
bash scripts/r03_exp.sh $O ntst "-DS360_SHBWD_NT"
bash scripts/r03_exp.sh $O w4 "-DS360_BWD_W4 -DS360_BWD_NOPREFETCH"
bash scripts/r03_exp.sh $O nopf "-DS360_BWD_NOPREFETCH"
# leave the default build in place
python -c "
import sys; sys.path.insert(0,'.')
from splatter360_amd import _lib; _lib.build(force=True)"
