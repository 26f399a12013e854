#!/bin/bash
# PMC counters of the backward composite only (developer aid): bash scripts/pmc_one.sh [env assignments...]
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/pmc_one
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 > /dev/null 2>&1
python - <<'P'
import csv, collections, glob, os
f = glob.glob(os.environ["GRAFT_REPO_ROOT"] + "/gpurun_out/pmc_one/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("s360::", "")
    if "render_bwd" in k or "k_render<" in k:
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in acc.items():
    print(k, {n: round(sum(v) / len(v) / 1e6, 2) for n, v in c.items()})
P
