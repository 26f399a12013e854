#!/bin/bash
# quick rocprofv3 kernel stats of the default bench (gpurun): prints the top kernels
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/quick
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o q -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline 0 "$@" > $O/bench.json 2>/dev/null
f=$(find $O -name '*kernel_stats.csv' | head -1)
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:28]:
    print("%-80s %5s %10.1f" % (r["Name"][:80], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
