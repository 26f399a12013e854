#!/bin/bash
# GPU box: kernel stats + PMC passes of scripts/raw_prof.py -> gpurun_out/$1/   (counters in their own passes, --kernel-trace only)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o raw -- python $R/scripts/raw_prof.py 6 > $O/stats.log 2>&1
timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc1 -o pmc -- python $R/scripts/raw_prof.py 3 > $O/pmc1.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS --kernel-trace --output-format csv -d $O/pmc2 -o pmc -- python $R/scripts/raw_prof.py 3 > $O/pmc2.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/scripts/raw_prof.py 3 > $O/pmc_$c.log 2>&1
done
rocprofv3 -L 2>/dev/null | grep -i -o "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*\|SQ_INSTS_[A-Z_0-9]*\|SQ_[A-Z_]*LDS[A-Z_0-9]*" | sort -u > $O/counters_available.txt
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
python - <<PY
import csv, glob, collections
O = "$O"
for f in sorted(glob.glob(O + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "raw" in k or "adapter" in k or "sh_" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, d in acc.items():
        print(k[:40], {c: round(sum(v) / len(v)) for c, v in d.items()})
for f in glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["AverageNs"]) > 30000: print("%-50s %s %8.1f us" % (r["Name"].split("(")[0][:50], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
du -sh $O
