#!/bin/bash
set -x
ROUND=r05
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/rccl
S360_RCCL_TEST_W=512 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rccl -o rccl -- python $R/scripts/rccl_single_rank.py > $O/rccl_single_rank.log 2>&1
grep "rccl single rank" $O/rccl_single_rank.log
S360_HIPCC_EXTRA=-DS360_DBG_TIMING python -c "import sys; sys.path.insert(0, '$R'); from splatter360_amd import _lib; _lib.build(force=True)"
timeout 120 python $R/scripts/bwdtiming.py encoder_like > $O/bwd_unit_timing.txt 2>/dev/null
timeout 120 python $R/scripts/bwdtiming.py surface_like > $O/bwd_unit_timing_surface_like.txt 2>/dev/null
timeout 120 python $R/scripts/fwdtiming.py encoder_like > $O/fwd_unit_timing_encoder_like.txt 2>/dev/null
timeout 120 python $R/scripts/fwdtiming.py surface_like 1 > $O/fwd_unit_timing_surface_like.txt 2>/dev/null
find $O -name '*.db' -delete; find $O -name '*agent_info.csv' -delete
python - <<PY
import csv, glob
fs = glob.glob("$O/rccl/**/*kernel_trace.csv", recursive=True)
if fs:
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    # the last chunked-exchange backward: from its k_order_units to the end
    idx = [i for i, r in enumerate(rows) if "k_order_units" in r["Kernel_Name"]]
    lo = idx[-3] if len(idx) >= 3 else 0
    t0 = int(rows[lo]["Start_Timestamp"])
    with open("$O/rccl_kernel_sequence.txt", "w") as f:
        f.write("# scripts/rccl_single_rank.py under rocprofv3 --kernel-trace: device activity in time order (us since this backward's first\\n"
                "# launch, duration us, name) of one chunked-exchange backward with force_collectives on a ONE-rank RCCL communicator and the steps\\n"
                "# after it.  With a single rank RCCL implements all-reduce / all-gather as device copies (__amd_rocclr_copyBuffer): no ring kernel\\n"
                "# exists to show; what the trace shows is the per-range copies issued on the communicator's stream between the backward's\\n"
                "# per-range kernels (k_preprocess_bwd, k_sh_bwd).\\n")
        for r in rows[lo:lo + 260]:
            f.write("%10.1f %8.1f  %s\\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:100]))
PY
find $O -name '*kernel_trace.csv' -delete
