#!/bin/bash
# Run on the GPU box (via gpurun): bench JSON + rocprofv3 kernel stats + PMC HBM counters (separate passes, as
# MI355X_MICROARCH.md prescribes: --pmc never together with the trace domains other than --kernel-trace) + the side configurations
# (forward only, evaluation shape, BASELINE configs[4] single-rank shape, 2 gloo ranks on the one GPU, ONE RCCL rank) + the traces
# of the RCCL single-rank exchange, the stand-alone adapter kernels and the per-face drop-in training step.
# Outputs land in gpurun_out/$ROUND/ ; scripts/make_profiles.py then condenses them into profiles/.
# meta.json stamps the kernel-source hash the counters belong to (bench.py refuses to quote a mismatching profile).
set -x
ROUND=${ROUND:-r06}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -c "import sys, json; sys.path.insert(0, '$R'); from splatter360_amd import _lib; print(json.dumps(dict(source_hash=_lib.source_hash(), gaussians=1048576, face=256)))" > $O/meta.json
timeout 400 python $R/bench.py > $O/bench_fwdbwd.json 2> $O/bench_fwdbwd.err
timeout 200 python $R/bench.py --mode fwd --cpu-baseline 0 --workloads 0 > $O/bench_fwd.json 2> $O/bench_fwd.err
timeout 200 python $R/bench.py --steps 10 --warmup 3 --mode eval --cpu-baseline 0 > $O/bench_eval.json 2> /dev/null
timeout 300 python $R/bench.py --steps 10 --warmup 3 --pano-h 1024 --cpu-baseline 0 --workloads 0 > $O/bench_c5_4m_fwdbwd.json 2> /dev/null
S360_DIST_BACKEND=gloo S360_FORCE_DEVICE=0 timeout 300 python $R/bench.py --gpus 2 --steps 4 --warmup 2 --cpu-baseline 0 --workloads 0 > $O/bench_2rank_gloo_one_gpu.json 2> /dev/null
timeout 300 python $R/bench.py --cpu-baseline 0 --workloads 0 --single-rank-rccl 1 > $O/bench_1rank_nccl.json 2> $O/bench_1rank_nccl.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline 0 --forward-figure 0 --workloads 0 > $O/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --forward-figure 0 --workloads 0 > /dev/null 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_SQ1 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --forward-figure 0 --workloads 0 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_SQ2 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --forward-figure 0 --workloads 0 > /dev/null 2>&1
# the RCCL single-rank exchange under the kernel trace (RCCL kernels between the backward's per-range kernels)
S360_RCCL_TEST_W=512 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/rccl -o rccl -- python $R/scripts/rccl_single_rank.py > $O/rccl_single_rank.log 2>&1
# the stand-alone adapter kernels + fused raw path, and the per-face drop-in training step of the unchanged reference
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/adapter -o adapter -- python $R/scripts/prof_adapter_dropin.py adapter > $O/prof_adapter.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/dropin -o dropin -- python $R/scripts/prof_adapter_dropin.py dropin > $O/prof_dropin.log 2>&1
# round 6: the 4 M / 512^2 shape with the stress workloads (surface-like: list splitting inside k_render's launch), and the kernel table of
# the 1 M surface-like step with splitting off and on
timeout 400 python $R/bench.py --steps 10 --warmup 3 --pano-h 1024 --cpu-baseline 0 --workloads 1 > $O/bench_c5_4m_workloads.json 2> /dev/null
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/surface -o surface -- python $R/scripts/split_diag.py surface_like 10 > $O/split_diag_surface_like.txt 2>&1
# per-unit timing of both composites (instrumented build; the box is discarded afterwards)
S360_HIPCC_EXTRA=-DS360_DBG_TIMING python -c "import sys; sys.path.insert(0, '$R'); from splatter360_amd import _lib; _lib.build(force=True)"
timeout 120 python $R/scripts/bwdtiming.py encoder_like > $O/bwd_unit_timing.txt 2>/dev/null
timeout 120 python $R/scripts/bwdtiming.py surface_like > $O/bwd_unit_timing_surface_like.txt 2>/dev/null
timeout 120 python $R/scripts/fwdtiming.py encoder_like > $O/fwd_unit_timing_encoder_like.txt 2>/dev/null
timeout 120 python $R/scripts/fwdtiming.py surface_like 1 > $O/fwd_unit_timing_surface_like.txt 2>/dev/null
timeout 120 python $R/scripts/fwdtiming.py surface_like 0 > $O/fwd_unit_timing_surface_like_unsplit.txt 2>/dev/null
# keep only what make_profiles.py reads (gpurun_out is capped at 64 MiB)
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' ! -path '*rccl*' -delete; find $O -name '*agent_info.csv' -delete
# the RCCL trace: keep a time-ordered excerpt (kernel name, start, end) instead of the full csv
python - <<PY
import csv, glob
fs = glob.glob("$O/rccl/**/*kernel_trace.csv", recursive=True)
if fs:
    rows = sorted(csv.DictReader(open(fs[0])), key=lambda r: int(r["Start_Timestamp"]))
    t0 = int(rows[0]["Start_Timestamp"])
    with open("$O/rccl_kernel_sequence.txt", "w") as f:
        f.write("# kernel launches of scripts/rccl_single_rank.py in time order (us since the first launch): the last 400\n")
        for r in rows[-400:]:
            f.write("%10.1f %8.1f  %s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r["Kernel_Name"].split("(")[0][:100]))
PY
find $O -name '*kernel_trace.csv' -delete
du -sh $O
