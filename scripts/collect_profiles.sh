#!/bin/bash
# Run on the GPU box (via gpurun): bench JSON + rocprofv3 kernel stats + PMC HBM counters (separate passes, as
# MI355X_MICROARCH.md prescribes: --pmc never together with the trace domains other than --kernel-trace) + the side configurations
# (forward only, evaluation shape, BASELINE configs[4] single-rank shape, 2 gloo ranks on the one GPU).
# Outputs land in gpurun_out/$ROUND/ ; scripts/make_profiles.py then condenses them into profiles/.
# meta.json stamps the kernel-source hash the counters belong to (bench.py refuses to quote a mismatching profile).
set -x
ROUND=${ROUND:-r04}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$ROUND
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python -c "import sys, json; sys.path.insert(0, '$R'); from splatter360_amd import _lib; print(json.dumps(dict(source_hash=_lib.source_hash(), gaussians=1048576, face=256)))" > $O/meta.json
python $R/bench.py --steps 20 --warmup 5 > $O/bench_fwdbwd.json 2> $O/bench_fwdbwd.err
python $R/bench.py --steps 20 --warmup 5 --mode fwd --cpu-baseline 0 --workloads 0 > $O/bench_fwd.json 2> $O/bench_fwd.err
python $R/bench.py --steps 10 --warmup 3 --mode eval --cpu-baseline 0 > $O/bench_eval.json 2> /dev/null
python $R/bench.py --steps 10 --warmup 3 --pano-h 1024 --cpu-baseline 0 --workloads 0 > $O/bench_c5_4m_fwdbwd.json 2> /dev/null
python $R/bench.py --steps 10 --warmup 3 --pano-h 1024 --mode fwd --cpu-baseline 0 --workloads 0 > $O/bench_c5_4m_fwd.json 2> /dev/null
S360_DIST_BACKEND=gloo S360_FORCE_DEVICE=0 python $R/bench.py --gpus 2 --steps 4 --warmup 2 --cpu-baseline 0 --workloads 0 > $O/bench_2rank_gloo_one_gpu.json 2> /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline 0 --forward-figure 0 --workloads 0 > $O/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --forward-figure 0 --workloads 0 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_SQ1 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --forward-figure 0 --workloads 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_SQ2 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 --forward-figure 0 --workloads 0 > /dev/null 2>&1
# per-unit timing of both composites (instrumented build), then the production library again
S360_HIPCC_EXTRA=-DS360_DBG_TIMING python -c "import sys; sys.path.insert(0, '$R'); from splatter360_amd import _lib; _lib.build(force=True)"
python $R/scripts/bwdtiming.py > $O/bwd_unit_timing.txt 2>/dev/null
python $R/scripts/fwdtiming.py encoder_like > $O/fwd_unit_timing_encoder_like.txt 2>/dev/null
python $R/scripts/fwdtiming.py surface_like > $O/fwd_unit_timing_surface_like.txt 2>/dev/null
python -c "import sys; sys.path.insert(0, '$R'); from splatter360_amd import _lib; _lib.build(force=True)"
# keep only what make_profiles.py reads (gpurun_out is capped at 64 MiB)
find $O -name '*.db' -delete; find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete
du -sh $O
