#!/bin/bash
# Run on the GPU box (via gpurun): bench JSON + rocprofv3 kernel stats + PMC HBM counters (separate passes).
# Outputs land in gpurun_out/r01/ ; scripts/make_profiles.py then condenses them into profiles/.
set -x
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python $R/bench.py --steps 20 --warmup 5 > $O/bench_fwdbwd.json 2> $O/bench_fwdbwd.err
python $R/bench.py --steps 20 --warmup 5 --mode fwd --cpu-baseline 0 > $O/bench_fwd.json 2> $O/bench_fwd.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 20 --warmup 5 --cpu-baseline 0 > $O/bench_under_rocprof.json 2>/dev/null
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $O/pmc_$c -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 > /dev/null 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d $O/pmc_SQ1 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 > /dev/null 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_SQ2 -o pmc -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline 0 > /dev/null 2>&1
ls -R $O | head -40
