#!/bin/bash
# round 5, first GPU call: new split-list path diagnostics + the tests written this round
mkdir -p gpurun_out/r05a
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fused_loss.py -x -q -m gpu > gpurun_out/r05a/t_small.log 2>&1; echo "small rc $?" 
timeout 400 python scripts/split_diag.py surface_like 5 > gpurun_out/r05a/diag_surface.log 2>&1; echo "diag surf rc $?"
timeout 300 python scripts/split_diag.py encoder_like 5 > gpurun_out/r05a/diag_encoder.log 2>&1; echo "diag enc rc $?"
timeout 900 python -m pytest tests/test_gpu_saturating_parity.py -q -m gpu > gpurun_out/r05a/t_sat.log 2>&1; echo "sat rc $?"
timeout 600 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_lean.py tests/test_sh_rotation_wigner.py tests/test_gpu_rccl_single_rank.py -q -m gpu -k "merge_is_race or thin_diagonal or hip_kernel or rccl or small_scene or overflow" > gpurun_out/r05a/t_new.log 2>&1; echo "new rc $?"
timeout 300 python bench.py --steps 10 --warmup 3 --cpu-baseline 0 > gpurun_out/r05a/bench.json 2> gpurun_out/r05a/bench.err; echo "bench rc $?"
tail -5 gpurun_out/r05a/t_small.log; cat gpurun_out/r05a/diag_surface.log | tail -25; tail -12 gpurun_out/r05a/diag_encoder.log; tail -15 gpurun_out/r05a/t_sat.log; tail -8 gpurun_out/r05a/t_new.log; head -c 1500 gpurun_out/r05a/bench.json; tail -3 gpurun_out/r05a/bench.err
