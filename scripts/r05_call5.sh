#!/bin/bash
mkdir -p gpurun_out/r05d
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 60 python scripts/split_dbg.py 2>&1 | grep -v amdgpu.ids | tail -3
timeout 90 python scripts/split_diag.py surface_like 5 > gpurun_out/r05d/diag_surface.log 2>&1; echo "diag surf rc $?"
grep -v amdgpu.ids gpurun_out/r05d/diag_surface.log | tail -14
timeout 240 python -m pytest tests/test_gpu_saturating_parity.py -x -q -m gpu > gpurun_out/r05d/t_sat.log 2>&1; echo "sat rc $?"; tail -12 gpurun_out/r05d/t_sat.log
timeout 200 python -m pytest tests/test_gpu_edge_cases.py tests/test_gpu_lean.py tests/test_sh_rotation_wigner.py tests/test_gpu_rccl_single_rank.py -q -m gpu -k "merge_is_race or thin_diagonal or hip_kernel or rccl" > gpurun_out/r05d/t_new.log 2>&1; echo "new rc $?"; tail -8 gpurun_out/r05d/t_new.log
S360_HIPCC_EXTRA=-DS360_DBG_TIMING python -c "from splatter360_amd import _lib; _lib.build(force=True)" > gpurun_out/r05d/build.log 2>&1; echo "build rc $?"
timeout 90 python scripts/fwdtiming.py surface_like 1 > gpurun_out/r05d/fwd_surface_like_split.txt 2>&1; echo "fwd rc $?"
grep -v amdgpu.ids gpurun_out/r05d/fwd_surface_like_split.txt | grep -v "^unit"
