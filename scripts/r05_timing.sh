#!/bin/bash
# per-wave timing of the composites (DBG_TIMING build, on the GPU box only) + function check of the normal build first
mkdir -p gpurun_out/r05t
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 300 python scripts/split_diag.py surface_like 5 > gpurun_out/r05t/diag_surface.log 2>&1; echo "diag surf rc $?"
timeout 300 python scripts/split_diag.py encoder_like 5 > gpurun_out/r05t/diag_encoder.log 2>&1; echo "diag enc rc $?"
timeout 300 python scripts/split_diag.py uniform 3 > gpurun_out/r05t/diag_uniform.log 2>&1; echo "diag uni rc $?"
grep -v amdgpu.ids gpurun_out/r05t/diag_surface.log | tail -22; grep -v amdgpu.ids gpurun_out/r05t/diag_encoder.log | tail -14;  grep -v amdgpu.ids gpurun_out/r05t/diag_uniform.log | tail -14
S360_HIPCC_EXTRA=-DS360_DBG_TIMING python -c "from splatter360_amd import _lib; _lib.build(force=True)" > gpurun_out/r05t/build.log 2>&1; echo "build rc $?"
for c in surface_like encoder_like; do
  timeout 300 python scripts/fwdtiming.py $c 1 > gpurun_out/r05t/fwd_${c}_split.txt 2>&1; echo "fwd $c rc $?"
done
cat gpurun_out/r05t/fwd_surface_like_split.txt gpurun_out/r05t/fwd_encoder_like_split.txt | grep -v amdgpu.ids
