#!/bin/bash
# build the library here (cross-compile), then run a command on the GPU box:  scripts/gpu.sh <timeout-s> '<command>'
set -e
cd "$(dirname "$0")/.."
python -c "from splatter360_amd import _lib; _lib.build(force=False)"
python -c "from oracle import oracle; oracle.build()" 2>/dev/null || true
T=$1; shift
exec /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
