#!/bin/bash
mkdir -p gpurun_out/r05g
cd "$GRAFT_REPO_ROOT"
export PYTHONUNBUFFERED=1
timeout 1500 python -m pytest tests -q -m gpu --timeout 400 -p no:cacheprovider > gpurun_out/r05g/t_all.log 2>&1; echo "all rc $?"
tail -25 gpurun_out/r05g/t_all.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
