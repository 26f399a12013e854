#!/bin/bash
# GPU box: rebuild the library with a few tuning macros of the split composite and time the step (scripts/split_diag.py)
# usage: split_variants.sh "<cloud> [<cloud> ...]" <steps> "<flags>" ...
cd $GRAFT_REPO_ROOT
clouds=$1; steps=$2; shift; shift
for cfg in "$@"; do
  S360_HIPCC_EXTRA="$cfg" python -c "from splatter360_amd import _lib; _lib.build(force=True)" || continue
  for cloud in $clouds; do
    echo "== $cloud $cfg"
    timeout 400 python scripts/split_diag.py $cloud $steps 2>&1 | grep "split True step\|split False step\|segment work\|n_contrib mism\|image |diff" | cut -c1-400
  done
done
