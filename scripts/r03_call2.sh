#!/bin/bash
cd $GRAFT_REPO_ROOT
bash scripts/quickprof.sh
S360_HIPCC_EXTRA="-DS360_DBG_TIMING" python -c "
import sys; sys.path.insert(0,'.')
from splatter360_amd import _lib; _lib.build(force=True)"
python scripts/bwdtiming.py
