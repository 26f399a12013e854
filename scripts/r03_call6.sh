#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -3
bash scripts/r03_exp.sh $O base ""
bash scripts/r03_exp.sh $O tail1024 "-DS360_MERGE_TAIL_GRID=1024"
bash scripts/r03_exp.sh $O ppt8 "-DS360_EMIT_PPT=8"
bash scripts/r03_exp.sh $O ppt2 "-DS360_EMIT_PPT=2"
python -c "
import sys; sys.path.insert(0,'.')
from splatter360_amd import _lib; _lib.build(force=True)"
