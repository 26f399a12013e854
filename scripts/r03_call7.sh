#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
bash scripts/r03_exp.sh $O prec3 ""
bash scripts/r03_exp.sh $O prec4 "-DS360_PREC_F4=4"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_gpu_edge_cases.py -m gpu -q -x 2>&1 | tail -3
bash scripts/r03_exp.sh $O prec3b ""
