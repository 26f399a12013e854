#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03c; mkdir -p $O
bash scripts/r03_exp.sh $O fwdorder "-DS360_BWD_FWD_ORDER"
bash scripts/r03_exp.sh $O ownorder ""
bash scripts/r03_exp.sh $O fwdorder2 "-DS360_BWD_FWD_ORDER"
