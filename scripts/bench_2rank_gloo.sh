#!/bin/bash
# exercises the N>1 code path of bench.py on a ONE-GPU box: 2 processes sharing cuda:0, gloo collectives (functional check,
# not a performance number)
cd $GRAFT_REPO_ROOT
export S360_DIST_BACKEND=gloo S360_FORCE_DEVICE=0
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --cpu-baseline 0 "$@" 2>&1 | grep -v amdgpu.ids | tail -3 | cut -c1-900
