cd /root/repo
timeout 900 python -m pytest tests/test_gpu_factored_sync.py -x -q 2>&1 | tail -15
echo ---- 2-rank gloo bench factored
export S360_DIST_BACKEND=gloo S360_FORCE_DEVICE=0
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-baseline 0 2>&1 | tail -3 | cut -c1-600
