cd /root/repo
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8
for fl in 1 0; do timeout 300 python bench.py --cpu-baseline 0 --fused-loss $fl 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused_loss=$fl', round(d['value'],1), round(d['ms_per_step'],4), {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})"; done
