cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
python bench.py --cpu-baseline 0 --mode eval 2>/dev/null | tail -1 > gpurun_out/r01/bench_eval.json
python bench.py --cpu-baseline 0 --steps 10 --warmup 3 --pano-h 512 --face 512 2>/dev/null | tail -1 > gpurun_out/r01/bench_c5_1m_fwdbwd.json
python bench.py --cpu-baseline 0 --steps 10 --warmup 3 --pano-h 512 --face 512 --mode fwd 2>/dev/null | tail -1 > gpurun_out/r01/bench_c5_1m_fwd.json
python bench.py --cpu-baseline 0 --steps 10 --warmup 3 --pano-h 1024 2>/dev/null | tail -1 > gpurun_out/r01/bench_c5_4m_fwdbwd.json
python bench.py --cpu-baseline 0 --steps 10 --warmup 3 --pano-h 1024 --mode fwd 2>/dev/null | tail -1 > gpurun_out/r01/bench_c5_4m_fwd.json
python scripts/dropin_time.py 2>&1 | tail -1 > gpurun_out/r01/dropin.txt
for f in eval c5_1m_fwdbwd c5_1m_fwd c5_4m_fwdbwd c5_4m_fwd; do python -c "
import json; d=json.load(open('gpurun_out/r01/bench_$f.json')); print('$f', round(d['value'],1), round(d['ms_per_step'],3), {k:round(v['avg_us']) for k,v in d['kernels'].items()})"; done; cat gpurun_out/r01/dropin.txt
