cd /root/repo
mkdir -p gpurun_out
for args in "--pano-h 512 --face 512" "--pano-h 1024"; do
  for mode in fwdbwd fwd; do
    timeout 600 python bench.py --cpu-baseline 0 --steps 10 --warmup 3 --mode $mode $args 2>&1 | tail -1 > gpurun_out/c5.json
    python - <<PY
import json
d=json.loads(open('gpurun_out/c5.json').read())
print("$args $mode", round(d['value'],1), 'Msplats/s', round(d['ms_per_step'],3), 'ms', 'L=',d['config']['num_rendered'], {k:round(v['avg_us'],1) for k,v in d['kernels'].items()})
PY
    cp gpurun_out/c5.json "gpurun_out/c5_$(echo $args | tr -d ' -')_$mode.json"
  done
done
