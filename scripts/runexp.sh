for cfg in "" "S360_NO_ORDER=1"; do env $cfg python bench.py --steps 10 --warmup 3 --cpu-baseline 0 2>&1 | tail -1 | python -c "
import json,sys
r=json.loads(sys.stdin.read()); k=r['kernels']; print('$cfg', round(r['value'],1), round(k['render']['avg_us'],1), round(k['render_bwd']['avg_us'],1))"; done
