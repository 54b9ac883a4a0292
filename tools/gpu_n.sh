#!/bin/bash
# usage: gpu_n.sh "ENV1" "ENV2" ...: step time at N = 64 / 128 / 256 in the two-launch form per environment
for cfg in "$@"; do for ipb in 2 4 8; do
  env ${cfg//,/ } timeout 300 python bench.py --no-cpu-baseline --no-extras --inst-per-box $ipb --flags 2 --steps 400 --sets 4 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg ipb $ipb: %.2f us' % (r['ms_per_step']*1e3), {k: round(v['avg_us'],2) for k,v in r.get('kernels',{}).items()})"
done; done
