#!/bin/bash
for cfg in "$@"; do for ipb in ${IPBS:-4 8}; do
  fl=${cfg%%:*}; envs=${cfg#*:}
  env ${envs//,/ } timeout 300 python bench.py --no-cpu-baseline --no-extras --inst-per-box $ipb --flags $fl --steps 400 --sets 4 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flags $fl $envs ipb $ipb: %.2f us' % (r['ms_per_step']*1e3), {k: round(v['avg_us'],2) for k,v in r.get('kernels',{}).items()})"
done; done
