#!/bin/bash
# usage: gpu_n1.sh ipb flags "ENV1" "ENV2" ...
ipb=$1; fl=$2; shift 2
for rep in 1 2; do for cfg in "$@"; do
  env ${cfg//,/ } timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --inst-per-box $ipb --flags $fl --steps 600 --sets 6 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ipb $ipb flags $fl $cfg: %.2f us' % (r['ms_per_step']*1e3), r['config']['kernels_per_step'])"
done; done
