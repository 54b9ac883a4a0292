#!/bin/bash
# Developer helper (GPU box): A/B two library builds at 32 / 64 / 128 instances.  Usage: tools/gpu_ab_n.sh libA.so libB.so
for ipb in 1 2 4; do
  for l in "$@"; do
    BXI_LIB=$l timeout 200 python tools/bench_lib.py --inst-per-box $ipb --no-cpu-baseline --no-extras 2>/dev/null < /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('inst/box $ipb', '$l', round(r['ms_per_step']*1e3,2), {k: round(v['avg_us'],2) for k,v in r['kernels'].items()})"
  done
done
