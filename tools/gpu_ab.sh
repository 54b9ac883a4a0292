#!/bin/bash
# Developer helper (GPU box): A/B the evaluation between builds of the library on the same box.
# Usage: tools/gpu_ab.sh libboxinst_hip_base.so libboxinst_hip.so   (file names in boxinstseg_amd/lib; two rounds each)
for rep in 1 2; do
  for l in "$@"; do
    BXI_LIB=$l timeout 200 python tools/bench_lib.py --no-cpu-baseline --no-extras 2>/dev/null < /dev/null | python -c "import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', round(r['ms_per_step']*1e3,2), {k: round(v['avg_us'],2) for k,v in r['kernels'].items()})"
  done
done
