#!/bin/bash
# usage: REPS=n ab_lib_r3.sh lib1.so ... ; like ab_lib.sh, with the round-3 tree (.r3tree, if present) as one more contestant in every round
mkdir -p gpurun_out
i=0
one() { python -c "
import json,sys
try:
    r=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print('%-60s %.2f us' % (sys.argv[1], r['ms_per_step']*1e3))
except Exception as e: print(sys.argv[1], 'FAILED', e)" "$1" "$2"; }
for rep in $(seq 1 ${REPS:-2}); do
for lib in "$@"; do
  i=$((i+1))
  timeout 300 python tools/with_lib.py $lib bench.py --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/abl$i.json 2> gpurun_out/abl$i.err
  one $lib gpurun_out/abl$i.json
done
if [ -d .r3tree ]; then (cd .r3tree; timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing > ../gpurun_out/abl_r3.json 2>/dev/null); one r3tree gpurun_out/abl_r3.json; fi
done
