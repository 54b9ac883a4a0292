#!/bin/bash
# usage: REPS=n ab_trees.sh dir1 dir2 ... ; `python bench.py` (timing only) in each built tree, interleaved, one box
mkdir -p gpurun_out
root=$(pwd)
for rep in $(seq 1 ${REPS:-2}); do
for d in "$@"; do
  (cd $d && timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing 2>/dev/null | tail -1 > $root/gpurun_out/abt.json)
  python -c "
import json,sys
try:
    r=json.loads(open('gpurun_out/abt.json').read().strip().splitlines()[-1]); print('%-30s %.2f us' % (sys.argv[1], r['ms_per_step']*1e3))
except Exception as e: print(sys.argv[1], 'FAILED', e)" $d
done; done
