#!/bin/bash
# one box, interleaved: "label|dir|lib|ENV=..." entries; lib empty = the tree's own library
mkdir -p gpurun_out
root=$(pwd)
for rep in $(seq 1 ${REPS:-3}); do
for e in "$@"; do
  IFS='|' read -r label dir lib envs <<< "$e"
  if [ -n "$lib" ]; then cmd="python tools/with_lib.py $lib bench.py"; else cmd="python bench.py"; fi
  (cd $dir && env $envs timeout 300 $cmd --no-cpu-baseline --no-extras --no-kernel-timing $BENCH_ARGS 2>/dev/null | tail -1 > $root/gpurun_out/abt.json)
  python -c "
import json,sys
try:
    r=json.loads(open('gpurun_out/abt.json').read().strip().splitlines()[-1]); print('%-30s %.2f us' % (sys.argv[1], r['ms_per_step']*1e3))
except Exception as e: print(sys.argv[1], 'FAILED', e)" $label
done; done
