#!/bin/bash
# the round-3 tree (git archive of f8d8501 under .r3tree/) against the current one, the driver's invocation and the default, one box
for rep in 1 2 3; do
  for tree in .r3tree .; do
    (cd $tree; python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-kernel-timing 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tree steps 20: %.2f us' % (r['ms_per_step']*1e3))")
    (cd $tree; python bench.py --no-cpu-baseline --no-extras --no-kernel-timing 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$tree default : %.2f us' % (r['ms_per_step']*1e3))")
  done
done
