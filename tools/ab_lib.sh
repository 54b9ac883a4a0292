#!/bin/bash
# usage: ab_lib.sh lib1.so lib2.so ... ; bench per library, twice, interleaved (one box)
mkdir -p gpurun_out
i=0
for rep in $(seq 1 ${REPS:-2}); do
for lib in "$@"; do
  i=$((i+1))
  timeout 300 python tools/with_lib.py $lib bench.py --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/abl$i.json 2> gpurun_out/abl$i.err
  python - "$lib" gpurun_out/abl$i.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print('%-60s %.2f us' % (sys.argv[1], r['ms_per_step'] * 1e3))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[2].replace('.json', '.err')).read()[-1500:])
PY
done; done
