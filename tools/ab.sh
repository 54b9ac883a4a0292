#!/bin/bash
# usage: ab.sh "CFG1" "CFG2" ... (each a comma-separated env list); runs bench per cfg twice, interleaved
mkdir -p gpurun_out
i=0
for rep in 1 2; do
for cfg in "$@"; do
  i=$((i+1))
  env ${cfg//,/ } timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/ab$i.json 2> gpurun_out/ab$i.err
  python - "$cfg" gpurun_out/ab$i.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print('%-40s %.2f us' % (sys.argv[1], r['ms_per_step'] * 1e3))
except Exception as e:
    print(sys.argv[1], 'FAILED', e)
PY
done; done
