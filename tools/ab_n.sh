#!/bin/bash
# usage: ab_n.sh <inst-per-box> CFG1 CFG2 ... : bench at 32 x inst-per-box instances per configuration (comma-separated env lists), twice
mkdir -p gpurun_out
ipb=$1; shift
i=0
for rep in 1 2; do
for cfg in "$@"; do
  i=$((i+1))
  env ${cfg//,/ } timeout 300 python bench.py --inst-per-box $ipb --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/abn$i.json 2> gpurun_out/abn$i.err
  python - "$cfg" gpurun_out/abn$i.json <<'PY'
import json, sys
try:
    r = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print('%-44s N=%d %.2f us  %s' % (sys.argv[1], r['config']['instances'], r['ms_per_step'] * 1e3, r['config']['kernels_per_step']))
except Exception as e:
    print(sys.argv[1], 'FAILED', e); print(open(sys.argv[2].replace('.json', '.err')).read()[-800:])
PY
done; done
