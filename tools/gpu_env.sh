#!/bin/bash
mkdir -p gpurun_out
for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v"
  HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_k$v.json 2> gpurun_out/bench_k$v.err
  python - $v <<'PY'
import json, sys
v = sys.argv[1]
try:
    r = json.loads(open(f'gpurun_out/bench_k{v}.json').read().strip().splitlines()[-1])
    print('value %.0f img/s  us/step %.2f' % (r['value'], r['ms_per_step'] * 1e3), {k: round(x['avg_us'], 2) for k, x in r.get('kernels', {}).items()})
except Exception as e:
    print('FAILED', e); print(open(f'gpurun_out/bench_k{v}.err').read()[-1500:])
PY
done
HIP_FORCE_DEV_KERNARG=1 python tools/trace_eval.py 2>&1 | grep -v amdgpu.ids | grep -E "pair:|leaders|count waves|math waves"
