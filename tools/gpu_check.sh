#!/bin/bash
# Developer helper (GPU box): loss parity subset + bench + trace, both tile heights
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x > gpurun_out/pytest_loss.log 2>&1; tail -3 gpurun_out/pytest_loss.log
for R in 8; do
  echo "== tile rows $R"
  BXI_TILE_ROWS=$R timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_R$R.json 2> gpurun_out/bench_R$R.err
  python - $R <<'PY'
import json, sys
R = sys.argv[1]
try:
    r = json.loads(open(f'gpurun_out/bench_R{R}.json').read().strip().splitlines()[-1])
    print('value %.0f img/s  us/step %.2f' % (r['value'], r['ms_per_step'] * 1e3), {k: round(v['avg_us'], 2) for k, v in r.get('kernels', {}).items()}, 'roofline frac %.3f' % r['roofline']['frac'])
except Exception as e:
    print('FAILED', e); print(open(f'gpurun_out/bench_R{R}.err').read()[-2500:])
PY
  BXI_TILE_ROWS=$R python tools/trace_eval.py 2>&1 | grep -v amdgpu.ids
done
