#!/bin/bash
# Developer helper (GPU box), round 3: the single-launch form of the evaluation (BXI_ONE_LAUNCH=1, default) against the two-launch
# form (BXI_ONE_LAUNCH=0): loss-path parity in both, A/B of the bench on one box, per-wave trace
mkdir -p gpurun_out
BXI_ONE_LAUNCH=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu -x > gpurun_out/pytest_one.log 2>&1; tail -4 gpurun_out/pytest_one.log
BXI_ONE_LAUNCH=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_dynamic_head.py -q -m gpu -x > gpurun_out/pytest_two.log 2>&1; tail -4 gpurun_out/pytest_two.log
summ() {
python - "$1" <<'PY'
import json, sys
f = sys.argv[1]
try:
    r = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, 'value %.0f img/s  us/step %.2f' % (r['value'], r['ms_per_step'] * 1e3), {k: (round(v['avg_us'], 2), round(v['raw_event_avg_us'], 2)) for k, v in r.get('kernels', {}).items()})
except Exception as e:
    print(f, 'FAILED', e); print(open(f.replace('.json', '.err')).read()[-2500:])
PY
}
i=0
for cfg in "BXI_ONE_LAUNCH=0" "BXI_ONE_LAUNCH=1" "BXI_ONE_LAUNCH=0" "BXI_ONE_LAUNCH=1" $EXTRA_CFGS; do
  i=$((i+1))
  echo "== $cfg"
  env ${cfg//,/ } timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_one$i.json 2> gpurun_out/bench_one$i.err
  summ gpurun_out/bench_one$i.json
done
