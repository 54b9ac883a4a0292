#!/bin/bash
# Developer helper (GPU box), round 3: loss-path parity, A/B of the round-2 (BXI_EVAL_V2=1) and round-3 kernels, per-wave trace
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_dynamic_head.py -q -m gpu -x > gpurun_out/pytest_v3.log 2>&1; tail -4 gpurun_out/pytest_v3.log
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
for cfg in "BXI_EVAL_V2=1" "BXI_X=0" "BXI_EVAL_V2=1" "BXI_X=0"; do
  i=$((i+1))
  echo "== $cfg"
  env $cfg timeout 300 python bench.py --no-cpu-baseline --no-extras > gpurun_out/bench_ab$i.json 2> gpurun_out/bench_ab$i.err
  summ gpurun_out/bench_ab$i.json
done
BXI_POOL_FIRST=0 timeout 300 python tools/trace_eval.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/block_trace_pf0.txt
