#!/bin/bash
# round 4, call A: the new tests + a baseline bench on this round's box
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py tests/test_gpu_dist.py -q -m gpu --maxfail=25 -p no:cacheprovider > gpurun_out/r4a_pytest.log 2>&1
tail -40 gpurun_out/r4a_pytest.log
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing > gpurun_out/r4a_bench$i.json 2> gpurun_out/r4a_bench$i.err; python - <<PY
import json
try:
    r = json.loads(open('gpurun_out/r4a_bench$i.json').read().strip().splitlines()[-1]); print('bench default: %.2f us' % (r['ms_per_step'] * 1e3), r['config']['kernels_per_step'])
except Exception as e:
    print('bench FAILED', e); print(open('gpurun_out/r4a_bench$i.err').read()[-2000:])
PY
done
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --mode graph > gpurun_out/r4a_graph.json 2> gpurun_out/r4a_graph.err; tail -c 600 gpurun_out/r4a_graph.json; tail -3 gpurun_out/r4a_graph.err
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --flags 2 > gpurun_out/r4a_two.json 2> gpurun_out/r4a_two.err; python -c "
import json; r=json.loads(open('gpurun_out/r4a_two.json').read().strip().splitlines()[-1]); print('two launches: %.2f us' % (r['ms_per_step']*1e3))"
timeout 300 python bench.py --no-cpu-baseline --no-extras --no-kernel-timing --flags 4 > gpurun_out/r4a_nostay.json 2> gpurun_out/r4a_nostay.err; python -c "
import json; r=json.loads(open('gpurun_out/r4a_nostay.json').read().strip().splitlines()[-1]); print('single, no stay-on: %.2f us' % (r['ms_per_step']*1e3))"
