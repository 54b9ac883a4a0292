#!/bin/bash
# Developer helper run on the GPU box: parity tests, trace, bench (eager+graph), rocprof kernel stats.
mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x > gpurun_out/pytest.log 2>&1; tail -4 gpurun_out/pytest.log
python tools/trace_eval.py > gpurun_out/trace.log 2>&1; tail -1 gpurun_out/trace.log
timeout 300 python bench.py --mode eager --no-cpu-baseline --no-pipelined > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; python - <<'PY'
import json
for m in ('eager',):
    try:
        r = json.loads(open(f'gpurun_out/bench_{m}.json').read().strip().splitlines()[-1])
        print(m, 'value %.0f img/s  ms/step %.4f' % (r['value'], r['ms_per_step']), {k: round(v['avg_us'], 2) for k, v in r.get('kernels', {}).items()})
    except Exception as e:
        print(m, 'FAILED', e); print(open(f'gpurun_out/bench_{m}.err').read()[-1500:])
PY
timeout 300 python bench.py --mode graph --no-cpu-baseline --no-kernel-timing --no-pipelined > gpurun_out/bench_graph.json 2> gpurun_out/bench_graph.err; tail -c 400 gpurun_out/bench_graph.json | head -c 400; echo
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_eager -o r01 -- python $GRAFT_REPO_ROOT/bench.py --mode eager --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing --no-pipelined > $GRAFT_REPO_ROOT/gpurun_out/prof_eager.log 2>&1
cut -d, -f1-4 $GRAFT_REPO_ROOT/gpurun_out/prof_eager/r01_kernel_stats.csv | head -8
