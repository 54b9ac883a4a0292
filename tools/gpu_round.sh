#!/bin/bash
# Developer helper run on the GPU box: bench (eager), rocprofv3 kernel stats of the same command.  Usage: tools/gpu_round.sh [tag]
TAG=${1:-r02}
mkdir -p gpurun_out
timeout 300 python bench.py --no-cpu-baseline > gpurun_out/bench_eager.json 2> gpurun_out/bench_eager.err; python - <<'PY'
import json
try:
    r = json.loads(open('gpurun_out/bench_eager.json').read().strip().splitlines()[-1])
    print('value %.0f img/s  us/step %.2f' % (r['value'], r['ms_per_step'] * 1e3), {k: round(v['avg_us'], 2) for k, v in r.get('kernels', {}).items()},
          'roofline frac %.3f' % r['roofline']['frac'])
    for k in ('pipelined_throughput_extra', 'warm_cache_extra', 'autograd_backward_extra', 'module_api'):
        print(k, r.get(k))
except Exception as e:
    print('FAILED', e); print(open('gpurun_out/bench_eager.err').read()[-2500:])
PY
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eager -o $TAG -- python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing --no-extras > $R/gpurun_out/prof_eager.log 2>&1
cut -d, -f1-4 $R/gpurun_out/prof_eager/${TAG}_kernel_stats.csv | head -8
