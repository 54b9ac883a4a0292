#!/bin/bash
# Developer helper (GPU box): dynamic-head timings + rocprofv3 kernel stats of the same command.
mkdir -p gpurun_out
python tools/bench_dynamic_head.py > gpurun_out/dyn_bench.json 2> gpurun_out/dyn_bench.err; tail -c 700 gpurun_out/dyn_bench.json
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dyn -o r01 -- python $GRAFT_REPO_ROOT/tools/bench_dynamic_head.py > /dev/null 2>&1
grep "bxi::" $GRAFT_REPO_ROOT/gpurun_out/prof_dyn/r01_kernel_stats.csv | cut -c1-160
