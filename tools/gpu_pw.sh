#!/bin/bash
# Developer helper (GPU box): op-level pairwise_nlog parity, rocprofv3 kernel stats and PMC counters -> gpurun_out/
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 python -m pytest $R/tests/test_gpu_parity.py -q -m gpu -x -k "pairwise_op" -p no:cacheprovider > $R/gpurun_out/pytest_pw.log 2>&1; tail -1 $R/gpurun_out/pytest_pw.log
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_pairwise_op -o r03 -- python $R/tools/bench_pairwise_op.py > $R/gpurun_out/pairwise_op_bench.json 2> $R/gpurun_out/pairwise_op_bench.err
python3 - $R/gpurun_out/prof_pairwise_op/r03_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pairwise' in r['Name']:
        print(r['Name'][:64], r['Calls'], round(float(r['AverageNs'])/1e3, 2), round(float(r['MinNs'])/1e3, 2))
PY
