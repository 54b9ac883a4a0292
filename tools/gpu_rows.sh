#!/bin/bash
# Developer helper (GPU box): rocprofv3 kernel stats + timings of the "next" rows and the op-level kernels.
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name script
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$1 -o r01 -- \
      python $GRAFT_REPO_ROOT/tools/$2 > $GRAFT_REPO_ROOT/gpurun_out/$1_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/$1_bench.err
  echo "== $1"; grep "bxi::" $GRAFT_REPO_ROOT/gpurun_out/prof_$1/r01_kernel_stats.csv | cut -d, -f1-2,4 | cut -c1-110
}
run dynamic_head bench_dynamic_head.py
run discobox bench_discobox.py
run levelset bench_levelset.py
run tree_filter bench_tree_filter.py
run pairwise_op bench_pairwise_op.py
