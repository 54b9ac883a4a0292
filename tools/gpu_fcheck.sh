#!/bin/bash
# Developer helper (GPU box): parity of the section 8(f) rows + per-kernel rocprof stats of their benches.  Usage: tools/gpu_fcheck.sh levelset discobox
R=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $R/gpurun_out
cd $R && timeout 600 python -m pytest tests/test_gpu_levelset.py tests/test_gpu_discobox.py tests/test_gpu_dynamic_head.py -q -m gpu -x 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  rm -rf $R/gpurun_out/prof_$t
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$t -o x -- python $R/tools/bench_$t.py > $R/gpurun_out/${t}_bench.json 2> $R/gpurun_out/${t}_bench.err
  f=$(ls $R/gpurun_out/prof_$t/*kernel_stats.csv $R/gpurun_out/prof_$t/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $t ($f)"
  if [ -n "$f" ]; then grep "bxi::" "$f" | awk -F'","' '{printf "%-90.90s calls %s avg_ns %s\n", $1, $2, $4}'; fi
done
