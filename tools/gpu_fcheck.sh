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
  if [ -n "$f" ]; then python3 - "$f" <<'PY'
import csv, sys
for row in csv.reader(open(sys.argv[1])):
    if 'bxi::' in row[0]:
        print('%-70.70s calls %5s avg_us %8.2f min %8.2f max %8.2f' % (row[0], row[1], float(row[3]) / 1e3, float(row[5]) / 1e3, float(row[6]) / 1e3))
PY
  fi
done
