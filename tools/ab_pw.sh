#!/bin/bash
# usage: ab_pw.sh lib1.so lib2.so ...: rocprofv3 kernel stats of the op-level pairwise kernels per library build (one box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do
for lib in "$@"; do
  rm -rf /tmp/prof_ab
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_ab -o x -- python $R/tools/with_lib.py $R/$lib $R/tools/bench_pairwise_op.py > /dev/null 2>&1
  python3 - /tmp/prof_ab/x_kernel_stats.csv "$lib" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if 'pairwise3' in r['Name'] and 'double' not in r['Name']:
        print('%-44s' % sys.argv[2][-44:], r['Name'][10:45], r['Calls'], round(float(r['AverageNs'])/1e3, 2), round(float(r['MinNs'])/1e3, 2))
PY
done; done
