#!/bin/bash
# Developer helper (GPU box): per-kernel SQ counters of any command, one rocprofv3 --pmc pass per counter set.
# Usage: tools/gpu_pmc.sh <kernel-name substring> "<set 1>" "<set 2>" ... -- <command>
R=$GRAFT_REPO_ROOT
pat=$1; shift
sets=()
while [ "$1" != "--" ] && [ $# -gt 0 ]; do sets+=("$1"); shift; done
shift
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for set in "${sets[@]}"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_$i
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$i -o x -- "$@" > $R/gpurun_out/pmc_$i.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$pat" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    k = row['Kernel_Name']
    if sys.argv[2] in k:
        a = acc[(k.split('(')[0][-44:], row['Counter_Name'])]; a[0] += float(row['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print('%-46s %-26s %14.0f per launch (%d)' % (k, c, v / n, n))
PY
done
