#!/bin/bash
# Developer helper (GPU box): SQ / instruction-cache counters of the evaluation's two kernels (separate --pmc passes).
# Usage: tools/gpu_sq.sh [lib file in boxinstseg_amd/lib]
R=$GRAFT_REPO_ROOT
export BXI_LIB=${1:-libboxinst_hip.so}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null < /dev/null | grep -oiE "SQC?_[A-Z_]*(ICACHE|IFETCH|INST_PREFETCH)[A-Z_]*" | sort -u | tr '\n' ' '; echo
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" "SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  rm -rf $R/gpurun_out/sq_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/sq_$i -o x -- python $R/tools/bench_lib.py --steps 50 --warmup 10 --no-cpu-baseline --no-kernel-timing --no-extras > $R/gpurun_out/sq_$i.log 2>&1 < /dev/null
  f=$(find $R/gpurun_out/sq_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
rd = csv.DictReader(open(sys.argv[1]))
for row in rd:
    k = row['Kernel_Name'].split('(')[0][-40:]
    if 'pair_kernel' in k or 'prep_kernel' in k:
        a = acc[(k, row['Counter_Name'])]; a[0] += float(row['Counter_Value']); a[1] += 1
for (k, c), (v, n) in sorted(acc.items()):
    print('%-42s %-28s %14.0f per launch (%d)' % (k, c, v / n, n))
PY
done
