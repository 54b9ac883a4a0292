#!/bin/bash
# usage (GPU box): tools/micro/pmc_pw.sh <binary> : per-kernel PMC sums of the micro harness, one pass per counter set
B=$1; R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum" "FETCH_SIZE" "SQ_INSTS_VMEM SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum TCP_TA_DATA_STALL_CYCLES_sum" "TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc$i -o p -- $R/$B > /tmp/pmc$i.log 2>&1
  f=$(find /tmp/pmc$i -name '*counter_collection.csv' | head -1)
  [ -z "$f" ] && { echo "set '$set': no output"; tail -3 /tmp/pmc$i.log; continue; }
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for r in csv.DictReader(open(sys.argv[1])):
    k = (r['Kernel_Name'][:60], r['Counter_Name']); acc[k][0] += float(r['Counter_Value']); acc[k][1] += 1
for (kn, c), (v, n) in sorted(acc.items()):
    print('%-62s %-34s %14.0f per launch (%d)' % (kn, c, v / n, n))
PY
done
