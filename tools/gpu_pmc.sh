#!/bin/bash
# Developer helper (GPU box): HBM traffic counters for the bench kernels, one --pmc pass per counter
# (TCC slots: FETCH_SIZE 3 + WRITE_SIZE 2 do not fit one pass; see MI355X_MICROARCH.md).
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$c -o r01 -- \
     python $GRAFT_REPO_ROOT/bench.py --mode eager --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-pipelined > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.log 2>&1
  ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$c | head -5
done
