#!/bin/bash
# Developer helper (GPU box): SQ / cache counters of the op-level pairwise kernels (one rocprofv3 --pmc pass per counter set)
R=$GRAFT_REPO_ROOT

bash $R/tools/gpu_pmc.sh pairwise3 "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM" \
   "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE" \
   "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" -- python $R/tools/bench_pairwise_op.py 2>&1 | grep -v "double" | tee $R/gpurun_out/pw_pmc.txt
