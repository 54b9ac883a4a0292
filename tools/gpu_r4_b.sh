#!/bin/bash
# round 4, call B: pool workgroups evaluate the predicates (BXI_POOL_PRED=1) vs predicate workgroups of their own (=0)
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu --maxfail=10 -p no:cacheprovider > gpurun_out/r4b_pytest.log 2>&1
tail -15 gpurun_out/r4b_pytest.log
bash tools/ab.sh "BXI_POOL_PRED=0" "BXI_POOL_PRED=1" "BXI_POOL_PRED=1,BXI_ONE_MERGE=0"
