#!/bin/bash
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_golden.py -q -m gpu --maxfail=10 -p no:cacheprovider > gpurun_out/r4c_pytest.log 2>&1
tail -12 gpurun_out/r4c_pytest.log
bash tools/ab.sh "BXI_RESIDENT=0" "BXI_RESIDENT=1"
