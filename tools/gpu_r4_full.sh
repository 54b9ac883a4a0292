#!/bin/bash
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -q -m gpu --maxfail=20 -p no:cacheprovider > gpurun_out/r4_pytest_full.log 2>&1
tail -15 gpurun_out/r4_pytest_full.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
