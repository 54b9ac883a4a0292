#!/bin/bash
mkdir -p gpurun_out
for cfg in "$@"; do
  echo "=== $cfg"
  env ${cfg//,/ } python tools/trace_eval.py 2>&1 | grep -v amdgpu.ids | cut -c1-1900
done
