#!/bin/bash
# Developer helper (GPU box): everything profiles/ is built from.  Usage: tools/gpu_profiles.sh r03 [eval-only]
TAG=${1:-r02}
ONLY=${2:-all}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing --no-extras"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_eager -o $TAG -- $CMD > $R/gpurun_out/prof_eager.log 2>&1
cut -d, -f1-4 $R/gpurun_out/prof_eager/${TAG}_kernel_stats.csv | head -6
# the two-launch form of the same evaluation (what larger instance counts, dilation 4 and the head-fused call run)
BXI_ONE_LAUNCH=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_two_launch -o $TAG -- $CMD > $R/gpurun_out/prof_two_launch.log 2>&1
cut -d, -f1-4 $R/gpurun_out/prof_two_launch/${TAG}_kernel_stats.csv | head -4
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -o $TAG -- \
     python $R/bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-extras > $R/gpurun_out/pmc_$c.log 2>&1
  ls $R/gpurun_out/pmc_$c | head -3
done
for t in pairwise_op dynamic_head head_fused discobox levelset tree_filter; do
  [ "$ONLY" = eval-only ] && break
  # kernel durations under the profiler; the wall-clock JSON from a run WITHOUT it (a row of ~200 tiny launches is twice as slow under rocprofv3)
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$t -o $TAG -- python $R/tools/bench_$t.py > /dev/null 2> $R/gpurun_out/${t}_bench.err
  timeout 300 python $R/tools/bench_$t.py > $R/gpurun_out/${t}_bench.json 2>> $R/gpurun_out/${t}_bench.err
  tail -c 300 $R/gpurun_out/${t}_bench.json | tr '\n' ' '; echo
done
cd $R
# the per-wave trace needs the -DBXI_TRACE build of the library (tools/trace_eval.py's docstring); it is not kept in the tree
if [ -f boxinstseg_amd/lib/libboxinst_hip_trace.so ]; then
  python tools/trace_eval.py 2>&1 | grep -v amdgpu.ids > gpurun_out/block_trace.txt; tail -3 gpurun_out/block_trace.txt | cut -c1-300
else
  rm -f gpurun_out/block_trace.txt; echo "no trace build: block trace skipped"
fi
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
