#!/bin/bash
# Developer helper (GPU box): everything profiles/ is built from.  Usage: tools/gpu_profiles.sh r05 [eval-only]
TAG=${1:-r05}
ONLY=${2:-all}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 400 --warmup 50 --no-cpu-baseline --no-kernel-timing --no-extras"
prof() {   # prof <dir> <command...>: rocprofv3 kernel stats of one command -> gpurun_out/<dir>/<TAG>_kernel_stats.csv
  local d=$1; shift
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/$d -o $TAG -- "$@" > $R/gpurun_out/$d.log 2>&1
  cut -d, -f1-4 $R/gpurun_out/$d/${TAG}_kernel_stats.csv | head -5
}
prof prof_eager $CMD
# the two-launch form of the same evaluation (BXI_EVAL_TWO_LAUNCHES = 2: what larger instance counts, dilation 3 / 4
# and the head-fused call run); the shape real training runs, 128 and 64 instances (default form);
# the targets-ahead pair (bxi_boxinst_targets_f32 + BXI_EVAL_TARGETS_READY) at 32 and 128 instances
prof prof_two_launch $CMD --flags 2
prof prof_n128 $CMD --inst-per-box 4 --sets 6
prof prof_n64 $CMD --inst-per-box 2 --sets 6
prof prof_targets_n32 python $R/tools/ab_forms.py --ipb 1 --forms ready,targets_only --reps 1 --steps 300 --sets 6
prof prof_targets_n128 python $R/tools/ab_forms.py --ipb 4 --forms ready,targets_only --reps 1 --steps 300 --sets 6
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
# op-level pairwise_nlog backward: PMC passes of the stand-alone harness (round-5 kernel, pair kernel, linear copy side by side) + its timings
if [ -x tools/micro/pw_bwd_nt ]; then
  (echo "# tools/micro/pw_bwd.hip (wide = the round-5 backward, pair = pairwise3_bwd_pair_kernel, copy / geo = copies of the same bytes), then rocprofv3 --pmc per launch (tools/micro/pmc_pw.sh)"; tools/micro/pw_bwd_nt; tools/micro/pmc_pw.sh tools/micro/pw_bwd_nt 2>&1 | grep -v fillBuffer) > gpurun_out/pairwise_op_pmc.txt 2>&1
  tail -3 gpurun_out/pairwise_op_pmc.txt | cut -c1-200
fi
# the per-wave trace needs the -DBXI_TRACE build of the library (tools/trace_forms.py's docstring); it is not kept in the tree
if [ -f boxinstseg_amd/lib/libboxinst_hip_trace.so ]; then
  (python tools/trace_forms.py; IPB=4 python tools/trace_forms.py; IPB=4 BXI_FLAGS=34 python tools/trace_forms.py) 2>&1 | grep -v amdgpu.ids > gpurun_out/block_trace.txt; tail -3 gpurun_out/block_trace.txt | cut -c1-300
else
  rm -f gpurun_out/block_trace.txt; echo "no trace build: block trace skipped"
fi
python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -c 600 gpurun_out/bench_default.json
