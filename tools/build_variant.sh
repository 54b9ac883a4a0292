#!/bin/bash
# usage: build_variant.sh <source.hip> <out.so> [-D...]: the library with ONE source recompiled under extra flags (the other objects are cached
# in /tmp/bxi_obj): A/B builds in seconds instead of a minute.  Developer tool; the shipped library is built by boxinstseg_amd/build.py.
set -e
src=$1; out=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
C=$ROOT/boxinstseg_amd/csrc; O=/tmp/bxi_obj; mkdir -p $O
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-kernarg-preload-count=16"
pids=()
for f in abi pairwise_op color_affinity mask_loss fused_eval dynamic_head dynamic_head_generic meanfield levelset tree_filter tree_filter_large sol_eval; do
  if [ "$f.hip" != "$src" ] && { [ ! -f $O/$f.o ] || [ $C/$f.hip -nt $O/$f.o ] || [ -n "$(find $C $ROOT/include -name '*.h*' -newer $O/$f.o | grep -v '\.hip$')" ]; }; then
    hipcc $FLAGS -c $C/$f.hip -o $O/$f.o & pids+=($!)
  fi
done
tag=$(echo "$src $@" | md5sum | cut -c1-8)
hipcc $FLAGS "$@" -c $C/$src -o $O/var_$tag.o
for p in "${pids[@]}"; do wait $p; done
objs=""
for f in abi pairwise_op color_affinity mask_loss fused_eval dynamic_head dynamic_head_generic meanfield levelset tree_filter tree_filter_large sol_eval; do
  if [ "$f.hip" == "$src" ]; then objs="$objs $O/var_$tag.o"; else objs="$objs $O/$f.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $out $objs
echo built $out
