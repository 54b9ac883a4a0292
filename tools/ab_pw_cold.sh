#!/bin/bash
# usage: ab_pw_cold.sh lib1.so ...: the op-level bench (cold rotating sets) per library build, twice, interleaved
for rep in 1 2; do for lib in "$@"; do
  python tools/with_lib.py $lib tools/bench_pairwise_op.py 2>/dev/null | python -c "
import json,sys; r=json.loads(sys.stdin.read())['torch.float32']; print('%-52s fwd %.2f bwd %.2f (warm %.2f / %.2f; sol %.2f / %.2f)' % ('$lib'[-52:], r['fwd_us'], r['bwd_us'], r['fwd_warm_us'], r['bwd_warm_us'], r['fwd_sol_us'], r['bwd_sol_us']))"
done; done
