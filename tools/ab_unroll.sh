#!/bin/bash
# tools/ab_unroll.sh -- P2 residual (27-point mass quadrature loop) with LLVM's default unroll threshold vs a raised one.
# Output: gpurun_out/ab_unroll.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_unroll.txt
: > $OUT
run() {
  echo "=== FDHIP_CFLAGS='$1'" >> $OUT
  FDHIP_CFLAGS="$1" timeout 100 python bench.py --workload c5 --only residual --variants '' --traffic off --cpu-sample 0 --no-secondary --steps 6 --warmup 2 2> gpurun_out/ab_unroll.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('res_ms', round(d['roofline_residual']['ms'], 4), 'frac', round(d['roofline_residual']['frac'], 4))" >> $OUT
  grep -i "error\|Traceback" gpurun_out/ab_unroll.err | tail -2 >> $OUT
}
run "-mllvm -unroll-threshold=4000"
run ""
cat $OUT
