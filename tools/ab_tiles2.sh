#!/bin/bash
# tools/ab_tiles2.sh -- owner-computes-rows Jacobian: smaller row tiles (more resident workgroups per CU against more
# redundant instances) and workgroup sizes.  Output: gpurun_out/ab_tiles2.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_tiles2.txt
: > $OUT
run() {
  echo "=== tile=$1 env: $2" >> $OUT
  env $2 FDHIP_DEBUG=1 python bench.py --only jacobian --variants '' --traffic off --cpu-sample 0 --steps 8 --warmup 2 --tile $1 2> gpurun_out/ab_tiles2.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('jac_kernel_ms', round(d['roofline_jacobian']['ms'], 4), 'frac', round(d['roofline_jacobian']['frac'], 4))" >> $OUT
  grep "OCR" gpurun_out/ab_tiles2.err | tail -1 >> $OUT
}
run 8,8,4 "A=1"
run 8,8,3 "A=1"
run 8,6,4 "A=1"
run 6,6,4 "A=1"
run 8,8,2 "A=1"
run 8,8,4 "FDHIP_OCR_BLOCK_THREADS=1024"
run 8,8,4 "FDHIP_OCR_BLOCK_THREADS=256"
run 8,8,4 "FDHIP_PREFETCH=0"
run 8,8,4 "FDHIP_OCR_PACK=0"
cat $OUT
