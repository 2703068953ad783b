#!/bin/bash
# tools/ab_tiles.sh -- sensitivity of the owner-computes-rows Jacobian to the row-tile shape (instance redundancy vs LDS
# footprint / resident workgroups).  Output: gpurun_out/ab_tiles.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_tiles.txt
: > $OUT
run() {
  echo "=== tile=$1 env: $2" >> $OUT
  env $2 FDHIP_DEBUG=1 python bench.py --only jacobian --variants '' --traffic off --cpu-sample 0 --steps 8 --warmup 2 --tile $1 2> gpurun_out/ab_tiles.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('jac_kernel_ms', d['roofline_jacobian']['ms'], 'frac', d['roofline_jacobian']['frac'])" >> $OUT
  grep "OCR" gpurun_out/ab_tiles.err | tail -1 >> $OUT
}
run 8,8,4 "A=1"
run 8,8,8 "FDHIP_LDS_LIMIT=110000"
run 16,8,4 "FDHIP_LDS_LIMIT=110000"
run 8,8,8 "FDHIP_LDS_LIMIT=110000 FDHIP_OCR_BLOCK_THREADS=1024"
run 16,16,4 "FDHIP_LDS_LIMIT=163000 FDHIP_OCR_BLOCK_THREADS=1024"
run 4,8,8 "A=1"
run 8,4,4 "A=1"
cat $OUT
