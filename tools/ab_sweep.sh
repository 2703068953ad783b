#!/bin/bash
# tools/ab_sweep.sh -- sliding-window owner-computes-rows on the sweep numbering: tile shapes (chain length, plane size)
# and workgroup sizes.  Output: gpurun_out/ab_sweep.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_sweep.txt
: > $OUT
run() {
  echo "=== numbering=$1 tile=$2 env: $3" >> $OUT
  env $3 FDHIP_DEBUG=1 python bench.py --numbering $1 --variants '' --traffic off --cpu-sample 0 --steps 8 --warmup 2 --tile $2 2> gpurun_out/ab_sweep.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step_ms', round(d['ms_per_step'], 4), 'res_ms', round(d['roofline_residual']['ms'], 4), 'jac_ms', round(d['roofline_jacobian']['ms'], 4), 'jac_frac', round(d['roofline_jacobian']['frac'], 4), 'setup', d['setup_s'])" >> $OUT
  grep "OCR" gpurun_out/ab_sweep.err | tail -1 >> $OUT
  grep "epb=" gpurun_out/ab_sweep.err | tail -1 >> $OUT
}
run tiled 8,8,4 "A=1"
run sweep 27,16,8 "A=1"
run sweep 27,16,8 "FDHIP_OCR_BLOCK_THREADS=256"
run sweep 27,8,8 "A=1"
run sweep 27,8,8 "FDHIP_OCR_BLOCK_THREADS=256"
run sweep 27,16,16 "FDHIP_LDS_LIMIT=100000"
run sweep 27,16,16 "FDHIP_LDS_LIMIT=100000 FDHIP_OCR_BLOCK_THREADS=256"
run sweep 54,16,8 "A=1"
run sweep 14,16,8 "A=1"
cat $OUT
