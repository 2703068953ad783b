#!/bin/bash
# tools/ab_sweep_c5.sh -- CG2 (config C5, one GPU's share): tiled vs sweep numbering (chained owner-computes-rows, window 3)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_sweep_c5.txt
: > $OUT
run() {
  echo "=== numbering=$1 tile=$2 env: $3" >> $OUT
  env $3 FDHIP_DEBUG=1 python bench.py --workload c5 --numbering $1 --variants '' --traffic off --cpu-sample 0 --steps 6 --warmup 2 --tile $2 2> gpurun_out/ab_sweep_c5.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step_ms', round(d['ms_per_step'], 4), 'res_ms', round(d['roofline_residual']['ms'], 4), 'jac_ms', round(d['roofline_jacobian']['ms'], 4), 'setup', d['setup_s'])" >> $OUT
  grep "OCR" gpurun_out/ab_sweep_c5.err | tail -1 >> $OUT
  grep "epb=" gpurun_out/ab_sweep_c5.err | tail -1 >> $OUT
  tail -3 gpurun_out/ab_sweep_c5.err | grep -i "error\|Traceback" >> $OUT
}
run tiled 8,8,4 "A=1"
run sweep 13,4,4 "A=1"
run sweep 13,8,4 "A=1"
run sweep 13,8,4 "FDHIP_LDS_LIMIT=159000"
run sweep 13,8,8 "FDHIP_LDS_LIMIT=159000"
cat $OUT
