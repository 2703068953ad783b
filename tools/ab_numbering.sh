#!/bin/bash
# tools/ab_numbering.sh -- un-hinted numberings: backend-derived locality (FDHIP_LOCALITY_ORDER=1) vs the caller's order cut
# into uniform blocks (=0).  Output: gpurun_out/ab_numbering.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_numbering.txt
: > $OUT
run() {
  echo "=== numbering=$1 env: $2" >> $OUT
  env $2 FDHIP_DEBUG=1 python bench.py --numbering $1 --variants '' --traffic off --cpu-sample 0 --no-secondary --steps 6 --warmup 2 2> gpurun_out/ab_numbering.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step_ms', round(d['ms_per_step'], 4), 'res_ms', round(d['roofline_residual']['ms'], 4), 'jac_ms', round(d['roofline_jacobian']['ms'], 4), 'setup', {k: round(v, 2) for k, v in d['setup_s'].items()})" >> $OUT
  grep "OCR" gpurun_out/ab_numbering.err | tail -1 >> $OUT
  grep "epb=" gpurun_out/ab_numbering.err | tail -1 >> $OUT
  grep -i "error\|Traceback" gpurun_out/ab_numbering.err | tail -3 >> $OUT
}
run lexicographic "FDHIP_LOCALITY_ORDER=1"
run random "FDHIP_LOCALITY_ORDER=1"
run lexicographic "FDHIP_LOCALITY_ORDER=0"
run random "FDHIP_LOCALITY_ORDER=0"
cat $OUT
