#!/bin/bash
# tools/ab_numbering2.sh -- block sizes of the backend-derived orders (lexicographic numbering, no hints)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_numbering2.txt
: > $OUT
run() {
  echo "=== numbering=$1 env: $2" >> $OUT
  env $2 FDHIP_DEBUG=1 python bench.py --numbering $1 --variants '' --traffic off --cpu-sample 0 --no-secondary --steps 6 --warmup 2 2> gpurun_out/ab_numbering2.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step_ms', round(d['ms_per_step'], 4), 'res_ms', round(d['roofline_residual']['ms'], 4), 'jac_ms', round(d['roofline_jacobian']['ms'], 4))" >> $OUT
  grep "OCR" gpurun_out/ab_numbering2.err | tail -1 >> $OUT
  grep "epb=" gpurun_out/ab_numbering2.err | tail -1 >> $OUT
}
run lexicographic "FDHIP_ENTS_PER_BLOCK=1536"
run lexicographic "FDHIP_ENTS_PER_BLOCK=2048"
run lexicographic "FDHIP_ENTS_PER_BLOCK=3072"
run lexicographic "FDHIP_ENTS_PER_BLOCK=2048 FDHIP_OCR_NNZ_ORDERED=3300"
cat $OUT
