#!/bin/bash
# tools/ab_sliced.sh -- row-sliced owner-computes-rows (FDHIP_OCR_SLICED=1) vs whole-entity instances (=0) on the P2
# workload (bench.py --workload c5), block sizes and numberings.  Output: gpurun_out/ab_sliced.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
OUT=gpurun_out/ab_sliced.txt
: > $OUT
run() {
  echo "=== workload=$1 numbering=$2 env: $3" >> $OUT
  env $3 FDHIP_DEBUG=1 timeout 600 python bench.py --workload $1 --numbering $2 --variants '' --traffic off --cpu-sample 0 --no-secondary --steps 6 --warmup 2 2> gpurun_out/ab_sliced.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('step_ms', round(d['ms_per_step'], 4), 'res_ms', round(d['roofline_residual']['ms'], 4), 'jac_ms', round(d['roofline_jacobian']['ms'], 4), 'jac_frac', round(d['roofline_jacobian']['frac'], 4), 'setup', {k: round(v, 2) for k, v in d['setup_s'].items()})" >> $OUT
  grep "OCR" gpurun_out/ab_sliced.err | tail -1 >> $OUT
  grep -i "error\|Traceback" gpurun_out/ab_sliced.err | tail -3 >> $OUT
}
# (the runs behind profiles/r2t_ab_sliced_ocr.txt; comment in what is needed)
run c5 tiled "FDHIP_OCR_SLICED=1"
run c5 tiled "FDHIP_OCR_SLICED=0"
run c5 lexicographic "FDHIP_OCR_SLICED=1"
run c5 lexicographic "FDHIP_OCR_SLICED=0"
run c5 random "FDHIP_OCR_SLICED=1"
# run c5 tiled "FDHIP_PREFETCH=0"
# run c5 tiled "FDHIP_OCRS_INTERLEAVE=0"
# run c5 tiled "FDHIP_OCRS_NNZ=2048"
# run c5 tiled "FDHIP_OCRS_NNZ=6144"
# run c5 tiled "FDHIP_OCRS_BLOCK_THREADS=512"
# run c5 tiled "FDHIP_OCRS_BALANCED_CUTS=1"
# run c2 tiled "FDHIP_OCR_SLICED_MIN_ARITY=4"
cat $OUT
