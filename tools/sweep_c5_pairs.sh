#!/bin/bash
# tools/sweep_c5_pairs.sh -- block geometry of the paired row-sliced P2 Jacobian (CG2 share, un-hinted numbering), same box
run() {
echo "== c5 n=107 lexicographic $*"
env "$@" python bench.py --workload c5 --n 107 --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; q=d['roofline_residual']
print('  step_ms %.4f  jacobian_ms %.4f frac %.4f  residual_ms %.4f frac %.4f' % (d['ms_per_step'], r['ms'], r['frac'], q['ms'], q['frac']))"
}
for rep in 1 2; do
run FDHIP_OCRS_PAIRS=1 FDHIP_OCRS_NNZ=4096
run FDHIP_OCRS_PAIRS=1 FDHIP_OCRS_NNZ=5200
run FDHIP_OCRS_PAIRS=1 FDHIP_OCRS_NNZ=3200
run FDHIP_OCRS_PAIRS=1 FDHIP_OCRS_NNZ=4096 FDHIP_OCRS_BLOCK_THREADS=192
run FDHIP_OCRS_PAIRS=1 FDHIP_OCRS_NNZ=4096 FDHIP_OCRS_BLOCK_THREADS=128
run FDHIP_OCRS_PAIRS=1 FDHIP_OCRS_NNZ=8192 FDHIP_OCRS_BLOCK_THREADS=512 FDHIP_LDS_LIMIT=90112
run FDHIP_OCRS_PAIRS=0 FDHIP_OCRS_NNZ=4096
done
