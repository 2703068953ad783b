#!/bin/bash
# tools/sweep_c5_r4.sh -- block size x workgroup size of the row-sliced CG2 Jacobian (un-hinted share, n = 107) on the round-4 wrapper
run() { echo "== $*"; env "$@" python bench.py --workload c5 --n 107 --steps 15 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  jacobian_ms %.4f frac %.4f' % (r['ms'], r['frac']))"; }
for rep in 1 2; do
run FDHIP_OCRS_NNZ=4096 FDHIP_OCRS_BLOCK_THREADS=256
run FDHIP_OCRS_NNZ=8192 FDHIP_OCRS_BLOCK_THREADS=512 FDHIP_LDS_LIMIT=90112
run FDHIP_OCRS_NNZ=6144 FDHIP_OCRS_BLOCK_THREADS=512 FDHIP_LDS_LIMIT=90112
run FDHIP_OCRS_NNZ=8192 FDHIP_OCRS_BLOCK_THREADS=256 FDHIP_LDS_LIMIT=90112
run FDHIP_OCRS_NNZ=3072 FDHIP_OCRS_BLOCK_THREADS=256
run FDHIP_OCRS_NNZ=6144 FDHIP_OCRS_BLOCK_THREADS=384 FDHIP_LDS_LIMIT=90112
done
