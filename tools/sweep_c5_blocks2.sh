#!/bin/bash
# tools/sweep_c5_blocks2.sh -- 4096-entry / 256-lane against 8192-entry / 512-lane row blocks of the paired row-sliced P2 Jacobian on the
# hinted (tiled) and the un-hinted (lexicographic) numbering of the CG2 share; only the Jacobian time is read (the LDS limit is global)
run() {
nb=$1; shift
echo "== c5 n=107 $nb $*"
env "$@" python bench.py --workload c5 --n 107 --numbering $nb --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  jacobian_ms %.4f frac %.4f' % (r['ms'], r['frac']))"
}
for rep in 1 2; do for nb in tiled lexicographic random; do
run $nb FDHIP_OCRS_NNZ=4096
run $nb FDHIP_OCRS_NNZ=8192 FDHIP_OCRS_BLOCK_THREADS=512 FDHIP_LDS_LIMIT=90112
done; done
