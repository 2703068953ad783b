#!/bin/bash
# tools/ab_c5_paths.sh -- the CG2 share (bench.py --workload c5 --n 107) on the hinted numbering with and without its hints and on the
# un-hinted numbering: separates the numbering (memory locality) from the code path (derived row order, row-by-row flush)
run() { echo "== $*"; env "$@" timeout 40 python bench.py --workload c5 --n 107 --steps 15 --warmup 3 --cpu-sample 0 --numbering $NB 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; q=d['roofline_residual']
print('  step_ms %.4f  jacobian_ms %.4f  residual_ms %.4f' % (d['ms_per_step'], r['ms'], q['ms']))"; }
NB=tiled run FDHIP_PREFERRED_BLOCKS=1
NB=tiled run FDHIP_PREFERRED_BLOCKS=0
NB=lexicographic run FDHIP_PREFERRED_BLOCKS=1
