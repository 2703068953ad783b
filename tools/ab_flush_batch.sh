#!/bin/bash
# tools/ab_flush_batch.sh -- FDHIP_FLUSH_BATCH 1 | 4 on the un-hinted CG2 share (row-sliced, row-by-row flush) and on the un-hinted C2 Jacobian at
# n = 128 (whole-entity, per-entry flush table), same box
for v in 1 4; do
echo "== c5 n=107 lexicographic FDHIP_FLUSH_BATCH=$v"
FDHIP_FLUSH_BATCH=$v timeout 40 python bench.py --workload c5 --n 107 --steps 15 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  step_ms %.4f  jacobian_ms %.4f  first_call_s %.3f' % (d['ms_per_step'], r['ms'], d['setup_s']['plans_jacobian_first_call']))"
done
for v in 1 4; do
echo "== c2 n=128 lexicographic FDHIP_FLUSH_BATCH=$v"
FDHIP_FLUSH_BATCH=$v timeout 40 python bench.py --n 128 --steps 20 --warmup 3 --only jacobian --numbering lexicographic --variants "" --no-secondary --cpu-sample 0 --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  jacobian_ms %.4f  frac %.4f  first_call_s %.3f' % (r['ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
done
