#!/bin/bash
# tools/ab_c5.sh VAR "v1 v2 ..." -- same-box A/B of one environment switch on the CG2 share (bench.py --workload c5 --n 107), un-hinted numbering
VAR=$1; VALS=$2; shift 2
for v in $VALS $VALS; do
echo "== c5 n=107 numbering=lexicographic $VAR=$v"
env $VAR=$v python bench.py --workload c5 --n 107 --steps 20 --warmup 3 --cpu-sample 0 "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; q=d['roofline_residual']
print('  step_ms %.4f  jacobian_ms %.4f frac %.4f  residual_ms %.4f frac %.4f  first_call_s %.3f' % (d['ms_per_step'], r['ms'], r['frac'], q['ms'], q['frac'], d['setup_s']['plans_jacobian_first_call']))"
done
