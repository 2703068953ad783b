#!/bin/bash
# tools/ab_env.sh VAR "v1 v2 ..." [bench args] -- A/B of one environment switch on the C2 Jacobian, both numberings, same box
VAR=$1; VALS=$2; shift 2
for nb in tiled lexicographic; do for v in $VALS $VALS; do
echo "== numbering=$nb $VAR=$v"
env $VAR=$v python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic off "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  first_call_s %.3f' % (r['ms'], r['assemble_ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
done; done
