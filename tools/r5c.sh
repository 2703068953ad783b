#!/bin/bash
# tools/r5c.sh -- round 5, GPU session c (one box): block geometry of the row-sliced CG2 Jacobian after the flush / staging changes
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
run() {
  env "$@" python bench.py --workload c5 --n 107 --steps 20 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; q=d['roofline_residual']
print('  step_ms %.4f  jacobian_ms %.4f frac %.4f  residual_ms %.4f frac %.4f' % (d['ms_per_step'], r['ms'], r['frac'], q['ms'], q['frac']))"
}
{
for rep in 1 2; do
for cfg in "FDHIP_OCRS_NNZ=4096" "FDHIP_OCRS_NNZ=3400" "FDHIP_OCRS_NNZ=3072" "FDHIP_OCRS_NNZ=8192 FDHIP_OCRS_BLOCK_THREADS=512 FDHIP_LDS_LIMIT=90112" \
           "FDHIP_OCRS_NNZ=6144 FDHIP_OCRS_BLOCK_THREADS=512 FDHIP_LDS_LIMIT=90112" "FDHIP_OCRS_NNZ=4096 FDHIP_OCRS_BLOCK_THREADS=512"; do
  echo "== c5 n=107 lexicographic $cfg"; timeout 200 bash -c "$(declare -f run); run $cfg"
done; done
} > gpurun_out/r5c_sweep_c5_blocks.txt 2>&1
cat gpurun_out/r5c_sweep_c5_blocks.txt
