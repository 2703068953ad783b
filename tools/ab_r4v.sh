#!/bin/bash
# tools/ab_r4v.sh -- round 4, session v (one box): is the fp64 LDS atomic rate what binds the P1 Jacobian?  The same kernel with
# 64-bit FIXED-POINT accumulators (integer LDS atomics, FDHIP_OCR_FIXED_POINT=50: |A| 2^50 < 2^47 at C2 size) against the default
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_records.py -x -q -m gpu -k "fixed_point" 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -3
python - <<'PY' 2>&1 | grep -v Warning | tee gpurun_out/r4v_fixed_point_parity.txt
import numpy as np
from firedrake_amd import forms, mesh as fmesh
from firedrake_amd.configuration import configuration
m = fmesh.UnitCubeMesh(215, degrees=(1,), perturb=0.1, numbering="lexicographic")
vals = {}
for bits in (0, 50):
    configuration["ocr_fixed_point"] = bits
    prob = forms.PoissonProblem(m, 1, bcs=True)
    vals[bits] = prob.assemble_jacobian().csr()[2]
    del prob
vmax = np.abs(vals[0]).max()
print("C2 size: max|A| %.4g, max |fixed - fp64| / max|A| = %.3g" % (vmax, np.abs(vals[50] - vals[0]).max() / vmax))
PY
run() {  # numbering, env...
  nb=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f' % (r['ms'], r['assemble_ms'], r['frac']))"
}
{
for rep in 1 2; do
for nb in lexicographic tiled; do
for cfg in "FDHIP_OCR_FIXED_POINT=0" "FDHIP_OCR_FIXED_POINT=50"; do
  echo "== $nb $cfg"; run $nb $cfg
done; done; done
} 2>&1 | tee gpurun_out/r4v_ab_fixed_point.txt
