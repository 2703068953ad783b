#!/bin/bash
# tools/ab_r4q.sh -- round 4, session q (one box): the incremental packer (ocr_pack2_k + fd_permute_rows) against the legacy one:
# parity tests, cost at C2 size (tools/setup_profile.py with the packer at construction), kernel after either
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_records.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -6 > gpurun_out/r4q_tests.txt; tail -3 gpurun_out/r4q_tests.txt
for lg in 0 1; do
  echo "== FDHIP_OCR_PACK_AFTER=0 FDHIP_PACK_LEGACY=$lg: first Jacobian call"
  FDHIP_OCR_PACK_AFTER=0 FDHIP_PACK_LEGACY=$lg python tools/setup_profile.py 2>&1 | awk '/== jacobian first call/,/C-ABI total/' | head -12
done 2>&1 | tee gpurun_out/r4q_pack_cost.txt
run() {  # numbering, env...
  nb=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic off 2>gpurun_out/r4q_last.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  first_call_s %.3f' % (r['ms'], r['assemble_ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
}
{
for rep in 1 2; do
for nb in lexicographic tiled; do
for cfg in "FDHIP_OCR_PACK_AFTER=64" "FDHIP_OCR_PACK_AFTER=0" "FDHIP_OCR_PACK_AFTER=0 FDHIP_PACK_LEGACY=1"; do
  echo "== $nb $cfg"; run $nb $cfg
done; done; done
} 2>&1 | tee gpurun_out/r4q_ab_packer.txt
