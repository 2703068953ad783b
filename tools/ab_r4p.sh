#!/bin/bash
# tools/ab_r4p.sh -- round 4, session p (one box): 16-lane LDS atomic windows of the P1 Jacobian kept on distinct banks by filling
# the rest of a window with instances from the TAIL of the block's list (FDHIP_OCR_PAD_WINDOWS=2: a permutation, no extra slots)
# against the plain stencil order and against dummies (=1); with and without the bank-aware packer on top
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_records.py -x -q -m gpu -k "window_padding or deferred" 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -6 > gpurun_out/r4p_tests.txt; tail -3 gpurun_out/r4p_tests.txt
run() {  # numbering, env...
  nb=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic off 2>gpurun_out/r4p_last.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  first_call_s %.3f' % (r['ms'], r['assemble_ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
}
{
for rep in 1 2; do
for nb in lexicographic tiled; do
for cfg in "FDHIP_OCR_PAD_WINDOWS=0" "FDHIP_OCR_PAD_WINDOWS=2" "FDHIP_OCR_PAD_WINDOWS=0 FDHIP_OCR_PACK_AFTER=1" "FDHIP_OCR_PAD_WINDOWS=2 FDHIP_OCR_PACK_AFTER=1"; do
  echo "== $nb $cfg"; run $nb $cfg
done; done; done
} 2>&1 | tee gpurun_out/r4p_ab_fill_windows.txt
