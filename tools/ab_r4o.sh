#!/bin/bash
# tools/ab_r4o.sh -- round 4, session o (one box): stencil order of the whole-entity owner-computes-rows plan grouped by ownership
# pattern ("stencil": skips the atomics of rows no lane of a wavefront owns) against grouped by shape ("shape": 16 consecutive
# instances on 16 consecutive rows, conflict-free LDS atomic windows), P1 Jacobian at C2 size, un-hinted and hinted numbering
mkdir -p gpurun_out
run() {  # numbering, env...
  nb=$1; shift
  env "$@" FDHIP_DEBUG=1 python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic off 2>gpurun_out/r4o_last.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  first_call_s %.3f' % (r['ms'], r['assemble_ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
  grep "OCR \\[" gpurun_out/r4o_last.err | head -1 | sed 's/^/  /'
}
{
for rep in 1 2; do
for nb in lexicographic tiled; do
for cfg in "FDHIP_OCR_ORDER=stencil" "FDHIP_OCR_ORDER=shape" "FDHIP_OCR_ORDER=shape FDHIP_OCR_PAD_WINDOWS=1" "FDHIP_OCR_ORDER=shape FDHIP_OCR_PACK_AFTER=1" "FDHIP_OCR_ORDER=stencil FDHIP_OCR_PACK_AFTER=1"; do
  echo "== $nb $cfg"; run $nb $cfg
done; done; done
} 2>&1 | tee gpurun_out/r4o_ab_shape_order.txt
