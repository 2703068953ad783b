#!/bin/bash
# tools/ab_r4w.sh -- round 4, last GPU minute: row blocks twice as large on the un-hinted P1 Jacobian (512 rows: fewer redundant
# instances, two workgroups per CU instead of three)
mkdir -p gpurun_out
run() {
  env "$@" FDHIP_DEBUG=1 python bench.py --steps 12 --warmup 3 --only jacobian --numbering lexicographic --variants "" --no-secondary --cpu-sample 0 --traffic off 2>gpurun_out/r4w_last.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  frac %.4f' % (r['ms'], r['frac']))"
  grep "OCR \[" gpurun_out/r4w_last.err | head -1 | sed 's/^/  /' | cut -c1-230
}
{
for cfg in "X=0" "FDHIP_OCR_NNZ_ORDERED=7680 FDHIP_OCR_LDS_LIMIT=98304" "FDHIP_OCR_NNZ_ORDERED=7680 FDHIP_OCR_LDS_LIMIT=98304 FDHIP_OCR_BLOCK_THREADS=1024"; do
  echo "== $cfg"; run $cfg
done
} 2>&1 | tee gpurun_out/r4w_ab_large_blocks.txt
