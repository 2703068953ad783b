#!/bin/bash
# tools/ab_r4n.sh -- round 4, session n (one box):
#  (1) conflict-free LDS atomic windows of the P1 Jacobian (FDHIP_OCR_PAD_WINDOWS: dummy instances, fd_ocrplan_pad_windows), un-hinted and hinted
#  (2) C3 matrix: single-cell rows stored + shared rows zeroed (coalesced kernel) against the plain fill + atomics everywhere
#  (3) Q4 action at n = 64: half tables / five cubes / register caps
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_records.py tests/test_gpu_q4_hex.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -12 > gpurun_out/r4n_tests.txt; tail -3 gpurun_out/r4n_tests.txt
run() {  # numbering, env...
  nb=$1; shift
  env "$@" FDHIP_DEBUG=1 python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic off 2>gpurun_out/r4n_last.err | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  first_call_s %.3f' % (r['ms'], r['assemble_ms'], r['frac'], d['setup_s']['plans_jacobian_first_call']))"
  grep "OCR \[" gpurun_out/r4n_last.err | head -1 | sed 's/^/  /'
}
{
for rep in 1 2; do
for cfg in "FDHIP_OCR_PAD_WINDOWS=0" "FDHIP_OCR_PAD_WINDOWS=1"; do
  echo "== lexicographic $cfg"; run lexicographic $cfg
done
for cfg in "FDHIP_OCR_PAD_WINDOWS=0" "FDHIP_OCR_PAD_WINDOWS=1"; do
  echo "== tiled $cfg"; run tiled $cfg
done
done
} 2>&1 | tee gpurun_out/r4n_ab_pad_windows.txt
{
for rep in 1 2; do
for v in 1 0; do
  echo "== FDHIP_TP_STORE_SINGLE_ROWS=$v rep $rep"
  FDHIP_TP_STORE_SINGLE_ROWS=$v python bench.py --workload c3 --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; a=d.get('roofline_action',{})
print('  jacobian kernel_ms %.4f frac %.4f assemble_ms %.4f frac_assemble %.4f | action ms %.4f frac_valu %.4f' % (r['ms'], r['frac'], r.get('assemble_ms',0), r.get('frac_assemble',0), a.get('ms',0), a.get('frac_valu',0)))"
done; done
} 2>&1 | tee gpurun_out/r4n_c3_single_rows.txt
{
for rep in 1 2; do
for cfg in "X=0" "FDHIP_CFLAGS=-DFDT_ACTION_FULL_TABLES" "FDHIP_CFLAGS=-DFDT_ACTION_SIX_CUBES" "FDHIP_CFLAGS=-DFDT_ACTION_FULL_TABLES -DFDT_ACTION_SIX_CUBES" \
           "FDHIP_TP_ACTION_WAVES=3" "FDHIP_TP_ACTION_WAVES=4"; do
  echo "== action n=64 $cfg"
  env "$cfg" python -c "
import bench, json
r = bench.measure_c3_action(64, 10, 3)
print('  kernel_ms %.4f frac_valu %.4f' % (r['kernel_ms'], r['frac_valu']))" 2>/dev/null
done; done
} 2>&1 | tee gpurun_out/r4n_action_variants.txt
