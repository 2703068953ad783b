#!/bin/bash
# tools/r5a.sh -- round 5, first GPU session (one box): parity of the checked fixed-point path, then same-box A/Bs of
#   FDHIP_OCR_FIXED_POINT (fp64 / checked fixed-point LDS accumulators), FDHIP_EARLY_LOADS (index rows requested ahead of the staging
#   phase) and the packing policy on the un-hinted C2 Jacobian; FDHIP_EARLY_LOADS on the un-hinted CG2 share; per-block phase times.
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R; mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 400 python -m pytest tests/test_gpu_records.py tests/test_gpu_sliced.py -x -q -m gpu 2>&1 | grep -v "Warning\|getlimits\|_float_to_str" | tail -5 > gpurun_out/r5a_tests.txt
run() {
  env "$@" python bench.py --steps 20 --warmup 3 --only jacobian --numbering lexicographic --variants "" --no-secondary --cpu-sample 0 --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']; ap=r.get('after_packing') or {}
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  first_call_s %.3f  after_packing %s' % (r['ms'], r['assemble_ms'], r['frac'], d['setup_s']['plans_jacobian_first_call'], ap.get('ms')))"
}
{
for rep in 1 2; do
for cfg in "FDHIP_OCR_FIXED_POINT=0 FDHIP_EARLY_LOADS=0" "FDHIP_OCR_FIXED_POINT=0 FDHIP_EARLY_LOADS=1" "FDHIP_OCR_FIXED_POINT=1 FDHIP_EARLY_LOADS=0" \
           "FDHIP_OCR_FIXED_POINT=1 FDHIP_EARLY_LOADS=1" "FDHIP_OCR_FIXED_POINT=1 FDHIP_EARLY_LOADS=1 FDHIP_OCR_PACK_AFTER=0" \
           "FDHIP_OCR_FIXED_POINT=0 FDHIP_EARLY_LOADS=1 FDHIP_OCR_PACK_AFTER=0"; do
  echo "== c2 lexicographic $cfg"; timeout 120 bash -c "$(declare -f run); run $cfg"
done; done
} > gpurun_out/r5a_ab_p1.txt 2>&1
timeout 200 tools/ab_c5.sh FDHIP_EARLY_LOADS "0 1" > gpurun_out/r5a_ab_c5_early.txt 2>&1
{ timeout 120 python tools/phase_times.py c2; timeout 120 python tools/phase_times.py c5; FDHIP_EARLY_LOADS=0 timeout 120 python tools/phase_times.py c5; } > gpurun_out/r5a_phase_times.txt 2>&1
cat gpurun_out/r5a_tests.txt gpurun_out/r5a_ab_p1.txt gpurun_out/r5a_ab_c5_early.txt gpurun_out/r5a_phase_times.txt
