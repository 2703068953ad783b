#!/bin/bash
# tools/ab_r4a.sh -- round 4, A/B of the owner-computes-rows index tables on the C2 Jacobian (one box):
# bit-packed instance records (FDHIP_OCR_RECORDS, _DIAG) and the run-coded flush of a derived row order (FDHIP_OCR_RUN_FLUSH)
run() {  # numbering, env...
  nb=$1; shift
  env "$@" python bench.py --steps 20 --warmup 3 --only jacobian --numbering $nb --variants "" --no-secondary --cpu-sample 0 --traffic ${TRAFFIC:-off} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
t=r.get('traffic')
print('  kernel_ms %.4f  assemble_ms %.4f  frac %.4f  traffic_GB %s  first_call_s %.3f' % (r['ms'], r['assemble_ms'], r['frac'], ('%.3f' % (t/1e9)) if t else '-', d['setup_s']['plans_jacobian_first_call']))"
}
for rep in 1 2; do
for cfg in "FDHIP_OCR_RECORDS=0 FDHIP_OCR_RUN_FLUSH=0" "FDHIP_OCR_RECORDS=1 FDHIP_OCR_RUN_FLUSH=0" "FDHIP_OCR_RECORDS=0 FDHIP_OCR_RUN_FLUSH=1" \
           "FDHIP_OCR_RECORDS=1 FDHIP_OCR_RUN_FLUSH=1 FDHIP_OCR_RECORDS_DIAG=0" "FDHIP_OCR_RECORDS=1 FDHIP_OCR_RUN_FLUSH=1"; do
  echo "== lexicographic $cfg"; run lexicographic $cfg
done
for cfg in "FDHIP_OCR_RECORDS=0" "FDHIP_OCR_RECORDS=1 FDHIP_OCR_RECORDS_DIAG=0" "FDHIP_OCR_RECORDS=1"; do
  echo "== tiled $cfg"; run tiled $cfg
done
done
echo "== traffic (PMC child passes), lexicographic"
for cfg in "FDHIP_OCR_RECORDS=0 FDHIP_OCR_RUN_FLUSH=0" "FDHIP_OCR_RECORDS=1 FDHIP_OCR_RUN_FLUSH=1"; do echo "== $cfg"; TRAFFIC=auto run lexicographic $cfg; done
