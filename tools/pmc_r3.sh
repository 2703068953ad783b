#!/bin/bash
# tools/pmc_r3.sh <numbering> -- limiter counters of the C2 Jacobian kernel under one numbering (separate --pmc passes, kernel
# trace only) + the bench line with live FETCH/WRITE traffic.  Output: gpurun_out/r3_pmc_<numbering>.txt
NB=${1:-tiled}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --variants= --traffic off --no-secondary --only jacobian --numbering $NB"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r3pmc_${NB}_$name -o p -- python $R/bench.py $ARGS > $R/gpurun_out/r3pmc_${NB}_$name.log 2>&1
done
cd $R
{ echo "== wrap_poisson_p1_tet_jacobian, numbering=$NB"; python tools/pmc_summary.py wrap_poisson_p1_tet_jacobian gpurun_out/r3pmc_${NB}_*/;
  python bench.py --steps 10 --warmup 2 --cpu-sample 0 --variants= --no-secondary --only jacobian --numbering $NB 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('kernel_ms %.4f traffic_GB %.3f algorithmic_GB %.3f counters_kb %s' % (r['ms'], (r['traffic'] or 0)/1e9, r['algorithmic_bytes']/1e9, r.get('traffic_counters_kb')))"; } > gpurun_out/r3_pmc_$NB.txt
rm -rf gpurun_out/r3pmc_${NB}_*
cat gpurun_out/r3_pmc_$NB.txt
