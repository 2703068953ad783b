#!/bin/bash
# tools/pmc_c5_r3.sh <numbering> -- limiter + traffic counters of the row-sliced CG2 Jacobian (bench.py --workload c5 --n 107) under one
# numbering; counters in their own passes, kernel trace only.  Output: gpurun_out/r3_pmc_c5_<numbering>.txt
NB=${1:-tiled}
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--workload c5 --n 107 --steps 3 --warmup 1 --cpu-sample 0 --numbering $NB"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r3pmc5_${NB}_$name -o p -- python $R/bench.py $ARGS > $R/gpurun_out/r3pmc5_${NB}_$name.log 2>&1
done
cd $R
{ echo "== wrap_poisson_p2_tet_jacobian, numbering=$NB"; python tools/pmc_summary.py wrap_poisson_p2_tet_jacobian gpurun_out/r3pmc5_${NB}_*/;
  python bench.py --workload c5 --n 107 --steps 10 --warmup 2 --cpu-sample 0 --numbering $NB 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('kernel_ms %.4f algorithmic_GB %.3f' % (r['ms'], r['algorithmic_bytes']/1e9))"; } > gpurun_out/r3_pmc_c5_$NB.txt
rm -rf gpurun_out/r3pmc5_${NB}_*
cat gpurun_out/r3_pmc_c5_$NB.txt
