#!/bin/bash
# tools/pmc.sh <tag> <bench args...> -- rocprofv3 PMC passes (own runs, no tracing domains) + kernel trace.
# Results land in gpurun_out/<tag>_*.  HBM bytes follow MI355X_MICROARCH.md: FETCH_SIZE/WRITE_SIZE in separate passes.
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o t -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_pmc_$name -o p -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_pmc_$name.log 2>&1
done
cd $R
