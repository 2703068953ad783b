#!/bin/bash
# tools/pmc_r2.sh -- round-2 PMC evidence: limiter counters of the C2 kernels (three passes) and the MFMA counters of the
# Q4 matrix kernel (two passes); counters in their own runs, no tracing domains.  Output: gpurun_out/r2_pmc_*.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --variants= --traffic off --no-secondary"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r2pmc_c2_$name -o p -- python $R/bench.py $ARGS > $R/gpurun_out/r2pmc_c2_$name.log 2>&1
done
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r2pmc_c3_$name -o p -- python $R/bench.py --workload c3 --steps 3 --warmup 1 > $R/gpurun_out/r2pmc_c3_$name.log 2>&1
done
cd $R
for k in wrap_poisson_p1_tet_jacobian wrap_poisson_p1_tet_residual; do
  echo "== $k" ; python tools/pmc_summary.py $k gpurun_out/r2pmc_c2_*/
done > gpurun_out/r2_pmc_c2.txt
for k in wrap_helmholtz_q4_hex_jacobian wrap_helmholtz_q4_hex_action; do
  echo "== $k" ; python tools/pmc_summary.py $k gpurun_out/r2pmc_c3_*/
done > gpurun_out/r2_pmc_c3.txt
cat gpurun_out/r2_pmc_c2.txt gpurun_out/r2_pmc_c3.txt
rm -rf gpurun_out/r2pmc_c2_* gpurun_out/r2pmc_c3_*
