#!/bin/bash
# tools/pmc_c5.sh -- limiter counters of the P2 kernels (bench.py --workload c5), counters in their own runs, no tracing
# domains.  Extra environment for the bench in $1.  Output: gpurun_out/r2_pmc_c5.txt
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
ARGS="--workload c5 --steps 3 --warmup 1 --cpu-sample 0 --variants= --traffic off --no-secondary"
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 env $1 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/r2pmc_c5_$name -o p -- python $R/bench.py $ARGS > $R/gpurun_out/r2pmc_c5_$name.log 2>&1
done
cd $R
for k in wrap_poisson_p2_tet_jacobian wrap_poisson_p2_tet_residual; do
  echo "== $k" ; python tools/pmc_summary.py $k gpurun_out/r2pmc_c5_*/
done > gpurun_out/r2_pmc_c5.txt
cat gpurun_out/r2_pmc_c5.txt
rm -rf gpurun_out/r2pmc_c5_*
