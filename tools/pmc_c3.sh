#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c3_trace -o t -- python $R/bench.py --workload c3 --steps 3 --warmup 1 > $R/gpurun_out/c3_trace.log 2>&1
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_F64" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "FETCH_SIZE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/c3_pmc_$name -o p -- python $R/bench.py --workload c3 --steps 3 --warmup 1 > $R/gpurun_out/c3_pmc_$name.log 2>&1
done
cd $R
python tools/pmc_summary.py wrap_helmholtz_q4_hex_jacobian gpurun_out/c3_pmc_*/
python - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/c3_trace/t_kernel_stats.csv")))[:6]:
    print(r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3)
PY
