#!/bin/bash
# tools/pmc_r4.sh <tag> <kernel symbol> <bench.py args...> -- limiter + traffic counters of one wrapper kernel (separate --pmc
# passes, kernel trace only, as MI355X_MICROARCH prescribes) and the kernel time of the same command un-profiled.
# Output: gpurun_out/r4_pmc_<tag>.txt
TAG=$1; KERN=$2; shift 2
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_WAIT_INST_LDS SQ_WAIT_ANY" "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/r4pmc_${TAG}_$name -o p -- python $R/bench.py "$@" --steps 3 --warmup 1 --cpu-sample 0 --traffic off > $R/gpurun_out/r4pmc_${TAG}_$name.log 2>&1
done
cd $R
{ echo "== $KERN: bench.py $*"; python tools/pmc_summary.py $KERN gpurun_out/r4pmc_${TAG}_*/;
  python bench.py "$@" --steps 10 --warmup 2 --cpu-sample 0 --traffic off 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline_jacobian']
print('kernel_ms %.4f algorithmic_GB %.3f frac %.4f' % (r['ms'], r['algorithmic_bytes']/1e9, r['frac']))"; } > gpurun_out/r4_pmc_$TAG.txt
rm -rf gpurun_out/r4pmc_${TAG}_*
cat gpurun_out/r4_pmc_$TAG.txt
