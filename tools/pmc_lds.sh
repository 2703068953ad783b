#!/bin/bash
# tools/pmc_lds.sh <tag> <bench args...> -- one rocprofv3 PMC pass with the LDS counters only
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_ACTIVE_INST_LDS --output-format csv -d $R/gpurun_out/${TAG}_pmc_lds -o p -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_pmc_lds.log 2>&1
cd $R
python tools/pmc_summary.py wrap_poisson_p1_tet_jacobian gpurun_out/${TAG}_pmc_lds
