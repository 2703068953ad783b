#!/bin/bash
# tools/pmc_traffic.sh <tag> <bench args...> -- the two rocprofv3 PMC passes behind roofline.traffic only
# (FETCH_SIZE and WRITE_SIZE in separate runs, no tracing domains; MI355X_MICROARCH.md HBM section).
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for set in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/${TAG}_pmc_$set -o p -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_pmc_$set.log 2>&1
done
cd $R
python tools/pmc_to_json.py $TAG gpurun_out/${TAG}_traffic.json
