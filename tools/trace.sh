#!/bin/bash
# tools/trace.sh <tag> <bench args...> -- rocprofv3 kernel trace + stats (CSV) of a bench.py run
TAG=$1; shift
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_trace -o t -- python $R/bench.py "$@" > $R/gpurun_out/${TAG}_trace.log 2>&1
cd $R
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/${TAG}_trace/t_kernel_stats.csv")))
for r in rows[:14]:
    n = r["Name"]
    n = n.replace("(anonymous namespace)::", "")
    if "rocprim" in n:
        n = "rocprim::" + [t for t in ("radix_sort_onesweep_iteration","radix_sort_onesweep_global_offsets","partition_impl","transform_impl","init_lookback") if t in n][0]
    print("%-52s calls=%-4s avg_us=%10.1f total_us=%10.1f" % (n.split("(")[0][:52], r["Calls"], float(r["AverageNs"])/1e3, float(r["TotalDurationNs"])/1e3))
PY
