#!/usr/bin/env python
"""tools/trace_summary.py <t_kernel_stats.csv> [n] -- the top kernels of a rocprofv3 --kernel-trace --stats CSV as plain text."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
top = int(sys.argv[2]) if len(sys.argv) > 2 else 24
for r in rows[:top]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    if "rocprim" in n:
        n = "rocprim::" + next((t for t in ("radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets", "partition_impl", "transform_impl",
                                            "init_lookback", "scan") if t in n), "other")
    print("%-56s calls=%-5s avg_us=%10.1f total_us=%11.1f" % (n.split("(")[0][:56], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                             float(r["TotalDurationNs"]) / 1e3))
