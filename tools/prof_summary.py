#!/usr/bin/env python
"""Summarise a rocprofv3 --kernel-trace --stats sqlite database (rocpd) as plain text:
per-kernel calls / total / average duration, plus launch geometry and register usage."""
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    if "rocprim" in name:
        for tag in ("radix_sort_onesweep_iteration", "radix_sort_onesweep_global_offsets", "partition_impl", "transform_impl",
                    "init_lookback_scan_state"):
            if tag in name:
                return "rocprim::" + tag
    return name.split("(")[0][:70]


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    rows = list(cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"{'kernel':<60} {'calls':>6} {'total_us':>12} {'avg_us':>12} {'%':>7}")
    for n, c, t, a, p in rows:
        print(f"{short(n):<60} {c:>6} {t / 1e3 if t > 1e7 else t:>12.1f} {a / 1e3 if t > 1e7 else a:>12.1f} {p:>7.2f}")
    print()
    print(f"{'kernel':<60} {'grid':>10} {'wg':>5} {'lds':>7} {'scratch':>8} {'vgpr':>5} {'agpr':>5} {'sgpr':>5}")
    seen = set()
    for r in cur.execute("select name, grid_x, workgroup_x, lds_size, scratch_size, vgpr_count, accum_vgpr_count, sgpr_count from kernels"):
        k = (short(r[0]), r[1], r[3])
        if k in seen or "rocprim" in r[0] or "rocclr" in r[0]:
            continue
        seen.add(k)
        print(f"{short(r[0]):<60} {r[1]:>10} {r[2]:>5} {r[3]:>7} {r[4]:>8} {r[5]:>5} {r[6]:>5} {r[7]:>5}")


if __name__ == "__main__":
    main(sys.argv[1])
