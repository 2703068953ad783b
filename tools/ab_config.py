#!/usr/bin/env python
"""tools/ab_config.py KEY v1,v2[,...] WORKLOAD [reps] -- same-box A/B of a TUNING CONSTANT of firedrake_amd.configuration (the ones
that are not environment switches) on a secondary workload of bench.py: c4 (DG advection right-hand side), c3 (Q4 hex), c3a (Q4 action
n = 64).  Prints the per-loop kernel times of every setting, alternating `reps` times."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from firedrake_amd.configuration import configuration  # noqa: E402

key, values, workload = sys.argv[1], sys.argv[2].split(","), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
conv = type(configuration[key])
for _ in range(reps):
    for v in values:
        configuration[key] = conv(v)
        if workload == "c4":
            d = bench.measure_c4(2048, 20, 3)
            per = " | ".join("%s %.4f ms (%.3f)" % (r["kernel"].replace("wrap_dg_adv_", ""), r["ms"], r["frac"]) for r in d["roofline_per_loop"])
            print(f"{key}={v}: step {d['ms_per_step']:.4f} ms | {per}", flush=True)
        elif workload == "c3a":
            d = bench.measure_c3_action(64, 20, 3)
            print(f"{key}={v}: action n=64 {d['kernel_ms']:.4f} ms frac_valu {d['frac_valu']:.3f}", flush=True)
        else:
            d = bench.measure_c3(32, 5, 2)
            print(f"{key}={v}: matrix {d['roofline']['ms']:.3f} ms ({d['roofline']['frac']:.3f}) action {d['roofline_action']['ms']:.4f} ms", flush=True)
