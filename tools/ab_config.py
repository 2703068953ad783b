#!/usr/bin/env python
"""tools/ab_config.py KEY v1,v2[,...] WORKLOAD [reps] -- same-box A/B of a TUNING CONSTANT of firedrake_amd.configuration (the ones
that are not environment switches) on a secondary workload of bench.py: c4 (DG advection right-hand side), c3 (Q4 hex), c3a (Q4 action
n = 64), c5 (the CG2 share, un-hinted), c2 (the headline).  Prints the per-loop kernel times of every setting, alternating `reps` times."""
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
        elif workload in ("c5", "c2"):
            import subprocess
            # (the Poisson lines build their own process state: run bench.py as a child with the constant patched through a tiny shim)
            code = ("import sys; sys.argv = ['bench.py'] + %r; from firedrake_amd.configuration import configuration as c; c[%r] = type(c[%r])(%r); "
                    "import runpy; runpy.run_path(%r, run_name='__main__')") % (
                (["--workload", "c5", "--n", "107", "--numbering", "lexicographic"] if workload == "c5" else ["--variants", "", "--no-secondary", "--traffic", "off"])
                + ["--steps", "20", "--warmup", "3", "--cpu-sample", "0"], key, key, v, os.path.join(ROOT, "bench.py"))
            out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
            try:
                import json
                d = json.loads(out[-1])
                r, q = d["roofline_jacobian"], d["roofline_residual"]
                print(f"{key}={v}: step {d['ms_per_step']:.4f} ms | residual {q['ms']:.4f} ms ({q['frac']:.3f}) | jacobian {r['ms']:.4f} ms ({r['frac']:.3f})", flush=True)
            except Exception as exc:
                print(f"{key}={v}: FAILED {exc!r} {out[-1][:200] if out else ''}", flush=True)
        elif workload == "c3a":
            d = bench.measure_c3_action(64, 20, 3)
            print(f"{key}={v}: action n=64 {d['kernel_ms']:.4f} ms frac_valu {d['frac_valu']:.3f}", flush=True)
        else:
            d = bench.measure_c3(32, 5, 2)
            print(f"{key}={v}: matrix {d['roofline']['ms']:.3f} ms ({d['roofline']['frac']:.3f}) action {d['roofline_action']['ms']:.4f} ms", flush=True)
