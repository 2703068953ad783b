"""How many hipcc runs do NEW MESHES of a known kernel cost?  The C2 step (Poisson CG1 residual + Jacobian) on cubes of n = 40 .. 47
per axis, one after another in one process with an empty wrapper cache, for two settings of configuration["lds_const_stride"]
(1 = exact node strides compiled into the staged wrappers, 16 = rounded up to multiples of 16 nodes).
    python tools/variant_reuse_probe.py [n0] [count]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 40
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
CHILD = r"""
import sys, json, time
sys.path.insert(0, %r)
from firedrake_amd.configuration import configuration as c
c["lds_const_stride"] = %d
from firedrake_amd import _lib, compilation, forms, mesh as fmesh
rows = []
for n in range(%d, %d):
    before = compilation.stats["hipcc_runs"]
    t0 = time.perf_counter()
    prob = forms.PoissonProblem(fmesh.UnitCubeMesh((n, n, n), degrees=(1,), perturb=0.1, numbering="lexicographic"), 1, bcs=True)
    prob.assemble_residual(); prob.assemble_jacobian(); _lib.call("fd_device_sync")
    rows.append({"n": n, "hipcc_runs": compilation.stats["hipcc_runs"] - before, "first_step_s": round(time.perf_counter() - t0, 3)})
print(json.dumps(rows))
"""
for g in (1, 16):
    with tempfile.TemporaryDirectory() as cache:
        env = dict(os.environ, FDHIP_CACHE_DIR=cache)
        out = subprocess.run([sys.executable, "-c", CHILD % (ROOT, g, n0, n0 + count)], capture_output=True, text=True, env=env)
        try:
            rows = json.loads(out.stdout.strip().splitlines()[-1])
        except Exception:
            print(f"lds_const_stride={g}: FAILED\n{out.stderr[-2000:]}")
            continue
        print(f"lds_const_stride={g}: hipcc runs per new mesh " + " ".join(f"n={r['n']}:{r['hipcc_runs']}({r['first_step_s']}s)" for r in rows)
              + f" | total {sum(r['hipcc_runs'] for r in rows)}", flush=True)
