#!/usr/bin/env python
"""tools/setup_profile.py [n] [numbering] [degree] -- where the time to first assemble goes at C2 size: wall time per phase and, with
FDHIP_PROFILE_CALLS=1 (set here), per C-ABI entry point including the device work each call queued."""
import os
import sys
import time

os.environ["FDHIP_PROFILE_CALLS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from firedrake_amd import _lib, forms, mesh as fmesh   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
nb = sys.argv[2] if len(sys.argv) > 2 else "lexicographic"
_lib.require_gpu()
t0 = time.perf_counter()
degree = int(sys.argv[3]) if len(sys.argv) > 3 else 1
m = fmesh.UnitCubeMesh(n, degrees=(degree,), perturb=0.1, numbering=nb)
prob = forms.PoissonProblem(m, degree, bcs=True)
print(f"mesh + problem      {time.perf_counter() - t0:8.3f} s")
_lib.profile_report()
import cProfile
import io
import pstats
for label, fn in (("sparsity", lambda: prob.jacobian()[0].sparsity._build()), ("residual first call", prob.assemble_residual),
                  ("jacobian first call", prob.assemble_jacobian), ("residual second call", prob.assemble_residual),
                  ("jacobian second call", prob.assemble_jacobian)):
    pr = cProfile.Profile()
    t0 = time.perf_counter()
    pr.enable()
    fn()
    pr.disable()
    _lib.load().fd_device_sync()
    dt = time.perf_counter() - t0
    if "first" in label:
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(28)
        print("\n".join(l for l in buf.getvalue().splitlines() if l.strip())[:6000])
    print(f"== {label:<22} {dt:8.3f} s (wall, every C-ABI call synchronised)")
    rep = _lib.profile_report()
    tot = sum(float(l.split()[-2]) for l in rep.splitlines()) if rep else 0.0
    print(rep)
    print(f"   C-ABI total {tot:.3f} s, Python / numpy outside the library {dt - tot:.3f} s")
