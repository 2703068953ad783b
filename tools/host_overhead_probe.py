"""How much HOST time does one step of the C2 benchmark take to enqueue?  (If it approaches the device time of the step, launches
arrive late and the step is partly host-bound.)  python tools/host_overhead_probe.py [n]"""
import cProfile
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firedrake_amd import _lib, forms, mesh as fmesh                       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
mesh = fmesh.UnitCubeMesh((n, n, n), degrees=(1,), perturb=0.1, tile=(8, 8, 4), numbering="lexicographic")
prob = forms.PoissonProblem(mesh, 1, bcs=True)
sync = lambda: _lib.call("fd_device_sync")


def step():
    prob.u.dat_version += 1
    prob.assemble_residual()
    prob.assemble_jacobian()


for _ in range(5):
    step()
sync()
K = 50
for rep in range(3):
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    t1 = time.perf_counter()
    sync()
    t2 = time.perf_counter()
    print(f"enqueue {1e3 * (t1 - t0) / K:.4f} ms per step (host), drained after {1e3 * (t2 - t0) / K:.4f} ms per step (device)", flush=True)
# host time alone: a tiny problem has no device time to hide behind
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    step()
pr.disable()
sync()
pstats.Stats(pr).sort_stats("cumulative").print_stats(25)
