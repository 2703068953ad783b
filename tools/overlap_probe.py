"""Does the step gain from putting the Jacobian on a second stream?  And what do the 7 events per step of bench.measure cost?
python tools/overlap_probe.py [n] [degree]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firedrake_amd import _lib, forms, mesh as fmesh                       # noqa: E402
from firedrake_amd.device import Event, Stream                             # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
degree = int(sys.argv[2]) if len(sys.argv) > 2 else 1
numbering = sys.argv[3] if len(sys.argv) > 3 else "lexicographic"
K = 20
mesh = fmesh.UnitCubeMesh((n, n, n), degrees=(degree,), perturb=0.1, tile=(8, 8, 4), numbering=numbering)
prob = forms.PoissonProblem(mesh, degree, bcs=True)
sync = lambda: _lib.call("fd_device_sync")
for _ in range(3):
    prob.assemble_residual()
    prob.assemble_jacobian()
sync()
side = Stream()


def serial(ev=None):
    prob.u.dat_version += 1
    if ev:
        ev[0].record()
        prob.assemble_residual(events=(ev[1], ev[2]))
        ev[3].record()
        prob.assemble_jacobian(events=(ev[4], ev[5]))
        ev[6].record()
    else:
        prob.assemble_residual()
        prob.assemble_jacobian()


def overlapped(ev=None):
    prob.u.dat_version += 1
    with side.fork():
        prob.assemble_jacobian()
    prob.assemble_residual()
    side.join()


def overlapped_rj(ev=None):
    prob.u.dat_version += 1
    with side.fork():
        prob.assemble_residual()
    prob.assemble_jacobian()
    side.join()


def timeit(fn, with_events=False):
    ev = [[Event() for _ in range(7)] for _ in range(K)] if with_events else [None] * K
    sync()
    t0 = time.perf_counter()
    for k in range(K):
        fn(ev[k])
    sync()
    return (time.perf_counter() - t0) / K * 1e3


ref_r = prob.assemble_residual().data_ro.copy()
prob.assemble_jacobian()
for name, fn, we in (("serial, no events", serial, False), ("serial, 7 events per step", serial, True),
                     ("jacobian on a side stream", overlapped, False), ("residual on a side stream", overlapped_rj, False)) * 3:
    print(f"{name:32s} {timeit(fn, we):.4f} ms per step", flush=True)
overlapped()
sync()
r2 = prob.r.data_ro.copy()
print("residual after an overlapped step: max |diff| =", float(np.abs(r2 - ref_r).max()), "of", float(np.abs(ref_r).max()))
