"""Device memory over twelve problems built, run and dropped in a row (python tools/leak_probe.py): free memory after del + gc must not go down."""
import ctypes, gc, os, sys, time
sys.path.insert(0, os.getcwd())
from firedrake_amd import _lib, forms, mesh as fmesh
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    f = ctypes.c_size_t(); t = ctypes.c_size_t()
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**20
_lib.require_gpu()
print("start free MB", round(free_mb()))
for rep in range(12):
    n = 40 + (rep % 3)
    prob = forms.PoissonProblem(fmesh.UnitCubeMesh((n, n, n), degrees=(1,), perturb=0.1, numbering="lexicographic"), 1, bcs=True)
    for _ in range(3):
        prob.assemble_residual(); prob.assemble_jacobian()
    _lib.call("fd_device_sync")
    used_live = free_mb()
    del prob
    gc.collect()
    _lib.call("fd_device_sync")
    print(f"rep {rep} n={n}: free with problem alive {used_live:.0f} MB, after del+gc {free_mb():.0f} MB", flush=True)
