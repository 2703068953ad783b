"""tools/bench_vector.py -- vector-valued matrix assembly (MatSetValuesBlockedLocal): vector P1 elasticity-like element
matrices (12x12, 3x3 blocks) on UnitCubeMesh(n) tets; prints the time per assembly and the wrapper shape chosen.
Compare FDHIP_MAT_OCR=0 (direct wrapper, global atomics) with the default (row-sliced owner-computes-rows)."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from firedrake_amd import _lib, mesh as fmesh, op2          # noqa: E402
from mixed_cases import vector_p1_elasticity_kernel          # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
numbering = sys.argv[2] if len(sys.argv) > 2 else "tiled"
mesh = fmesh.UnitCubeMesh(n, degrees=(1,), perturb=0.1, numbering=numbering)
V = mesh.space(1)
cm = V.cell_node_map
sp = op2.Sparsity((V.node_set ** 3, V.node_set ** 3), [(cm, cm, None)])
lg = np.arange(V.node_set.total_size, dtype=np.int32)
lg[V.boundary_nodes] = -1
mat = op2.Mat(sp)
pl = op2.LegacyParloop(vector_p1_elasticity_kernel(3), mesh.cell_set, mat(op2.INC, (cm, cm), lgmaps=(lg, lg)), mesh.coordinates(op2.READ, cm))
t0 = time.perf_counter()
mat.zero(); pl(); _lib.call("fd_device_sync")
first = time.perf_counter() - t0
ts = []
for _ in range(5):
    mat.zero()
    _lib.call("fd_device_sync")
    t0 = time.perf_counter()
    pl()
    _lib.call("fd_device_sync")
    ts.append(time.perf_counter() - t0)
mode = pl._prepare()["cw"].src.mode
print(f"n={n} numbering={numbering} cells={mesh.cell_set.size} scalar nnz={sp.nz} mode={mode} first_call_s={first:.2f} assemble_ms={1e3 * min(ts):.3f} "
      f"values GB/s={sp.nz * 8 / min(ts) / 1e9:.0f}")
