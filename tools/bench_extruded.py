"""tools/bench_extruded.py -- matrix assembly over an extruded set: Q1 Helmholtz on make_extruded_hex_mesh(n, layers, degree=1)
(8x8 element matrices, map + offset*layer addressing, builder.py:94-124).  Compare FDHIP_MAT_OCR=0 (direct wrapper, global
atomics) with the default (row-sliced owner-computes-rows over the derived (column, layer) map)."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from firedrake_amd import _lib, mesh as fmesh, op2          # noqa: E402
from mixed_cases import q1_hex_helmholtz_kernel              # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
layers = int(sys.argv[2]) if len(sys.argv) > 2 else n
m = fmesh.make_extruded_hex_mesh(n, layers, degree=1)
cm, xm = m.cell_node_map, m.coord_map
sp = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, None)])
mat = op2.Mat(sp)
pl = op2.LegacyParloop(q1_hex_helmholtz_kernel(), m.cell_set, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm))
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
print(f"n={n} layers={layers} cells={m.ncells} nnz={sp.nz} mode={pl._prepare()['cw'].src.mode} first_call_s={first:.2f} "
      f"assemble_ms={1e3 * min(ts):.3f} values GB/s={sp.nz * 8 / min(ts) / 1e9:.0f}")
