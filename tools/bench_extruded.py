"""tools/bench_extruded.py [n] [layers] [--variable] [--interior] [--check] -- loops over an extruded set: Q1 Helmholtz matrix (8x8 element matrices) and
a Q1 nodal accumulation (Dat INC) on make_extruded_hex_mesh(n, layers, degree=1), map + offset*layer addressing
(builder.py:94-124).  --variable: a bathymetry -- every column its own [bottom, top) (set.py:326-337), the map row of a column
pointing at its own bottom cell -- i.e. VARIABLE layers.  Compare FDHIP_MODE=direct (one lane per column walking its layers,
global atomics) with the default (staged / row-sliced owner-computes-rows over the derived map of the existing cells)."""
import os
import sys
import time

import numpy as np

_ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [_ROOT, os.path.join(_ROOT, "tests")]
from firedrake_amd import _lib, mesh as fmesh, op2          # noqa: E402
from mixed_cases import q1_hex_helmholtz_kernel              # noqa: E402

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
variable = "--variable" in sys.argv
n = int(argv[0]) if len(argv) > 0 else 128
layers = int(argv[1]) if len(argv) > 1 else n
m = fmesh.make_extruded_hex_mesh(n, layers, degree=1)
cm, xm, cells = m.cell_node_map, m.coord_map, m.cell_set
ncells = m.ncells
if variable:
    # bottom of column (i, j): a smooth bowl, up to half the column missing; node levels [bottom, layers + 1)
    pts = m.node_points[np.asarray(cm.values_with_halo)[:, 0]]
    bot = np.floor(0.5 * layers * (np.sin(np.pi * pts[:, 0]) * np.sin(np.pi * pts[:, 1])) ** 2).astype(np.int64)
    la = np.stack([bot, np.full_like(bot, layers + 1)], axis=1)
    cells = op2.ExtrudedSet(m.base_set, layers=la)
    cm = op2.Map(cells, m.node_set, cm.arity, (np.asarray(cm.values_with_halo) + np.asarray(cm.offset)[None, :] * bot[:, None]).astype(np.int32),
                 offset=list(cm.offset))
    xm = op2.Map(cells, m.coord_node_set, xm.arity, (np.asarray(xm.values_with_halo) + np.asarray(xm.offset)[None, :] * bot[:, None]).astype(np.int32),
                 offset=list(xm.offset))
    ncells = int((la[:, 1] - 1 - la[:, 0]).sum())
interior = "--interior" in sys.argv
y = op2.Dat(m.node_set)
if interior:
    # --interior: the horizontal interior facets of the columns (dS_h, ON_INTERIOR_FACETS: the kernel sees the cell below and the cell
    # above, 16 nodes; builder.py:806-809) -- a jump-penalty-like 16 x 16 matrix and a 16-node accumulation
    sp = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, [op2.ON_INTERIOR_FACETS])])
    mat = op2.Mat(sp)
    kf = op2.Kernel("static void q1_face(double *A, const double *x) { double h = 0.0; for (int i = 0; i < 8; ++i) h += x[3*(8 + i) + 2] - x[3*i + 2]; "
                    "for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) A[i*16 + j] += 0.125 * h * ((i < 8) == (j < 8) ? 1.0 : -1.0) * (1.0 + x[3*i] * x[3*j + 1]); }",
                    "q1_face")
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS)
    pl = op2.LegacyParloop(kf, cells, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), **kw)
    kd = op2.Kernel("static void q1_jump(double *y, const double *x) { for (int i = 0; i < 16; ++i) { double s = 0.0; "
                    "for (int j = 0; j < 16; ++j) s += x[3*j] * x[3*((i + j) & 15) + 1] + x[3*j + 2]; y[i] += s; } }", "q1_jump")
    pd = op2.LegacyParloop(kd, cells, y(op2.INC, cm), m.coordinates(op2.READ, xm), **kw)
    ncells = int(np.maximum(la[:, 1] - 2 - la[:, 0], 0).sum()) if variable else m.base_set.size * (layers - 1)
else:
    kw = {}
    sp = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, None)])
    mat = op2.Mat(sp)
    pl = op2.LegacyParloop(q1_hex_helmholtz_kernel(), cells, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm))
    kd = op2.Kernel("static void q1_acc(double *y, const double *x) { for (int i = 0; i < 8; ++i) { double s = 0.0; "
                    "for (int j = 0; j < 8; ++j) s += x[3*j] * x[3*((i + j) & 7) + 1] + x[3*j + 2]; y[i] += s; } }", "q1_acc")
    pd = op2.LegacyParloop(kd, cells, y(op2.INC, cm), m.coordinates(op2.READ, xm))


def timed(zero, loop):
    t0 = time.perf_counter()
    zero(); loop(); _lib.call("fd_device_sync")
    first = time.perf_counter() - t0
    ts = []
    for _ in range(5):
        zero()
        _lib.call("fd_device_sync")
        t0 = time.perf_counter()
        loop()
        _lib.call("fd_device_sync")
        ts.append(time.perf_counter() - t0)
    return first, min(ts)


f1, t1 = timed(mat.zero, pl)
f2, t2 = timed(y.zero, pd)
print(f"n={n} layers={layers} variable={variable} interior_facets={interior} cells={ncells} nnz={sp.nz} FDHIP_MODE={os.environ.get('FDHIP_MODE', 'auto')}: "
      f"matrix mode={pl._prepare()['cw'].src.mode} first_call_s={f1:.2f} assemble_ms={1e3 * t1:.3f} ({sp.nz * 8 / t1 / 1e9:.0f} GB/s of values); "
      f"Dat loop mode={pd._prepare()['cw'].src.mode} first_call_s={f2:.2f} ms={1e3 * t2:.3f} ({ncells / t2 / 1e9:.2f} Gcells/s)")
if "--check" in sys.argv:
    from helpers import oracle_run
    ref = oracle_run(pl.global_kernel.local_kernel, cells, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), **kw)[0]
    got = mat.csr()
    assert np.array_equal(got[0], ref.rowptr) and np.array_equal(got[1], ref.colidx)
    print("  matrix vs oracle: max |diff| / max |A| = %.2e" % (np.abs(got[2] - ref.values).max() / np.abs(ref.values).max()))
    yr = oracle_run(kd, cells, op2.Dat(m.node_set)(op2.INC, cm), m.coordinates(op2.READ, xm), **kw)[0]
    print("  Dat loop vs oracle: max |diff| / max |y| = %.2e" % (np.abs(y.data_ro - yr[:, None] if y.data_ro.ndim > 1 else y.data_ro - yr).max() / np.abs(yr).max()))
