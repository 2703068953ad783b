"""tools/debug_pack.py -- packer at construction against packer after launches (same process): instance order, tables, kernel time."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from firedrake_amd import forms, mesh as fmesh, _lib
from firedrake_amd.configuration import configuration
from firedrake_amd.device import Event

def down(ptr, dt, shape):
    a = np.empty(shape, dtype=dt)
    _lib.call("fd_memcpy_d2h", a.ctypes.data, ptr, a.nbytes, None)
    return a

def time_jac(prob, n=8):
    evs = [(Event(), Event()) for _ in range(n)]
    for a, b in evs:
        prob.assemble_jacobian(events=(a, b))
    _lib.call("fd_device_sync")
    return float(np.median([a.elapsed_ms(b) for a, b in evs[2:]]))

n = int(sys.argv[1]) if len(sys.argv) > 1 else 215
m = fmesh.UnitCubeMesh(n, degrees=(1,), perturb=0.1, numbering="lexicographic")
out = {}
for tag, after in (("early", 0), ("late", 64)):
    configuration["ocr_pack_after"] = after
    prob = forms.PoissonProblem(m, 1, bcs=True)
    prob.assemble_jacobian()
    loop = prob.jacobian()[1]
    geo = next(g for key, g in loop._prepared["parts"].items() if key[0] == "ocr")
    op = geo["ocr"]
    t0 = time_jac(prob)
    print(tag, "packed", op.packed, "kernel_ms", round(t0, 4), "variant", geo["cw"].src.mode)
    if not op.packed:
        op.pack()
        _lib.call("fd_device_sync")
        print(tag, "after pack(): kernel_ms", round(time_jac(prob), 4), round(time_jac(prob), 4))
    key = next(iter(op.plans))
    out[tag] = (down(op.inst_ent, np.int32, (op.ninst,)), down(op.plans[key].lmap, np.uint16, (op.ninst, 4)),
                down(op.kidx.ptr, np.uint8, (op.ninst, 16)), prob.jacobian()[0].csr()[2])
    del prob, loop, geo, op
for i, name in enumerate(("inst_ent", "lmap", "kidx")):
    print(name, "equal:", np.array_equal(out["early"][i], out["late"][i]))
print("values max diff", np.abs(out["early"][3] - out["late"][3]).max())
