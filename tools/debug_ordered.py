import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from firedrake_amd import op2, _lib
from firedrake_amd.configuration import configuration
import golden_kernels as gk
from helpers import oracle_run, structured_tri_mesh, plan_ref, lane_slot_to_entity
configuration["debug"] = 1
coords, cells = structured_tri_mesh(64, 64, perturb=0.2)
nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
m = op2.Map(ele, nodes, 3, cells)
x = op2.Dat(nodes ** 2, coords)
f = op2.Dat(nodes, np.random.default_rng(3).standard_normal(len(coords)))
krhs = op2.Kernel(gk.RHS_Q6, "rhs_q6")
ob = oracle_run(krhs, ele, op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))[0]
b = op2.Dat(nodes)
pl = op2.LegacyParloop(krhs, ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
pl()
print("err ordered", np.abs(b.data_ro - ob).max())
geo = pl._staged_geometry(0, len(cells))
print("mode", geo["cw"].src.mode, "range", geo["range"], "epb", geo["epb"], "lds", geo["lds"])
order = geo["order"].download(np.int32, (len(cells),))
print("perm ok", sorted(order.tolist()) == list(range(len(cells))))
plan = next(iter(geo["plans"].values()))
blk, lst, lm = plan.download()
dm = [mm for mm in m._derived.values()][0]
rows = dm._dev.download(np.int32, (len(cells), 3))
print("gather ok", np.array_equal(rows, cells[order]))
bst = np.empty(plan.nblocks + 1, dtype=np.int32)
_lib.call("fd_memcpy_d2h", bst.ctypes.data, plan.bstart, bst.nbytes, None)
print("bstart", bst[:5], bst[-3:], "nblocks", plan.nblocks)
rb, rl, rlm = plan_ref(rows, 0, len(cells), geo["epb"])
T = geo["cw"].src.lane_threads
for b0 in range(0, len(cells), geo["epb"]):
    b1 = min(len(cells), b0 + geo["epb"])
    rlm[b0:b1] = rlm[b0:b1][lane_slot_to_entity(b1 - b0, T)]
print("plan blk", np.array_equal(blk, rb), "list", np.array_equal(lst, rl), "lmap", np.array_equal(lm, rlm))
print(geo["cw"].src.layout)
src = geo["cw"].src.source
i = src.index('extern "C"')
print(src[i:i + 5000])
configuration["locality_order"] = 0
b2 = op2.Dat(nodes)
pl2 = op2.LegacyParloop(krhs, ele, b2(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
pl2()
print("err unordered", np.abs(b2.data_ro - ob).max())
