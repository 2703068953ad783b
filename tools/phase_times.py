#!/usr/bin/env python
"""tools/phase_times.py [c2|c5] -- where a row block's lifetime goes (FDHIP_PHASE_TIMES=1: lane 0 of every block stores the 100 MHz
wall clock at its start, after the staging barrier, after the main loop's barrier and at its end).  Prints, for the Jacobian of the
C2 mesh (P1, whole-entity owner-computes-rows) or of the CG2 share (P2, row-sliced): mean / median microseconds per phase, the
blocks resident per CU over the kernel (sum of lifetimes / span / CUs) and the gap between a block's end and the start of the next
block on the same SIMD slot."""
import os
import sys

os.environ["FDHIP_PHASE_TIMES"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from firedrake_amd import _lib, forms, mesh as fmesh

which = sys.argv[1] if len(sys.argv) > 1 else "c2"
n, degree = (215, 1) if which == "c2" else (107, 2)
if len(sys.argv) > 2:
    n = int(sys.argv[2])
m = fmesh.UnitCubeMesh(n, degrees=(degree,), perturb=0.1, numbering="lexicographic")
prob = forms.PoissonProblem(m, degree, bcs=True)
for _ in range(4):
    prob.assemble_jacobian()
_lib.call("fd_device_sync")
loop = prob.jacobian()[1]
geo = next(g for key, g in loop._prepared["parts"].items() if key[0] == "ocr")
t = geo["phase_times"].download(np.int64, (geo["ocr"].nblocks, 5))
t0 = t[:, 0].min()
us = (t[:, :4] - t0) / 100.0
life = us[:, 3] - us[:, 0]
span = us[:, 3].max()
print(f"{which} n={n}: mode {geo['cw'].src.mode}, {len(t)} blocks, kernel span {span:.1f} us")
for name, a, b in (("stage", 0, 1), ("main loop", 1, 2), ("flush", 2, 3), ("lifetime", 0, 3)):
    d = us[:, b] - us[:, a]
    print(f"  {name:10s} mean {d.mean():7.2f} us  median {np.median(d):7.2f}  p10 {np.percentile(d, 10):7.2f}  p90 {np.percentile(d, 90):7.2f}")
print(f"  resident blocks per CU (sum of lifetimes / span / 256): {life.sum() / span / 256:.2f}")
# HW_ID: wave 0..3, simd 4..5, pipe 6..7, cu 8..11, sh 12, se 13..15 (+ XCC by the block's index modulo 8)
hw = t[:, 4]
nb_ = len(t)
q_, r_ = nb_ >> 3, nb_ & 7
starts = np.array([x * q_ + min(x, r_) for x in range(9)])                # fdw::xcd_block: XCD x owns blocks [starts[x], starts[x+1])
xcc = np.searchsorted(starts, np.arange(nb_), side="right") - 1
slot = xcc * (1 << 20) + (hw & 0xffff)
order = np.lexsort((us[:, 0], slot))
s_sorted, st, en = slot[order], us[order, 0], us[order, 3]
same = s_sorted[1:] == s_sorted[:-1]
gap = (st[1:] - en[:-1])[same]
gap = gap[gap > -1e-9]
if len(gap):
    print(f"  gap between consecutive blocks of one wavefront slot: mean {gap.mean():.2f} us, median {np.median(gap):.2f} ({len(gap)} pairs)")
