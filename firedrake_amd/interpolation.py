"""Interpolation (dual evaluation) parloops on the device -- SURVEY.md 8f rank 3.

``Function.interpolate(expr)`` in the reference compiles ``expr`` with TSFC's
``compile_expression_dual_evaluation`` and runs ONE parloop over the cells
(firedrake/interpolation.py:1087-1167):

    kernel(A, coords, w_0, ..., c_0, ...)      A: WRITE through the target's cell-node map,
                                               coords / coefficients: READ through theirs, constants: READ Globals

For point-evaluation (Lagrange-type) target elements the kernel evaluates the expression at the element's nodes.
TSFC/FInAT cannot run here, so this module restates that kernel for expressions given as C strings: for every
target node k it computes the physical point ``X = sum_v coords[v] * N_v(xi_k)`` (``N`` = coordinate-element basis
tabulated at the node's reference point, a static table like TSFC's), the coefficient values
``w<j> = sum_i w_j[i] * P^j_i(xi_k)`` and assigns ``A[k] = expr(X, w0, ..., c0, ...)``.  The parloop then runs through
the same wrapper generator as every other loop (indirect WRITE beside staged READ arguments since round 5: the coordinates and
coefficients are gathered through LDS, the lane writes the target in global memory; a node shared by several cells is
written with the same value by each of them, exactly as in the reference's sequential loop).  Keeping these loops on
the device means initial conditions and coefficient updates never round-trip through the host between assemblies.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Callable, Optional, Sequence

import numpy as np

from . import op2


# ---- reference elements: node points and tabulation -----------------------------------------------------------------
def simplex_node_points(dim, degree):
    """Reference points of the P1/P2 Lagrange nodes in mesh.py's node order (vertices, then edge midpoints)."""
    from .forms import _TET_EDGES, _TRI_EDGES
    v = np.concatenate([np.zeros((1, dim)), np.eye(dim)], axis=0)
    if degree == 1:
        return v
    if degree != 2:
        raise NotImplementedError("simplex Lagrange node points: degree 1 or 2")
    edges = _TET_EDGES if dim == 3 else _TRI_EDGES
    return np.concatenate([v, [(v[a] + v[b]) / 2 for a, b in edges]], axis=0)


def simplex_lagrange(dim, degree):
    """Tabulator of P<degree> on the reference simplex: points (n, dim) -> basis values (n, ndofs)."""
    from .forms import tabulate_lagrange
    return lambda pts: tabulate_lagrange(dim, degree, np.asarray(pts, dtype=np.float64))[0]


QUAD_VERTEX_POINTS = np.array([(0.0, 0.0), (0.0, 1.0), (1.0, 0.0), (1.0, 1.0)])     # vertex a*2 + b <-> (a, b)


def q1_quad(pts):
    """Bilinear basis on the reference quadrilateral in mesh.make_quad_mesh's vertex order a*2 + b."""
    pts = np.asarray(pts, dtype=np.float64)
    x, y = pts[:, 0], pts[:, 1]
    return np.stack([(1 - x) * (1 - y), (1 - x) * y, x * (1 - y), x * y], axis=1)


# ---- kernel generation -----------------------------------------------------------------------------------------------
def _table(name, a):
    a = np.asarray(a, dtype=np.float64)
    rows = ", ".join("{" + ", ".join(repr(float(x)) for x in r) + "}" for r in a)
    return f"  static const double {name}[{a.shape[0]}][{a.shape[1]}] = {{{rows}}};"


def dual_evaluation_kernel(name, exprs: Sequence[str], gdim: int, coord_table, coefficient_tables=(),
                           coefficient_value_sizes=(), n_constants=0):
    """The dual-evaluation kernel for a point-evaluation target element.

    ``exprs``: one C expression per component of the target's value (in terms of ``X[d]``, ``w<j>[c]``, ``c<j>[i]``);
    ``coord_table`` (nk, nv): coordinate basis at the target nodes; ``coefficient_tables[j]`` (nk, nd_j)."""
    coord_table = np.asarray(coord_table)
    nk, nv = coord_table.shape
    vs = len(exprs)
    args = ["double *A", "const double *coords"]
    args += [f"const double *w_{j}" for j in range(len(coefficient_tables))]
    args += [f"const double *c{j}" for j in range(n_constants)]
    body = [_table("N", coord_table)]
    for j, t in enumerate(coefficient_tables):
        body.append(_table(f"P{j}", t))
    body.append(f"  for (int k = 0; k < {nk}; ++k) {{")
    body.append(f"    double X[{gdim}];")
    body.append(f"    for (int d = 0; d < {gdim}; ++d) {{ X[d] = 0.0; for (int v = 0; v < {nv}; ++v) X[d] += N[k][v]*coords[v*{gdim} + d]; }}")
    for j, (t, s) in enumerate(zip(coefficient_tables, coefficient_value_sizes)):
        nd = np.asarray(t).shape[1]
        body.append(f"    double w{j}[{s}];")
        body.append(f"    for (int c = 0; c < {s}; ++c) {{ w{j}[c] = 0.0; for (int i = 0; i < {nd}; ++i) w{j}[c] += P{j}[k][i]*w_{j}[i*{s} + c]; }}")
    for c, e in enumerate(exprs):
        body.append(f"    A[k*{vs} + {c}] = {e};")
    body.append("  }")
    code = f"#include <math.h>\nstatic void {name}({', '.join(args)})\n{{\n" + "\n".join(body) + "\n}\n"
    return op2.Kernel(code, name, requires_zeroed_output_arguments=True)


@dataclass
class Space:
    """What the interpolator needs to know about a function space: its node Dat layout and reference element."""
    node_map: op2.Map                       # cell -> node map
    node_ref_points: np.ndarray             # (nk, tdim) reference points of the nodes (point-evaluation dual basis)
    tabulate: Optional[Callable] = None     # points -> (npts, ndofs) basis values (needed for sources only)
    value_size: int = 1


@dataclass
class Interpolator:
    """``Interpolator(exprs, target, coords_dat, coords).interpolate(out)`` mirrors firedrake's
    ``Interpolator.interpolate`` (interpolation.py:1000-1167) for C-string expressions."""
    exprs: Sequence[str]
    target: Space
    coordinates: op2.Dat
    coord_space: Space
    coefficients: Sequence = field(default_factory=tuple)       # [(Dat, Space)]
    constants: Sequence = field(default_factory=tuple)          # [Global]
    name: str = "expression_kernel"

    def __post_init__(self):
        pts = np.asarray(self.target.node_ref_points, dtype=np.float64)
        if len(self.exprs) != self.target.value_size:
            raise ValueError("one expression per component of the target space")
        gdim = self.coordinates.cdim
        self.kernel = dual_evaluation_kernel(
            self.name, self.exprs, gdim, self.coord_space.tabulate(pts),
            [sp.tabulate(pts) for _, sp in self.coefficients], [sp.value_size for _, sp in self.coefficients],
            len(self.constants))
        self._loops = {}

    def interpolate(self, out: op2.Dat):
        """Run the dual-evaluation parloop into ``out`` (WRITE)."""
        pl = self._loops.get(id(out))
        if pl is None or pl[0] is not out:
            tmap = self.target.node_map
            args = [out(op2.WRITE, tmap), self.coordinates(op2.READ, self.coord_space.node_map)]
            args += [d(op2.READ, sp.node_map) for d, sp in self.coefficients]
            args += [g(op2.READ) for g in self.constants]
            pl = (out, op2.LegacyParloop(self.kernel, tmap.iterset, *args))
            self._loops[id(out)] = pl
        pl[1]()
        return out
