"""Hand-restated TSFC-equivalent local kernels for the benchmark forms (C strings).

In Firedrake these kernels are produced per form by TSFC (tsfc/driver.py:57-182 ->
tsfc/loopy.py:215-276) with the argument order of
tsfc/kernel_interface/firedrake_loopy.py:408-522:  ``A, coords, w_0, w_1, ...`` -- output
element tensor first (zeroed by the wrapper, tsfc_interface.py:330-331), then the coordinate
field, then the coefficients in form order; constant tables are ``static const`` arrays
(tsfc/loopy.py:237-244) and the affine Jacobian is unrolled (tsfc/fem.py:793-797).
TSFC/UFL/FIAT cannot be imported in this environment, so the kernels for the
five configurations are written out here in that shape.  They are ordinary C: the oracle
compiles the same string with gcc, the backend with hipcc.

Element-tensor parity with a real Firedrake install is *unpinned* (no reference test stores
element tensors); tests/test_forms_identities.py validates them with the analytic identities
the reference's regression tests rely on (constants in the stiffness null space, sum(M) = |Omega|,
A*u = action(a, u), exactness on polynomial data).
"""
from __future__ import annotations

import functools

import numpy as np

from . import op2


def _c(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 0:
        return repr(float(a))
    return "{" + ", ".join(_c(x) for x in a) + "}"


# ------------------------------------------------------------------------------------------
# quadrature + tabulation on the reference simplex (what FIAT/FInAT provide: tsfc/fem.py:330-333, 711-735)
# ------------------------------------------------------------------------------------------
def gauss_jacobi_simplex(dim, degree):
    """Collapsed (Stroud conical-product) Gauss-Jacobi rule exact for polynomials of ``degree``."""
    from scipy.special import roots_jacobi
    m = degree // 2 + 1
    if dim == 2:
        x0, w0 = roots_jacobi(m, 0, 0)
        x1, w1 = roots_jacobi(m, 1, 0)
        pts, wts = [], []
        for a, wa in zip(x1, w1):
            for b, wb in zip(x0, w0):
                s = (1 + a) / 2
                t = (1 - a) / 2 * (1 + b) / 2
                pts.append((s, t))
                wts.append(wa * wb / 8.0)
        return np.array(pts), np.array(wts)
    x0, w0 = roots_jacobi(m, 0, 0)
    x1, w1 = roots_jacobi(m, 1, 0)
    x2, w2 = roots_jacobi(m, 2, 0)
    pts, wts = [], []
    for a, wa in zip(x2, w2):
        for b, wb in zip(x1, w1):
            for c, wc in zip(x0, w0):
                r = (1 + a) / 2
                s = (1 - a) / 2 * (1 + b) / 2
                t = (1 - a) / 2 * (1 - b) / 2 * (1 + c) / 2
                pts.append((r, s, t))
                wts.append(wa * wb * wc / 64.0)
    return np.array(pts), np.array(wts)


def _orbit_points(dim, orbits):
    """Points of fully symmetric orbits on the reference simplex, given in barycentric form: ("v", a) = the dim + 1
    permutations of (1 - dim a, a, ..., a); ("e", b) = the six permutations of (b, b, 1/2 - b, 1/2 - b) (tetrahedron)."""
    import itertools
    pts = []
    for kind, a in orbits:
        bary = (1.0 - dim * a,) + (a,) * dim if kind == "v" else (a, a, 0.5 - a, 0.5 - a)
        pts.append(np.array(sorted(set(itertools.permutations(bary))))[:, 1:])
    return pts


# orbit parameters (a, weight) of the symmetric rules below: the solution of the moment equations (solve_symmetric_rule), written out
# so that every machine generates the same kernel text -- the JIT cache is keyed on it
_SYMMETRIC_RULES = {
    2: (("v", "v"), (0.09157621350977267, 0.05497587182766015, 0.44594849091596767, 0.11169079483900653)),
    3: (("v", "v", "e"), (0.09273525031089183, 0.012248840519393862, 0.31088591926330167, 0.018781320953003302,
                          0.04550370412564618, 0.0070910034628463275)),
}


def _symmetric_rule_from(dim, x):
    kinds = _SYMMETRIC_RULES[dim][0]
    orbits = _orbit_points(dim, [(k, x[2 * i]) for i, k in enumerate(kinds)])
    return np.concatenate(orbits), np.concatenate([np.full(len(o), x[2 * i + 1]) for i, o in enumerate(orbits)])


def _moment_residual(dim, deg, pts, wts):
    import math
    powers = [p for p in np.ndindex(*(deg + 1,) * dim) if sum(p) <= deg]
    exact = np.array([math.prod(math.factorial(e) for e in p) / math.factorial(sum(p) + dim) for p in powers])
    return np.array([(wts * np.prod(pts ** np.array(p), axis=1)).sum() for p in powers]) - exact


def solve_symmetric_rule(dim):
    """The orbit parameters of ``symmetric_simplex_rule`` from three-digit starting values: Gauss-Newton on the moment equations of
    degree 4 (triangle) / 5 (tetrahedron).  Kept as the derivation of ``_SYMMETRIC_RULES`` (tests/test_forms_identities.py)."""
    from scipy.optimize import least_squares
    deg = 4 if dim == 2 else 5
    x0 = (0.0916, 0.055, 0.4459, 0.1117) if dim == 2 else (0.0927, 0.0122, 0.3109, 0.0188, 0.0455, 0.0071)
    sol = least_squares(lambda x: _moment_residual(dim, deg, *_symmetric_rule_from(dim, x)), x0, xtol=1e-15, ftol=1e-15, gtol=1e-15)
    return sol.x


@functools.lru_cache(maxsize=None)
def symmetric_simplex_rule(dim):
    """Fully symmetric rule with positive weights and interior points: 6 points, degree 4 on the triangle (two vertex
    orbits); 14 points, degree 5 on the tetrahedron (two vertex orbits and one edge orbit).  Checked against the moment
    equations when first used."""
    deg = 4 if dim == 2 else 5
    pts, wts = _symmetric_rule_from(dim, _SYMMETRIC_RULES[dim][1])
    if np.abs(_moment_residual(dim, deg, pts, wts)).max() > 2e-15 or wts.min() <= 0 or pts.min() <= 0 or pts.sum(axis=1).max() >= 1:
        raise RuntimeError("symmetric simplex rule is not exact")
    return pts, wts


def simplex_rule(dim, degree):
    """Quadrature on the reference simplex the way FIAT's default scheme picks it (tsfc/fem.py:330-333 ->
    FIAT create_quadrature): small fully symmetric rules with positive weights at low degree, collapsed Gauss-Jacobi
    above.  (FIAT's own tables are not available offline; degree 4 takes the 6-point triangle rule and degrees 4-5 the 14-point
    degree-5 tetrahedron rule, which is not fewer points than FIAT's choice for these degrees.)"""
    if degree <= 1:
        return np.full((1, dim), 1.0 / (dim + 1)), np.array([1.0 / (2 if dim == 2 else 6)])
    if degree == 2 and dim == 2:
        return np.array([[1 / 6, 1 / 6], [2 / 3, 1 / 6], [1 / 6, 2 / 3]]), np.full(3, 1.0 / 6.0)
    if degree == 2 and dim == 3:
        a, b = 0.5854101966249685, 0.1381966011250105
        return np.array([[b, b, b], [a, b, b], [b, a, b], [b, b, a]]), np.full(4, 1.0 / 24.0)
    if degree == 4 or (dim == 3 and degree == 5):
        return symmetric_simplex_rule(dim)
    return gauss_jacobi_simplex(dim, degree)


_TET_EDGES = [(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]
_TRI_EDGES = [(1, 2), (0, 2), (0, 1)]


def ffc_rounding(table, epsilon=float(np.finfo(np.float64).resolution)):
    """TSFC's treatment of every tabulated table before it becomes kernel text (tsfc/fem.py:726, 758: ``ffc_rounding(table,
    ctx.epsilon)``, epsilon = finfo(scalar_type).resolution = 1e-15, tsfc/fem.py:87-88; the function itself lives in
    gem.optimise, FInAT's repository, not vendored with the reference): entries within epsilon of a one-decimal number (0, +-0.1,
    ..., +-1, ...) are snapped to it, minus zeros cleared.  A generated kernel therefore holds EXACT zeros where the tabulation
    leaves 1e-17, and the C compiler folds the terms they multiply."""
    table = np.asarray(table, dtype=np.float64)
    one_decimal = np.asarray(np.round(table, 1))
    one_decimal[np.logical_not(one_decimal)] = 0.0
    return np.where(np.abs(table - one_decimal) < epsilon, one_decimal, table)


def tabulate_lagrange(dim, degree, pts):
    """(phi[q][i], dphi[q][i][b]) of P1/P2 on the reference simplex; node order = vertices, then edges
    in the UFC/FIAT edge order (matches mesh.py's cell-node maps); rounded like TSFC rounds its tables (ffc_rounding)."""
    phi, dphi = _tabulate_lagrange(dim, degree, pts)
    return ffc_rounding(phi), ffc_rounding(dphi)


def _tabulate_lagrange(dim, degree, pts):
    pts = np.asarray(pts)
    lam = np.concatenate([1 - pts.sum(axis=1, keepdims=True), pts], axis=1)        # (nq, dim+1)
    dlam = np.concatenate([-np.ones((1, dim)), np.eye(dim)], axis=0)               # (dim+1, dim)
    nq = len(pts)
    if degree == 1:
        return lam.copy(), np.broadcast_to(dlam, (nq, dim + 1, dim)).copy()
    edges = _TET_EDGES if dim == 3 else _TRI_EDGES
    nd = dim + 1 + len(edges)
    phi = np.zeros((nq, nd))
    dphi = np.zeros((nq, nd, dim))
    for v in range(dim + 1):
        phi[:, v] = lam[:, v] * (2 * lam[:, v] - 1)
        dphi[:, v, :] = (4 * lam[:, v] - 1)[:, None] * dlam[v][None, :]
    for e, (a, b) in enumerate(edges):
        k = dim + 1 + e
        phi[:, k] = 4 * lam[:, a] * lam[:, b]
        dphi[:, k, :] = 4 * (lam[:, a][:, None] * dlam[b][None, :] + lam[:, b][:, None] * dlam[a][None, :])
    return phi, dphi


# ------------------------------------------------------------------------------------------
# geometry preamble shared by the affine-simplex kernels
# ------------------------------------------------------------------------------------------
_GEOM = {
    2: """
  const double J00 = x[2] - x[0], J01 = x[4] - x[0];
  const double J10 = x[3] - x[1], J11 = x[5] - x[1];
  const double det = J00*J11 - J01*J10;
  const double idet = 1.0 / det;
  const double K[2][2] = {{ J11*idet, -J01*idet}, {-J10*idet,  J00*idet}};
  const double adet = fabs(det);
""",
    3: """
  const double J00 = x[3] - x[0], J01 = x[6] - x[0], J02 = x[9]  - x[0];
  const double J10 = x[4] - x[1], J11 = x[7] - x[1], J12 = x[10] - x[1];
  const double J20 = x[5] - x[2], J21 = x[8] - x[2], J22 = x[11] - x[2];
  const double c00 = J11*J22 - J12*J21, c01 = J12*J20 - J10*J22, c02 = J10*J21 - J11*J20;
  const double det = J00*c00 + J01*c01 + J02*c02;
  const double idet = 1.0 / det;
  const double K[3][3] = {
    { c00*idet, (J02*J21 - J01*J22)*idet, (J01*J12 - J02*J11)*idet },
    { c01*idet, (J00*J22 - J02*J20)*idet, (J02*J10 - J00*J12)*idet },
    { c02*idet, (J01*J20 - J00*J21)*idet, (J00*J11 - J01*J10)*idet } };
  const double adet = fabs(det);
""",
}


def poisson_residual_kernel(dim, degree, name=None):
    """F(u; v) = int grad(u).grad(v) - f v dx   (configs C1/C2/C5).
    Arguments: A[nd], coords[(dim+1)*dim], u[nd], f[nd]."""
    name = name or f"poisson_p{degree}_{'tet' if dim == 3 else 'tri'}_residual"
    nv = dim + 1
    if degree == 1:
        # P1: gradients constant on the cell -> stiffness term outside the quadrature loop
        qp, qw = simplex_rule(dim, 2)
        phi, _ = tabulate_lagrange(dim, 1, qp)
        nq = len(qw)
        dl = np.concatenate([-np.ones((1, dim)), np.eye(dim)], axis=0)
        body = f"""
static void {name}(double *restrict A, const double *restrict x, const double *restrict u, const double *restrict f)
{{
  static const double DL[{nv}][{dim}] = {_c(dl)};
  static const double PHI[{nq}][{nv}] = {_c(phi)};
  static const double W[{nq}] = {_c(qw)};
{_GEOM[dim]}
  double g[{nv}][{dim}];
  for (int i = 0; i < {nv}; ++i)
    for (int a = 0; a < {dim}; ++a) {{
      double s = 0.0;
      for (int b = 0; b < {dim}; ++b) s += DL[i][b] * K[b][a];
      g[i][a] = s;
    }}
  double gu[{dim}];
  for (int a = 0; a < {dim}; ++a) {{
    double s = 0.0;
    for (int j = 0; j < {nv}; ++j) s += u[j] * g[j][a];
    gu[a] = s;
  }}
  const double vol = adet * {1.0 / (2 if dim == 2 else 6)!r};
  for (int i = 0; i < {nv}; ++i) {{
    double s = 0.0;
    for (int a = 0; a < {dim}; ++a) s += g[i][a] * gu[a];
    A[i] += vol * s;
  }}
  for (int q = 0; q < {nq}; ++q) {{
    double fq = 0.0;
    for (int j = 0; j < {nv}; ++j) fq += f[j] * PHI[q][j];
    const double wf = W[q] * adet * fq;
    for (int i = 0; i < {nv}; ++i) A[i] -= wf * PHI[q][i];
  }}
}}
"""
        return op2.Kernel(body, name, flop_count=None)
    # P2: two quadrature loops (degree 2 for the stiffness term, degree 4 for f*v)
    qs, ws = simplex_rule(dim, 2 * (degree - 1))
    _, dphs = tabulate_lagrange(dim, degree, qs)
    qm, wm = simplex_rule(dim, 2 * degree)
    phm, _ = tabulate_lagrange(dim, degree, qm)
    nd = phm.shape[1]
    body = f"""
static void {name}(double *restrict A, const double *restrict x, const double *restrict u, const double *restrict f)
{{
  static const double DPHI[{len(ws)}][{nd}][{dim}] = {_c(dphs)};
  static const double WS[{len(ws)}] = {_c(ws)};
  static const double PHI[{len(wm)}][{nd}] = {_c(phm)};
  static const double WM[{len(wm)}] = {_c(wm)};
{_GEOM[dim]}
  double G[{dim}][{dim}];              /* K K^T : reference-space metric */
  for (int a = 0; a < {dim}; ++a)
    for (int b = 0; b < {dim}; ++b) {{
      double s = 0.0;
      for (int c = 0; c < {dim}; ++c) s += K[a][c] * K[b][c];
      G[a][b] = s * adet;
    }}
  for (int q = 0; q < {len(ws)}; ++q) {{
    double gr[{dim}];
    for (int b = 0; b < {dim}; ++b) {{
      double s = 0.0;
      for (int j = 0; j < {nd}; ++j) s += u[j] * DPHI[q][j][b];
      gr[b] = s;
    }}
    double t[{dim}];
    for (int a = 0; a < {dim}; ++a) {{
      double s = 0.0;
      for (int b = 0; b < {dim}; ++b) s += G[a][b] * gr[b];
      t[a] = s * WS[q];
    }}
    for (int i = 0; i < {nd}; ++i) {{
      double s = 0.0;
      for (int a = 0; a < {dim}; ++a) s += DPHI[q][i][a] * t[a];
      A[i] += s;
    }}
  }}
  for (int q = 0; q < {len(wm)}; ++q) {{
    double fq = 0.0;
    for (int j = 0; j < {nd}; ++j) fq += f[j] * PHI[q][j];
    const double wf = WM[q] * adet * fq;
    for (int i = 0; i < {nd}; ++i) A[i] -= wf * PHI[q][i];
  }}
}}
"""
    return op2.Kernel(body, name)


def poisson_jacobian_kernel(dim, degree, name=None):
    """J(du, v) = int grad(du).grad(v) dx.  Arguments: A[nd*nd], coords."""
    name = name or f"poisson_p{degree}_{'tet' if dim == 3 else 'tri'}_jacobian"
    nv = dim + 1
    if degree == 1:
        dl = np.concatenate([-np.ones((1, dim)), np.eye(dim)], axis=0)
        body = f"""
static void {name}(double *restrict A, const double *restrict x)
{{
  static const double DL[{nv}][{dim}] = {_c(dl)};
{_GEOM[dim]}
  double g[{nv}][{dim}];
  for (int i = 0; i < {nv}; ++i)
    for (int a = 0; a < {dim}; ++a) {{
      double s = 0.0;
      for (int b = 0; b < {dim}; ++b) s += DL[i][b] * K[b][a];
      g[i][a] = s;
    }}
  const double vol = adet * {1.0 / (2 if dim == 2 else 6)!r};
  for (int i = 0; i < {nv}; ++i)
    for (int j = 0; j < {nv}; ++j) {{
      double s = 0.0;
      for (int a = 0; a < {dim}; ++a) s += g[i][a] * g[j][a];
      A[i*{nv} + j] += vol * s;
    }}
}}
"""
        return op2.Kernel(body, name)
    qs, ws = simplex_rule(dim, 2 * (degree - 1))
    _, dphs = tabulate_lagrange(dim, degree, qs)
    nd = dphs.shape[1]
    body = f"""
static void {name}(double *restrict A, const double *restrict x)
{{
  static const double DPHI[{len(ws)}][{nd}][{dim}] = {_c(dphs)};
  static const double WS[{len(ws)}] = {_c(ws)};
{_GEOM[dim]}
  double G[{dim}][{dim}];
  for (int a = 0; a < {dim}; ++a)
    for (int b = 0; b < {dim}; ++b) {{
      double s = 0.0;
      for (int c = 0; c < {dim}; ++c) s += K[a][c] * K[b][c];
      G[a][b] = s * adet;
    }}
  for (int q = 0; q < {len(ws)}; ++q)
    for (int i = 0; i < {nd}; ++i) {{
      double t[{dim}];
      for (int a = 0; a < {dim}; ++a) {{
        double s = 0.0;
        for (int b = 0; b < {dim}; ++b) s += G[a][b] * DPHI[q][i][b];
        t[a] = s * WS[q];
      }}
      for (int j = 0; j < {nd}; ++j) {{
        double s = 0.0;
        for (int a = 0; a < {dim}; ++a) s += t[a] * DPHI[q][j][a];
        A[i*{nd} + j] += s;
      }}
    }}
}}
"""
    return op2.Kernel(body, name)


def mass_kernel(dim, degree, name=None):
    """a(u, v) = int u v dx.  Arguments: A[nd*nd], coords."""
    name = name or f"mass_p{degree}_{'tet' if dim == 3 else 'tri'}"
    qm, wm = simplex_rule(dim, 2 * degree)
    phm, _ = tabulate_lagrange(dim, degree, qm)
    nd = phm.shape[1]
    body = f"""
static void {name}(double *restrict A, const double *restrict x)
{{
  static const double PHI[{len(wm)}][{nd}] = {_c(phm)};
  static const double WM[{len(wm)}] = {_c(wm)};
{_GEOM[dim]}
  for (int q = 0; q < {len(wm)}; ++q)
    for (int i = 0; i < {nd}; ++i)
      for (int j = 0; j < {nd}; ++j)
        A[i*{nd} + j] += WM[q] * adet * PHI[q][i] * PHI[q][j];
}}
"""
    return op2.Kernel(body, name)


def helmholtz_kernel(dim, degree, name=None):
    """a(u, v) = int grad(u).grad(v) + u v dx in ONE local kernel, the way TSFC emits a form with two terms under one
    measure (tests/firedrake/regression/test_helmholtz.py:44): the stiffness and the mass quadrature loops of
    poisson_jacobian_kernel / mass_kernel, one after the other.  Arguments: A[nd*nd], coords."""
    name = name or f"helmholtz_p{degree}_{'tet' if dim == 3 else 'tri'}"
    kj, km = poisson_jacobian_kernel(dim, degree, name + "_stiffness"), mass_kernel(dim, degree, name + "_mass")
    body = kj.code + km.code + f"""
static void {name}(double *restrict A, const double *restrict x)
{{
  {name}_stiffness(A, x);
  {name}_mass(A, x);
}}
"""
    return op2.Kernel(body, name)


def rhs_kernel(dim, degree, name=None):
    """L(v) = int f v dx with f in the same space (test_helmholtz.py:45).  Arguments: b[nd], coords, f[nd]."""
    name = name or f"rhs_p{degree}_{'tet' if dim == 3 else 'tri'}"
    qm, wm = simplex_rule(dim, 2 * degree)
    phm, _ = tabulate_lagrange(dim, degree, qm)
    nd = phm.shape[1]
    body = f"""
static void {name}(double *restrict b, const double *restrict x, const double *restrict f)
{{
  static const double PHI[{len(wm)}][{nd}] = {_c(phm)};
  static const double WM[{len(wm)}] = {_c(wm)};
{_GEOM[dim]}
  for (int q = 0; q < {len(wm)}; ++q) {{
    double fq = 0.0;
    for (int j = 0; j < {nd}; ++j) fq += f[j] * PHI[q][j];
    const double wf = WM[q] * adet * fq;
    for (int i = 0; i < {nd}; ++i) b[i] += wf * PHI[q][i];
  }}
}}
"""
    return op2.Kernel(body, name)


# ------------------------------------------------------------------------------------------
# assemble()-shaped front end for these forms
# ------------------------------------------------------------------------------------------
class PoissonProblem:
    """The Newton-step assembly of configs C1/C2/C5: residual F(u) and Jacobian J on a CG space.

    Plays the role of OneFormAssembler / ExplicitMatrixAssembler (firedrake/assemble.py:1197-1293,
    1344-1558): parloops are built once and cached, ``assemble_residual`` zeroes the tensor, runs
    the cell parloop inside ``frozen_halo(INC)`` (assemble.py:1281-1286) and applies BCs
    (assemble.py:1243-1267); ``assemble_jacobian`` zeroes the CSR, scatters with BC-masked lgmaps
    (assemble.py:2075-2108) and sets the BC diagonal (assemble.py:1501-1507).
    """

    def __init__(self, mesh, degree=1, bcs=True, seed=0, bc_nodes=None):
        """``bcs``: Dirichlet conditions on the whole boundary; ``bc_nodes`` (local node numbers) restricts them to a part of
        it, like ``DirichletBC(V, g, sub_domain)`` (bcs.py:245-330)."""
        self.mesh = mesh
        self.degree = degree
        self.V = V = mesh.space(degree)
        dim = mesh.gdim
        rng = np.random.default_rng(seed + 17 * V.halo.rank if V.halo else seed)
        pts = V.node_points
        # deterministic fields: a smooth "state" plus the Helmholtz-test forcing (test_helmholtz.py:36-39)
        uvals = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + (0.3 * pts[:, 2] if dim == 3 else 0.0)
        fvals = (1 + 8 * np.pi ** 2) * np.cos(2 * np.pi * pts[:, 0]) * np.cos(2 * np.pi * pts[:, 1])
        self.u = V.dat(1, uvals, "u")
        self.f = V.dat(1, fvals, "f")
        self.r = V.dat(1, None, "residual")
        self.bc_nodes = (V.boundary_nodes if bc_nodes is None else np.asarray(bc_nodes, dtype=np.int32)) if bcs else np.zeros(0, dtype=np.int32)
        self._bc_all = self.bc_nodes                          # incl. ghost nodes: their columns are masked too
        self.bc_nodes = self.bc_nodes[self.bc_nodes < V.node_set.size]
        self.kres = poisson_residual_kernel(dim, degree)
        self.kjac = poisson_jacobian_kernel(dim, degree)
        cm, xm = V.cell_node_map, mesh.coord_space.cell_node_map
        self.res_loop = op2.LegacyParloop(self.kres, mesh.cell_set, self.r(op2.INC, cm), mesh.coordinates(op2.READ, xm),
                                          self.u(op2.READ, cm), self.f(op2.READ, cm))
        self._jac = None
        self._bc_dev = None
        del rng

    # -- residual ---------------------------------------------------------------------
    def _bc_rows(self):
        if self._bc_dev is None:
            from .device import DeviceBuffer
            self._bc_dev = DeviceBuffer.from_numpy(np.ascontiguousarray(self.bc_nodes, dtype=np.int32))
        return self._bc_dev

    def assemble_residual(self, events=None):
        """``events``: optional (before, after) device events recorded around the cell parloop alone (bench.py's
        per-kernel roofline; the zeroing pass and the BC fix-up stay outside that bracket)."""
        import ctypes
        from . import _lib
        self.r.zero()                                   # a13: zeroing is part of every assemble
        with self.r.frozen_halo(op2.INC):
            if events:
                events[0].record()
            self.res_loop()
            if events:
                events[1].record()
        if len(self.bc_nodes):                          # a14: bc.zero(tensor)  (bcs.py:192-221)
            _lib.call("fd_dat_set_rows", self.r._dev_ptr(True), 1, self._bc_rows().ptr, len(self.bc_nodes),
                      ctypes.c_double(0.0), None)
        return self.r

    # -- Jacobian ---------------------------------------------------------------------
    def jacobian(self):
        if self._jac is None:
            V = self.V
            cm, xm = V.cell_node_map, self.mesh.coord_space.cell_node_map
            sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
            mat = op2.Mat(sp)
            lg = None
            partitioned = V.node_set.total_size > V.node_set.size
            if len(self.bc_nodes) or partitioned:
                rlg = np.arange(V.node_set.total_size, dtype=np.int32)
                clg = rlg.copy()
                rlg[self.bc_nodes] = -1                 # functionspaceimpl.py:913-926
                clg[self.bc_nodes] = -1
                if partitioned:
                    rlg[V.node_set.size:] = -1          # rows owned elsewhere are assembled by their owner
                    gb = self._bc_all[self._bc_all >= V.node_set.size]
                    clg[gb] = -1                        # BC columns on ghost nodes are dropped too
                lg = (rlg, clg)
            loop = op2.LegacyParloop(self.kjac, self.mesh.cell_set, mat(op2.INC, (cm, cm), lgmaps=lg),
                                     self.mesh.coordinates(op2.READ, xm))
            loop.compute_ghost = partitioned
            self._jac = (mat, loop)
        return self._jac

    def assemble_jacobian(self, events=None):
        import ctypes
        from . import _lib
        mat, loop = self.jacobian()
        mat.zero()
        if events:
            events[0].record()
        loop()
        if events:
            events[1].record()
        if len(self.bc_nodes):
            mat.set_diagonal_rows(self._bc_rows(), len(self.bc_nodes), 1.0)
        return mat


# the geometry-specific wrapper variants the benchmark configurations end up launching (bench.py, FDHIP_DEBUG=1 prints them): compiled
# ahead of time by precompile_all() so that a fresh machine's first assemble does not start with hipcc (0.35 s per wrapper).  A
# variant missing here is not an error -- it is compiled on first use -- and bench.py reports how many were (``jit_compiles``).
BENCH_VARIANTS = {
    ("residual", 2, 1): ("staged_s304",),
    ("jacobian", 2, 1): ("ocrpm_q10k3d", "ocrpm_q9k3d", "ocrpm_q8k3d", "ocrpm_q7k3d", "ocrpm_q6k3d", "ocrp_q10k3d_fx", "ocr_q10k3d_fx", "ocrp_q9k3d_fx", "ocr_q9k3d_fx"),
    ("residual", 3, 1): ("stagedo_s432", "staged_s416"),
    ("jacobian", 3, 1): ("ocrpm_q10k4d", "ocrpm_q9k4d", "ocrp_q10k4d_fx", "ocr_q10k4d_fx", "ocrp_q9k4d_fx", "ocr_q9k4d_fx"),
    ("residual", 3, 2): ("stagedo_s1520x256", "stagedo_s1504x256"),       # n = 107 (one GPU's share), n = 215 (the whole cube of configs[4])
    # (two rows per instance since round 6: "_g" + the row pairs Parloop._ocrs_geometry picks on the benchmark meshes)
    ("jacobian", 3, 2): ("ocrspr_g0619283745_q8k7e13", "ocrspr_q8k7e13", "ocrs_q8k7e13", "ocrspr", "ocrsp", "ocrs"),
    "dg_advection": ("staged_d0_s576", "staged_s1024x528", "staged_s2336x672"),
}


def precompile_all():
    """Compile (hipcc, disk-cached) the wrappers of the benchmark forms so the code objects ship with the tree: the base wrapper
    shapes and the geometry-specific variants of BENCH_VARIANTS, each through ``GlobalKernel.compile`` (so the unroll / occupancy
    retries a first call would make are made here)."""
    from .codegen import ocr_eligible, sliced_eligible, staged_eligible
    from .configuration import configuration
    from .kernel import DatKernelArg, GlobalKernel, MapKernelArg, MatKernelArg
    from .op2types import INC, READ
    jobs = []

    def build(g, modes):
        jobs.extend((g, mode) for mode in modes)

    def run(job):
        g, mode = job
        try:
            return g.compile(mode).path
        except Exception as exc:                         # (a variant name that no longer fits this kernel's staged maps)
            import sys
            print(f"[fdhip] precompile {g.name} {mode}: {exc}", file=sys.stderr)
            return None

    for dim, degree in ((2, 1), (3, 1), (3, 2)):
        nd = {(2, 1): 3, (3, 1): 4, (3, 2): 10}[(dim, degree)]
        cm = MapKernelArg(nd)
        xm = cm if degree == 1 else MapKernelArg(dim + 1)
        f64 = np.dtype("float64")
        kres = poisson_residual_kernel(dim, degree).with_signature([INC, READ, READ, READ], [f64] * 4)
        gk = GlobalKernel(kres, [DatKernelArg((1,), cm), DatKernelArg((dim,), xm), DatKernelArg((1,), cm), DatKernelArg((1,), cm)])
        kjac = poisson_jacobian_kernel(dim, degree).with_signature([INC, READ], [f64] * 2)
        build(gk, [m for m in ("staged", "direct") if m != "staged" or staged_eligible(gk)] + list(BENCH_VARIANTS[("residual", dim, degree)]))
        for lg in (False, True):
            gj = GlobalKernel(kjac, [MatKernelArg(((1,), (1,)), (cm, cm), lgmaps=lg), DatKernelArg((dim,), xm)])
            modes = ["direct"] + (["staged"] if staged_eligible(gj) else []) + (["ocr"] if ocr_eligible(gj) else []) \
                + (["ocrs"] if (ocr_eligible(gj) and sliced_eligible(gj)) else [])
            extra = [v for v in BENCH_VARIANTS[("jacobian", dim, degree)] if v.startswith("ocrs") == (ocr_eligible(gj) and sliced_eligible(gj))]
            # the default accumulates in fp64; the opt-in fixed-point variants ("_fx") are built too: bench.py times them beside
            extra = extra + [v[:-3] for v in extra if v.endswith("_fx")]
            build(gj, modes + (extra if lg else []))
    # config C3: the tensor-product wrappers of the Q4 Helmholtz operator (matrix with and without BC lgmaps, action), also with
    # coefficient arguments
    from . import mesh as fmesh
    hm = fmesh.make_extruded_hex_mesh(1, 1, 4, perturb=0.0)
    for bcs in (False, True):
        for prob in (HelmholtzQ4Problem(hm, bcs=bcs), CoefficientHexProblem(hm, bcs=bcs, nq=5)):
            for loop in (prob.jac_loop, prob.act_loop):
                build(loop.global_kernel, [None])
    # the wider descriptors of the bench line: a coefficient gradient (Q3) and a vector-valued space ((Q2)^3)
    for prob in (NonlinearDiffusionHexProblem(fmesh.make_extruded_hex_mesh(1, 1, 3, perturb=0.0), bcs=True),
                 ElasticityHexProblem(fmesh.make_extruded_hex_mesh(1, 1, 2, perturb=0.0), bcs=True)):
        for loop in (prob.jac_loop, prob.act_loop):
            build(loop.global_kernel, [None])
    # ... and the Helmholtz operator on Q5, Q6, Q7 (column-chunked MFMA panels; the bench line's high-order entries)
    for degree, nq in ((5, 6), (6, 8), (7, 9)):
        prob = HelmholtzHexProblem(fmesh.make_extruded_hex_mesh(1, 1, degree, perturb=0.0), bcs=True, nq=nq)
        for loop in (prob.jac_loop, prob.act_loop):
            build(loop.global_kernel, [None])
    # config C4: the three DG advection loops
    qm = fmesh.make_quad_mesh(4, perturb=0.1)
    for loop, variant in zip(DGAdvectionProblem(qm).loops, BENCH_VARIANTS["dg_advection"]):
        build(loop.global_kernel, [None, variant])
    # hipcc is a subprocess: the compiles run side by side (a fresh tree: ~70 code objects, seconds each)
    import os
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=max(1, min(16, len(os.sched_getaffinity(0))))) as pool:
        return [p for p in pool.map(run, jobs) if p]


# ------------------------------------------------------------------------------------------
# Config C3: Helmholtz on Q4 hexahedra (extruded) -- dense quadrature contraction, fp64 MFMA kernel
# ------------------------------------------------------------------------------------------
def q4_tables(degree=4, nq=5):
    """1-D tables of CG_k on GLL nodes at Gauss-Legendre points: (L[q][i], DL[q][i], points, weights) on [0,1]."""
    from .tensor import gll_gauss_tables
    return gll_gauss_tables(degree, nq)


def helmholtz_hex_jacobian_kernel(degree=4, nq=None, name=None, alpha=1.0, beta=1.0, velocity=(0.0, 0.0, 0.0)):
    """a(u, v) = int alpha grad(u).grad(v) + (b.grad(u)) v + beta u v dx on a trilinear hexahedron with the Q_degree basis and nq^3
    Gauss points (default nq = degree + 1: dx(degree=2*degree)); alpha = beta = 1, b = 0 is the Helmholtz operator of
    config C3, (0, 1) the mass form, a non-zero ``velocity`` b a (non-symmetric) convection-diffusion-reaction operator.
    Arguments: A[nd*nd] (row = test function), coords[8*3] (Q1 vertices, index a*4 + b*2 + c), nd = (degree+1)^3.
    Dense formulation (what the MFMA kernel computes); used as the oracle's local kernel."""
    k1 = degree + 1
    nq = nq or k1
    nd = k1 ** 3
    name = name or f"helmholtz_q{degree}_hex_jacobian"
    L, DL, qp, qw = q4_tables(degree, nq)
    body = f"""
static void {name}(double *restrict A, const double *restrict x)
{{
  static const double L[{nq}][{k1}] = {_c(L)};
  static const double DL[{nq}][{k1}] = {_c(DL)};
  static const double QP[{nq}] = {_c(qp)};
  static const double QW[{nq}] = {_c(qw)};
  for (int q1 = 0; q1 < {nq}; ++q1) for (int q2 = 0; q2 < {nq}; ++q2) for (int q3 = 0; q3 < {nq}; ++q3) {{
    const double t[3] = {{QP[q1], QP[q2], QP[q3]}};
    double J[3][3] = {{{{0,0,0}},{{0,0,0}},{{0,0,0}}}};
    for (int v = 0; v < 8; ++v) {{
      const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
      const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
      const double da = a ? 1.0 : -1.0, db = b ? 1.0 : -1.0, dc = c ? 1.0 : -1.0;
      const double g[3] = {{da*Nb*Nc, Na*db*Nc, Na*Nb*dc}};
      for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) J[r][s] += x[3*v + r] * g[s];
    }}
    const double c00 = J[1][1]*J[2][2] - J[1][2]*J[2][1], c01 = J[1][2]*J[2][0] - J[1][0]*J[2][2], c02 = J[1][0]*J[2][1] - J[1][1]*J[2][0];
    const double det = J[0][0]*c00 + J[0][1]*c01 + J[0][2]*c02, id = 1.0 / det;
    const double K[3][3] = {{
      {{ c00*id, (J[0][2]*J[2][1] - J[0][1]*J[2][2])*id, (J[0][1]*J[1][2] - J[0][2]*J[1][1])*id }},
      {{ c01*id, (J[0][0]*J[2][2] - J[0][2]*J[2][0])*id, (J[0][2]*J[1][0] - J[0][0]*J[1][2])*id }},
      {{ c02*id, (J[0][1]*J[2][0] - J[0][0]*J[2][1])*id, (J[0][0]*J[1][1] - J[0][1]*J[1][0])*id }} }};
    const double w = QW[q1]*QW[q2]*QW[q3]*fabs(det);
    double G[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
      G[a][b] = w * (K[a][0]*K[b][0] + K[a][1]*K[b][1] + K[a][2]*K[b][2]);
    const double bv[3] = {{{float(velocity[0])!r}, {float(velocity[1])!r}, {float(velocity[2])!r}}};
    double cb[3];                       /* K b: the velocity in reference coordinates */
    for (int a = 0; a < 3; ++a) cb[a] = K[a][0]*bv[0] + K[a][1]*bv[1] + K[a][2]*bv[2];
    double ph[{nd}], dp[{nd}][3];
    for (int i1 = 0; i1 < {k1}; ++i1) for (int i2 = 0; i2 < {k1}; ++i2) for (int i3 = 0; i3 < {k1}; ++i3) {{
      const int i = (i1*{k1} + i2)*{k1} + i3;
      ph[i] = L[q1][i1]*L[q2][i2]*L[q3][i3];
      dp[i][0] = DL[q1][i1]*L[q2][i2]*L[q3][i3];
      dp[i][1] = L[q1][i1]*DL[q2][i2]*L[q3][i3];
      dp[i][2] = L[q1][i1]*L[q2][i2]*DL[q3][i3];
    }}
    for (int i = 0; i < {nd}; ++i) {{
      const double t0 = G[0][0]*dp[i][0] + G[0][1]*dp[i][1] + G[0][2]*dp[i][2];
      const double t1 = G[1][0]*dp[i][0] + G[1][1]*dp[i][1] + G[1][2]*dp[i][2];
      const double t2 = G[2][0]*dp[i][0] + G[2][1]*dp[i][1] + G[2][2]*dp[i][2];
      const double tm = w * ph[i];
      for (int j = 0; j < {nd}; ++j)
        A[i*{nd} + j] += {float(alpha)!r}*(t0*dp[j][0] + t1*dp[j][1] + t2*dp[j][2]) + {float(beta)!r}*tm*ph[j]
                         + tm*(cb[0]*dp[j][0] + cb[1]*dp[j][1] + cb[2]*dp[j][2]);
    }}
  }}
}}
"""
    from .kernel import TensorProductLocalKernel
    from .tensor import second_order_weights
    return TensorProductLocalKernel(body, name, kind="matrix", degree=degree, nq=nq, weights_code=second_order_weights(name, alpha, beta, velocity))


def helmholtz_hex_action_kernel(degree=4, nq=None, name=None, alpha=1.0, beta=1.0, velocity=(0.0, 0.0, 0.0)):
    """y += A_e(coords) u: the action of the same bilinear form on a coefficient (the matrix-free operator application of
    tests/firedrake/regression/test_matrix_free.py, and the Q4 "residual/action" of SURVEY.md 8d).  Arguments: y[nd],
    coords[24], u[nd].  The C text is the dense definition (element matrix times element vector) the oracle executes; the
    backend evaluates it sum-factorised from the descriptor (csrc/fd_tensor.h: hex_qk_action)."""
    from .kernel import TensorProductLocalKernel
    from .tensor import second_order_weights
    nq = nq or degree + 1
    nd = (degree + 1) ** 3
    name = name or f"helmholtz_q{degree}_hex_action"
    jac = helmholtz_hex_jacobian_kernel(degree, nq, name + "_matrix", alpha, beta, velocity)
    body = jac.code + f"""
static void {name}(double *restrict y, const double *restrict x, const double *restrict u)
{{
  static double A[{nd}*{nd}];
  for (int q = 0; q < {nd}*{nd}; ++q) A[q] = 0.0;
  {name}_matrix(A, x);
  for (int i = 0; i < {nd}; ++i) {{
    double s = 0.0;
    for (int j = 0; j < {nd}; ++j) s += A[i*{nd} + j] * u[j];
    y[i] += s;
  }}
}}
"""
    return TensorProductLocalKernel(body, name, kind="action", degree=degree, nq=nq, weights_code=second_order_weights(name, alpha, beta, velocity))


def helmholtz_q4_hex_jacobian_kernel(name="helmholtz_q4_hex_jacobian", alpha=1.0, beta=1.0):
    """Config C3: Q4, 5 x 5 x 5 Gauss points (dx(degree=8))."""
    return helmholtz_hex_jacobian_kernel(4, 5, name, alpha, beta)


def helmholtz_q4_hex_action_kernel(name="helmholtz_q4_hex_action", alpha=1.0, beta=1.0):
    return helmholtz_hex_action_kernel(4, 5, name, alpha, beta)


class HelmholtzHexProblem:
    """Config C3 (Q4) and its siblings: the Q_k Helmholtz operator on an extruded hex mesh, assembled and applied through ordinary parloops
    (``op2.LegacyParloop`` with a Mat / Dat argument over the extruded cell set, optional BC lgmaps) whose local kernels
    are TensorProductLocalKernels: ``GlobalKernel.compile`` picks the fp64-MFMA matrix wrapper and the sum-factorised
    action wrapper of csrc/fd_tensor.h.  Plays ExplicitMatrixAssembler / OneFormAssembler like PoissonProblem."""

    def __init__(self, hexmesh, bcs=False, nq=None, alpha=1.0, beta=1.0, velocity=(0.0, 0.0, 0.0), matrix=True):
        """``matrix=False``: the matrix-free side only (no Sparsity, no Mat: the action of a mesh whose matrix would not fit the 32-bit
        CSR index range -- Q4 at n = 64 holds 3.7e9 nonzeros)."""
        self.mesh = m = hexmesh
        nd, nqp = (m.degree + 1) ** 3, (nq or m.degree + 1) ** 3
        pad = -(-nd // 16) * 16
        self.FLOPS_PER_CELL = 2.0 * pad * pad * 4 * nqp       # MFMA work issued (element matrix padded to 16 x 16 tiles)
        self.ALGO_FLOPS_PER_CELL = 2.0 * nd * nd * 4 * nqp    # 15.6 MFLOP per Q4 cell
        cm, xm = m.cell_node_map, m.coord_map
        if matrix:
            self.sparsity = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, None)])
            self.mat = op2.Mat(self.sparsity)
        pts = m.node_points
        bnd = np.nonzero(((pts < 1e-12) | (pts > 1 - 1e-12)).any(axis=1))[0].astype(np.int32)
        self.bc_nodes = bnd if bcs else np.zeros(0, dtype=np.int32)
        lg = None
        if len(self.bc_nodes):
            rlg = np.arange(m.node_set.total_size, dtype=np.int32)
            rlg[self.bc_nodes] = -1
            lg = (rlg, rlg.copy())
        self.kjac = helmholtz_hex_jacobian_kernel(m.degree, nq, None, alpha, beta, velocity)
        self.kact = helmholtz_hex_action_kernel(m.degree, nq, None, alpha, beta, velocity)
        if matrix:
            self.jac_loop = op2.LegacyParloop(self.kjac, m.cell_set, self.mat(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm))
        self.u = op2.Dat(m.node_set, np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.3 * pts[:, 2], np.float64, "u")
        self.y = op2.Dat(m.node_set, None, np.float64, "y")
        self.act_loop = op2.LegacyParloop(self.kact, m.cell_set, self.y(op2.INC, cm), m.coordinates(op2.READ, xm), self.u(op2.READ, cm))
        self._bc_dev = None

    def _bc_rows(self):
        if self._bc_dev is None:
            from .device import DeviceBuffer
            self._bc_dev = DeviceBuffer.from_numpy(np.ascontiguousarray(self.bc_nodes, dtype=np.int32))
        return self._bc_dev

    def assemble_jacobian(self, events=None):
        import ctypes
        from . import _lib
        self.mat.zero()                       # a13: part of every assemble (the scatter adds into zeroed values)
        if events:
            self.jac_loop.zero_ahead()        # perform the zeroing outside the kernel bracket
            events[0].record()
        self.jac_loop()
        if events:
            events[1].record()
        if len(self.bc_nodes):
            self.mat.set_diagonal_rows(self._bc_rows(), len(self.bc_nodes), 1.0)
        return self.mat

    def assemble_action(self, events=None):
        """y = A u  (zero + one INC parloop inside frozen_halo, like a 1-form assembly)."""
        self.y.zero()
        with self.y.frozen_halo(op2.INC):
            if events:
                events[0].record()
            self.act_loop()
            if events:
                events[1].record()
        return self.y


def coefficient_hex_jacobian_kernel(degree=4, nq=None, name=None, kappa="1.0 + C[0]", react="1.0 + C[1]*C[1]", ncoef=2, spaces=None):
    """a(du, v) = int kappa grad(du).grad(v) + react du v dx on a trilinear hexahedron, Q_degree basis, nq^3 Gauss points, with
    ``kappa`` / ``react`` C expressions in the values C[0..ncoef) of ``ncoef`` coefficient fields at the point and the physical
    point X[0..2] -- a variable-coefficient operator, or the Jacobian of a residual with a nonlinear reaction term (react = g'(u0)).
    Arguments in TSFC's order (tsfc/kernel_interface/firedrake_loopy.py:432-522): A[nd*nd], coords[24], w_0[nd] ... w_{ncoef-1}[nd].
    ``spaces``: per coefficient "k" (a Q_degree field: nd values per cell, the default) or "1" (a field in the Q1 space of the
    coordinates: 8 vertex values per cell, interpolated trilinearly).
    The C text is the dense definition the oracle and the direct wrapper execute; the descriptor lets the backend evaluate the
    coefficients sum-factorised and contract on the fp64 matrix cores (csrc/fd_tensor.h)."""
    from .kernel import TensorProductLocalKernel
    from .tensor import coefficient_weights
    k1 = degree + 1
    nq = nq or k1
    nd = k1 ** 3
    name = name or f"coefficient_q{degree}_hex_jacobian"
    L, DL, qp, qw = q4_tables(degree, nq)
    spaces = list(spaces) if spaces is not None else ["k"] * ncoef
    wargs = "".join(f", const double *restrict w{m}" for m in range(ncoef))
    cvals = "\n".join(f"    for (int i = 0; i < {nd}; ++i) C[{m}] += ph[i] * w{m}[i];" if spaces[m] == "k" else
                      f"    for (int v = 0; v < 8; ++v) C[{m}] += NV[v] * w{m}[v];" for m in range(ncoef))
    body = f"""
static void {name}(double *restrict A, const double *restrict x{wargs})
{{
  static const double L[{nq}][{k1}] = {_c(L)};
  static const double DL[{nq}][{k1}] = {_c(DL)};
  static const double QP[{nq}] = {_c(qp)};
  static const double QW[{nq}] = {_c(qw)};
  for (int q1 = 0; q1 < {nq}; ++q1) for (int q2 = 0; q2 < {nq}; ++q2) for (int q3 = 0; q3 < {nq}; ++q3) {{
    const double t[3] = {{QP[q1], QP[q2], QP[q3]}};
    double J[3][3] = {{{{0,0,0}},{{0,0,0}},{{0,0,0}}}}, X[3] = {{0,0,0}}, NV[8];
    for (int v = 0; v < 8; ++v) {{
      const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
      const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
      const double da = a ? 1.0 : -1.0, db = b ? 1.0 : -1.0, dc = c ? 1.0 : -1.0;
      const double g[3] = {{da*Nb*Nc, Na*db*Nc, Na*Nb*dc}};
      NV[v] = Na*Nb*Nc;
      for (int r = 0; r < 3; ++r) {{ for (int s = 0; s < 3; ++s) J[r][s] += x[3*v + r] * g[s]; X[r] += x[3*v + r] * Na*Nb*Nc; }}
    }}
    const double c00 = J[1][1]*J[2][2] - J[1][2]*J[2][1], c01 = J[1][2]*J[2][0] - J[1][0]*J[2][2], c02 = J[1][0]*J[2][1] - J[1][1]*J[2][0];
    const double det = J[0][0]*c00 + J[0][1]*c01 + J[0][2]*c02, id = 1.0 / det;
    const double K[3][3] = {{
      {{ c00*id, (J[0][2]*J[2][1] - J[0][1]*J[2][2])*id, (J[0][1]*J[1][2] - J[0][2]*J[1][1])*id }},
      {{ c01*id, (J[0][0]*J[2][2] - J[0][2]*J[2][0])*id, (J[0][2]*J[1][0] - J[0][0]*J[1][2])*id }},
      {{ c02*id, (J[0][1]*J[2][0] - J[0][0]*J[2][1])*id, (J[0][0]*J[1][1] - J[0][1]*J[1][0])*id }} }};
    const double w = QW[q1]*QW[q2]*QW[q3]*fabs(det);
    double ph[{nd}], dp[{nd}][3];
    for (int i1 = 0; i1 < {k1}; ++i1) for (int i2 = 0; i2 < {k1}; ++i2) for (int i3 = 0; i3 < {k1}; ++i3) {{
      const int i = (i1*{k1} + i2)*{k1} + i3;
      ph[i] = L[q1][i1]*L[q2][i2]*L[q3][i3];
      dp[i][0] = DL[q1][i1]*L[q2][i2]*L[q3][i3];
      dp[i][1] = L[q1][i1]*DL[q2][i2]*L[q3][i3];
      dp[i][2] = L[q1][i1]*L[q2][i2]*DL[q3][i3];
    }}
    double C[{max(ncoef, 1)}] = {{0}};
{cvals}
    const double kappa = ({kappa}), react = ({react});
    double G[3][3];
    for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
      G[a][b] = kappa * w * (K[a][0]*K[b][0] + K[a][1]*K[b][1] + K[a][2]*K[b][2]);
    for (int i = 0; i < {nd}; ++i) {{
      const double t0 = G[0][0]*dp[i][0] + G[0][1]*dp[i][1] + G[0][2]*dp[i][2];
      const double t1 = G[1][0]*dp[i][0] + G[1][1]*dp[i][1] + G[1][2]*dp[i][2];
      const double t2 = G[2][0]*dp[i][0] + G[2][1]*dp[i][1] + G[2][2]*dp[i][2];
      const double tm = react * w * ph[i];
      for (int j = 0; j < {nd}; ++j)
        A[i*{nd} + j] += t0*dp[j][0] + t1*dp[j][1] + t2*dp[j][2] + tm*ph[j];
    }}
    (void)X; (void)NV;
  }}
}}
"""
    return TensorProductLocalKernel(body, name, kind="matrix", degree=degree, nq=nq, ncoef=ncoef,
                                    weights_code=coefficient_weights(name, kappa, react))


def coefficient_hex_action_kernel(degree=4, nq=None, name=None, kappa="1.0 + C[0]", react="1.0 + C[1]*C[1]", ncoef=2, spaces=None):
    """y += A_e(coords, w_0 ...) u for the same form.  Arguments: y[nd], coords[24], u[nd], w_0[nd] ...; dense C text for the
    oracle, sum-factorised on the device (the coefficients ride through the same axis passes as u)."""
    from .kernel import TensorProductLocalKernel
    from .tensor import coefficient_weights
    nq = nq or degree + 1
    nd = (degree + 1) ** 3
    name = name or f"coefficient_q{degree}_hex_action"
    jac = coefficient_hex_jacobian_kernel(degree, nq, name + "_matrix", kappa, react, ncoef, spaces)
    wargs = "".join(f", const double *restrict w{m}" for m in range(ncoef))
    wpass = "".join(f", w{m}" for m in range(ncoef))
    body = jac.code + f"""
static void {name}(double *restrict y, const double *restrict x, const double *restrict u{wargs})
{{
  static double A[{nd}*{nd}];
  for (int q = 0; q < {nd}*{nd}; ++q) A[q] = 0.0;
  {name}_matrix(A, x{wpass});
  for (int i = 0; i < {nd}; ++i) {{
    double s = 0.0;
    for (int j = 0; j < {nd}; ++j) s += A[i*{nd} + j] * u[j];
    y[i] += s;
  }}
}}
"""
    return TensorProductLocalKernel(body, name, kind="action", degree=degree, nq=nq, ncoef=ncoef,
                                    weights_code=coefficient_weights(name, kappa, react))


class CoefficientHexProblem(HelmholtzHexProblem):
    """a(du, v) = int kappa(w0) grad(du).grad(v) + c(u0) du v dx on extruded Q_k hexahedra: a variable diffusivity field w0 and the
    linearisation point u0 of a nonlinear reaction term enter the matrix and the action as coefficient arguments -- the shape of
    the Jacobians TSFC generates for nonlinear / variable-coefficient problems (tsfc/kernel_interface/firedrake_loopy.py:432-522,
    coefficient evaluation tsfc/fem.py:742-805).  Default: kappa = 1 + w0, c = 1 + u0^2 (the Jacobian of u + u^3/3)."""

    def __init__(self, hexmesh, bcs=False, nq=None, kappa="1.0 + C[0]", react="1.0 + C[1]*C[1]", q1_diffusivity=False):
        """``q1_diffusivity``: w0 lives in the Q1 space of the coordinates (8 vertex values per cell on the coordinate map) instead
        of the Q_k space of the unknown -- a piecewise-trilinear material field under a high-order discretisation."""
        super().__init__(hexmesh, bcs, nq)
        m = hexmesh
        cm, xm = m.cell_node_map, m.coord_map
        pts = m.node_points
        if q1_diffusivity:
            xp = np.asarray(m.coordinates.data_ro_with_halos)
            self.w0 = op2.Dat(m.coordinates.dataset.set, 0.5 + 0.4 * np.sin(2 * xp[:, 0] + xp[:, 1]) * np.cos(xp[:, 2]), np.float64, "kappa_q1")
        else:
            self.w0 = op2.Dat(m.node_set, 0.5 + 0.4 * np.sin(2 * pts[:, 0] + pts[:, 1]) * np.cos(pts[:, 2]), np.float64, "kappa_field")
        self.u0 = op2.Dat(m.node_set, np.cos(2 * pts[:, 0]) * np.sin(3 * pts[:, 1] + 1.0) + 0.2 * pts[:, 2], np.float64, "u0")
        lg = self.jac_loop.arguments[0].lgmaps
        spaces = ("1", "k") if q1_diffusivity else ("k", "k")
        tag = "q1coef" if q1_diffusivity else "coefficient"
        self.kjac = coefficient_hex_jacobian_kernel(m.degree, nq, f"{tag}_q{m.degree}_hex_jacobian", kappa, react, 2, spaces)
        self.kact = coefficient_hex_action_kernel(m.degree, nq, f"{tag}_q{m.degree}_hex_action", kappa, react, 2, spaces)
        wm = xm if q1_diffusivity else cm
        self.jac_loop = op2.LegacyParloop(self.kjac, m.cell_set, self.mat(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm),
                                          self.w0(op2.READ, wm), self.u0(op2.READ, cm))
        self.act_loop = op2.LegacyParloop(self.kact, m.cell_set, self.y(op2.INC, cm), m.coordinates(op2.READ, xm), self.u(op2.READ, cm),
                                          self.w0(op2.READ, wm), self.u0(op2.READ, cm))


_HEX_POINT_GEOMETRY = """
    const double t[3] = {QP[q1], QP[q2], QP[q3]};
    double J[3][3] = {{0,0,0},{0,0,0},{0,0,0}}, NV[8], DNV[8][3];
    for (int v = 0; v < 8; ++v) {
      const int a = v >> 2, b = (v >> 1) & 1, c = v & 1;
      const double Na = a ? t[0] : 1.0 - t[0], Nb = b ? t[1] : 1.0 - t[1], Nc = c ? t[2] : 1.0 - t[2];
      const double da = a ? 1.0 : -1.0, db = b ? 1.0 : -1.0, dc = c ? 1.0 : -1.0;
      NV[v] = Na*Nb*Nc; DNV[v][0] = da*Nb*Nc; DNV[v][1] = Na*db*Nc; DNV[v][2] = Na*Nb*dc;
      for (int r = 0; r < 3; ++r) for (int s = 0; s < 3; ++s) J[r][s] += x[3*v + r] * DNV[v][s];
    }
    const double c00 = J[1][1]*J[2][2] - J[1][2]*J[2][1], c01 = J[1][2]*J[2][0] - J[1][0]*J[2][2], c02 = J[1][0]*J[2][1] - J[1][1]*J[2][0];
    const double det = J[0][0]*c00 + J[0][1]*c01 + J[0][2]*c02, id = 1.0 / det;
    const double K[3][3] = {
      { c00*id, (J[0][2]*J[2][1] - J[0][1]*J[2][2])*id, (J[0][1]*J[1][2] - J[0][2]*J[1][1])*id },
      { c01*id, (J[0][0]*J[2][2] - J[0][2]*J[2][0])*id, (J[0][2]*J[1][0] - J[0][0]*J[1][2])*id },
      { c02*id, (J[0][1]*J[2][0] - J[0][0]*J[2][1])*id, (J[0][0]*J[1][1] - J[0][1]*J[1][0])*id } };
    const double w = QW[q1]*QW[q2]*QW[q3]*fabs(det);
    double ph[ND], dx[ND][3];                     /* basis values and PHYSICAL gradients at the point */
    for (int i1 = 0; i1 < K1; ++i1) for (int i2 = 0; i2 < K1; ++i2) for (int i3 = 0; i3 < K1; ++i3) {
      const int i = (i1*K1 + i2)*K1 + i3;
      const double dr[3] = {DL[q1][i1]*L[q2][i2]*L[q3][i3], L[q1][i1]*DL[q2][i2]*L[q3][i3], L[q1][i1]*L[q2][i2]*DL[q3][i3]};
      ph[i] = L[q1][i1]*L[q2][i2]*L[q3][i3];
      for (int s = 0; s < 3; ++s) dx[i][s] = K[0][s]*dr[0] + K[1][s]*dr[1] + K[2][s]*dr[2];
    }
"""


def _hex_dense_head(name, args, degree, nq):
    k1 = degree + 1
    L, DL, qp, qw = q4_tables(degree, nq)
    return f"""
static void {name}({args})
{{
  enum {{ K1 = {k1}, ND = {k1 ** 3} }};
  static const double L[{nq}][{k1}] = {_c(L)};
  static const double DL[{nq}][{k1}] = {_c(DL)};
  static const double QP[{nq}] = {_c(qp)};
  static const double QW[{nq}] = {_c(qw)};
  for (int q1 = 0; q1 < {nq}; ++q1) for (int q2 = 0; q2 < {nq}; ++q2) for (int q3 = 0; q3 < {nq}; ++q3) {{
{_HEX_POINT_GEOMETRY}"""


def _dense_action(name, nrow, extra_params, extra_pass):
    """y += A_e u on top of the dense matrix kernel ``<name>_matrix``: what the oracle executes for an action descriptor"""
    return f"""
static void {name}(double *restrict y, const double *restrict x, const double *restrict u{extra_params})
{{
  static double A[{nrow}*{nrow}];
  for (int q = 0; q < {nrow}*{nrow}; ++q) A[q] = 0.0;
  {name}_matrix(A, x{extra_pass});
  for (int i = 0; i < {nrow}; ++i) {{
    double s = 0.0;
    for (int j = 0; j < {nrow}; ++j) s += A[i*{nrow} + j] * u[j];
    y[i] += s;
  }}
}}
"""


def nonlinear_diffusion_hex_jacobian_kernel(degree=2, nq=None, name=None, space="k"):
    """The Newton Jacobian of F(u; v) = int (1 + |grad u|^2) grad(u).grad(v) dx at u0 on a trilinear hexahedron, Q_degree basis:
        a(du, v) = int (1 + |grad u0|^2) grad(du).grad(v) + 2 (grad u0 . grad du)(grad u0 . grad v) dx
    -- a form whose point weight needs the GRADIENT of a coefficient at the Gauss points (tsfc/fem.py:742-805).  Arguments in TSFC's
    order: A[nd*nd], coords[24], u0[nd] (``space`` "k") or u0[8] ("1": a field in the Q1 space of the coordinates).  Dense C text for
    the oracle and the direct wrapper; descriptor with ``coef_gradients`` for the tensor-product wrappers."""
    from .kernel import TensorProductLocalKernel
    from .tensor import nonlinear_diffusion_weights
    nq = nq or degree + 1
    name = name or f"nonlinear_diffusion_q{degree}_hex_jacobian"
    grad = ("for (int i = 0; i < ND; ++i) for (int s = 0; s < 3; ++s) g[s] += dx[i][s] * w0[i];" if space == "k" else
            "for (int v = 0; v < 8; ++v) for (int s = 0; s < 3; ++s) g[s] += (K[0][s]*DNV[v][0] + K[1][s]*DNV[v][1] + K[2][s]*DNV[v][2]) * w0[v];")
    body = _hex_dense_head(name, "double *restrict A, const double *restrict x, const double *restrict w0", degree, nq) + f"""
    double g[3] = {{0, 0, 0}};
    {grad}
    const double kappa = 1.0 + g[0]*g[0] + g[1]*g[1] + g[2]*g[2];
    for (int i = 0; i < ND; ++i) {{
      const double gi = g[0]*dx[i][0] + g[1]*dx[i][1] + g[2]*dx[i][2];
      for (int j = 0; j < ND; ++j)
        A[i*ND + j] += w * (kappa * (dx[i][0]*dx[j][0] + dx[i][1]*dx[j][1] + dx[i][2]*dx[j][2])
                            + 2.0 * gi * (g[0]*dx[j][0] + g[1]*dx[j][1] + g[2]*dx[j][2]));
    }}
    (void)NV; (void)ph;
  }}
}}
"""
    return TensorProductLocalKernel(body, name, kind="matrix", degree=degree, nq=nq, ncoef=1, coef_gradients=True,
                                    weights_code=nonlinear_diffusion_weights(name))


def nonlinear_diffusion_hex_action_kernel(degree=2, nq=None, name=None, space="k"):
    """y += A_e(coords, u0) du for the same Jacobian (the matrix-free Newton operator).  Arguments: y[nd], coords[24], du[nd], u0."""
    from .kernel import TensorProductLocalKernel
    from .tensor import nonlinear_diffusion_weights
    nq = nq or degree + 1
    name = name or f"nonlinear_diffusion_q{degree}_hex_action"
    jac = nonlinear_diffusion_hex_jacobian_kernel(degree, nq, name + "_matrix", space)
    body = jac.code + _dense_action(name, (degree + 1) ** 3, ", const double *restrict w0", ", w0")
    return TensorProductLocalKernel(body, name, kind="action", degree=degree, nq=nq, ncoef=1, coef_gradients=True,
                                    weights_code=nonlinear_diffusion_weights(name))


def elasticity_hex_jacobian_kernel(degree=2, nq=None, name=None, mu=1.0, lam=1.25, rho=0.0):
    """a(u, v) = int 2 mu eps(u):eps(v) + lam div(u) div(v) + rho u.v dx on (Q_degree)^3 over a trilinear hexahedron: a VECTOR-VALUED
    space -- Mat dims (3, 3), element tensor A[(i*3 + p)*(3 nd) + j*3 + r] (builder.py:573-625, MatSetValuesBlockedLocal).  Arguments:
    A[(3 nd)^2], coords[24].  Dense C text for the oracle; descriptor with ``vdim`` = 3 for the tensor-product wrappers."""
    from .kernel import TensorProductLocalKernel
    from .tensor import elasticity_weights
    nq = nq or degree + 1
    name = name or f"elasticity_q{degree}_hex_jacobian"
    body = _hex_dense_head(name, "double *restrict A, const double *restrict x", degree, nq) + f"""
    for (int i = 0; i < ND; ++i) for (int p = 0; p < 3; ++p) for (int j = 0; j < ND; ++j) for (int r = 0; r < 3; ++r) {{
      double v = {float(mu)!r} * dx[i][r] * dx[j][p] + {float(lam)!r} * dx[i][p] * dx[j][r];
      if (p == r) v += {float(mu)!r} * (dx[i][0]*dx[j][0] + dx[i][1]*dx[j][1] + dx[i][2]*dx[j][2]) + {float(rho)!r} * ph[i] * ph[j];
      A[(i*3 + p)*(3*ND) + j*3 + r] += w * v;
    }}
    (void)NV;
  }}
}}
"""
    return TensorProductLocalKernel(body, name, kind="matrix", degree=degree, nq=nq, vdim=3, weights_code=elasticity_weights(name, mu, lam, rho))


def elasticity_hex_action_kernel(degree=2, nq=None, name=None, mu=1.0, lam=1.25, rho=0.0):
    """y += A_e(coords) u for the same form: y, u Dats of dim 3 on the Q_degree map (y[i*3 + p])."""
    from .kernel import TensorProductLocalKernel
    from .tensor import elasticity_weights
    nq = nq or degree + 1
    name = name or f"elasticity_q{degree}_hex_action"
    jac = elasticity_hex_jacobian_kernel(degree, nq, name + "_matrix", mu, lam, rho)
    body = jac.code + _dense_action(name, 3 * (degree + 1) ** 3, "", "")
    return TensorProductLocalKernel(body, name, kind="action", degree=degree, nq=nq, vdim=3, weights_code=elasticity_weights(name, mu, lam, rho))


class NonlinearDiffusionHexProblem(HelmholtzHexProblem):
    """The Newton Jacobian of int (1 + |grad u|^2) grad(u).grad(v) dx at u0 on extruded Q_k hexahedra, as a matrix and matrix-free: the
    linearisation point enters through its gradient at the Gauss points (TensorProductLocalKernel ``coef_gradients``).
    ``q1_state``: u0 in the Q1 space of the coordinates instead of the Q_k space of the unknown."""

    def __init__(self, hexmesh, bcs=False, nq=None, q1_state=False):
        super().__init__(hexmesh, bcs, nq)
        m = hexmesh
        cm, xm = m.cell_node_map, m.coord_map
        if q1_state:
            xp = np.asarray(m.coordinates.data_ro_with_halos)
            self.u0 = op2.Dat(m.coordinates.dataset.set, np.cos(2 * xp[:, 0]) * np.sin(3 * xp[:, 1] + 1.0) + 0.2 * xp[:, 2] ** 2, np.float64, "u0_q1")
        else:
            pts = m.node_points
            self.u0 = op2.Dat(m.node_set, np.cos(2 * pts[:, 0]) * np.sin(3 * pts[:, 1] + 1.0) + 0.2 * pts[:, 2] ** 2, np.float64, "u0")
        lg = self.jac_loop.arguments[0].lgmaps
        space, tag = ("1", "q1state_") if q1_state else ("k", "")
        self.kjac = nonlinear_diffusion_hex_jacobian_kernel(m.degree, nq, f"nonlinear_diffusion_{tag}q{m.degree}_hex_jacobian", space)
        self.kact = nonlinear_diffusion_hex_action_kernel(m.degree, nq, f"nonlinear_diffusion_{tag}q{m.degree}_hex_action", space)
        wm = xm if q1_state else cm
        self.jac_loop = op2.LegacyParloop(self.kjac, m.cell_set, self.mat(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm),
                                          self.u0(op2.READ, wm))
        self.act_loop = op2.LegacyParloop(self.kact, m.cell_set, self.y(op2.INC, cm), m.coordinates(op2.READ, xm), self.u(op2.READ, cm),
                                          self.u0(op2.READ, wm))


class ElasticityHexProblem:
    """Linear elasticity on (Q_k)^3 over an extruded hex mesh: a vector-valued tensor-product space -- Mat with dims (3, 3), Dats of dim
    3 -- through the same two wrappers (fp64-MFMA matrix, one scalar contraction per component pair; sum-factorised action)."""

    def __init__(self, hexmesh, bcs=False, nq=None, mu=1.0, lam=1.25, rho=0.0):
        self.mesh = m = hexmesh
        nd, nqp = (m.degree + 1) ** 3, (nq or m.degree + 1) ** 3
        pad = -(-nd // 16) * 16
        self.FLOPS_PER_CELL = 9 * 2.0 * pad * pad * 4 * nqp       # MFMA work issued: nine padded scalar blocks
        self.ALGO_FLOPS_PER_CELL = 9 * 2.0 * nd * nd * 4 * nqp
        cm, xm = m.cell_node_map, m.coord_map
        vset = m.node_set ** 3
        self.sparsity = op2.Sparsity((vset, vset), [(cm, cm, None)])
        self.mat = op2.Mat(self.sparsity)
        pts = m.node_points
        bnd = np.nonzero(((pts < 1e-12) | (pts > 1 - 1e-12)).any(axis=1))[0].astype(np.int32)
        self.bc_nodes = bnd if bcs else np.zeros(0, dtype=np.int32)
        lg = None
        if len(self.bc_nodes):
            rlg = np.arange(m.node_set.total_size, dtype=np.int32)
            rlg[self.bc_nodes] = -1
            lg = (rlg, rlg.copy())
        self.kjac = elasticity_hex_jacobian_kernel(m.degree, nq, None, mu, lam, rho)
        self.kact = elasticity_hex_action_kernel(m.degree, nq, None, mu, lam, rho)
        self.jac_loop = op2.LegacyParloop(self.kjac, m.cell_set, self.mat(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm))
        uv = np.stack([np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.3 * pts[:, 2], pts[:, 0] * pts[:, 1] - 0.5 * pts[:, 2] ** 2,
                       np.cos(pts[:, 0] + 2 * pts[:, 2])], axis=1)
        self.u = op2.Dat(vset, uv, np.float64, "u")
        self.y = op2.Dat(vset, None, np.float64, "y")
        self.act_loop = op2.LegacyParloop(self.kact, m.cell_set, self.y(op2.INC, cm), m.coordinates(op2.READ, xm), self.u(op2.READ, cm))

    def assemble_jacobian(self, events=None):
        self.mat.zero()
        if events:
            self.jac_loop.zero_ahead()
            events[0].record()
        self.jac_loop()
        if events:
            events[1].record()
        return self.mat

    def assemble_action(self, events=None):
        self.y.zero()
        with self.y.frozen_halo(op2.INC):
            if events:
                events[0].record()
            self.act_loop()
            if events:
                events[1].record()
        return self.y


class HelmholtzQ4Problem(HelmholtzHexProblem):
    """BASELINE.json configs[2]: Q4, 5 x 5 x 5 Gauss points."""

    def __init__(self, hexmesh, bcs=False, matrix=True):
        assert hexmesh.degree == 4
        super().__init__(hexmesh, bcs, 5, matrix=matrix)


# ------------------------------------------------------------------------------------------
# Config C4: DG advection (demos/DG_advection/DG_advection.py.rst, form L1) -- cell + ds + dS kernels
# ------------------------------------------------------------------------------------------
_QUAD_GEOM = """
  /* bilinear geometry at reference point (s, t): vertex v = a*2 + b at (a, b) */
  #define FD_QGEOM(xc, s, t) \\
    const double N[4] = {(1-(s))*(1-(t)), (1-(s))*(t), (s)*(1-(t)), (s)*(t)}; \\
    const double dNs[4] = {-(1-(t)), -(t), (1-(t)), (t)}; \\
    const double dNt[4] = {-(1-(s)), (1-(s)), -(s), (s)}; \\
    double J00 = 0, J01 = 0, J10 = 0, J11 = 0; \\
    for (int v = 0; v < 4; ++v) { J00 += xc[2*v]*dNs[v]; J01 += xc[2*v]*dNt[v]; J10 += xc[2*v+1]*dNs[v]; J11 += xc[2*v+1]*dNt[v]; } \\
    const double det = J00*J11 - J01*J10, idet = 1.0/det; \\
    const double K00 = J11*idet, K01 = -J01*idet, K10 = -J10*idet, K11 = J00*idet;
"""


def dg_advection_kernels(nq=3):
    """(cell, exterior-facet, interior-facet) kernels of
        L1 = dtc*( q div(phi u) dx - [u.n<0] phi u.n q_in ds - [u.n>0] phi u.n q ds
                   - (phi('+') - phi('-'))*(un('+') q('+') - un('-') q('-')) dS ),   un = (u.n + |u.n|)/2
    with the TSFC argument order  A, coords, coefficients in form order (q, u), constants (dtc, q_in),
    facet numbers (firedrake_loopy.py:432-522).  DQ1 basis = bilinear shape functions, vertex a*2+b."""
    from numpy.polynomial import legendre as leg
    x, w = leg.leggauss(nq)
    x, w = 0.5 * (x + 1.0), 0.5 * w
    tab = f"static const double QP[{nq}] = {_c(x)}; static const double QW[{nq}] = {_c(w)};"
    cell = f"""
{_QUAD_GEOM}
static void dg_adv_cell(double *restrict A, const double *restrict xc, const double *restrict qd, const double *restrict ud,
                        const double *restrict dtc)
{{
  {tab}
  for (int a = 0; a < {nq}; ++a) for (int b = 0; b < {nq}; ++b) {{
    const double s = QP[a], t = QP[b];
    FD_QGEOM(xc, s, t)
    double qq = 0, u0 = 0, u1 = 0, divu = 0;
    double gx[4], gy[4];
    for (int v = 0; v < 4; ++v) {{
      gx[v] = dNs[v]*K00 + dNt[v]*K10;  gy[v] = dNs[v]*K01 + dNt[v]*K11;
      qq += qd[v]*N[v]; u0 += ud[2*v]*N[v]; u1 += ud[2*v+1]*N[v];
      divu += ud[2*v]*gx[v] + ud[2*v+1]*gy[v];
    }}
    const double wq = QW[a]*QW[b]*fabs(det)*dtc[0]*qq;
    for (int i = 0; i < 4; ++i) A[i] += wq * (gx[i]*u0 + gy[i]*u1 + N[i]*divu);
  }}
}}
"""
    facet_pt = """
    /* reference point of facet f at parameter p, outward reference normal (n0, n1), tangent axis */
    const double s = (f == 0) ? 0.0 : (f == 1) ? 1.0 : p;
    const double t = (f == 2) ? 0.0 : (f == 3) ? 1.0 : p;
    const double rn0 = (f == 0) ? -1.0 : (f == 1) ? 1.0 : 0.0;
    const double rn1 = (f == 2) ? -1.0 : (f == 3) ? 1.0 : 0.0;
"""
    ext = f"""
{_QUAD_GEOM}
static void dg_adv_ext(double *restrict A, const double *restrict xc, const double *restrict qd, const double *restrict ud,
                       const double *restrict dtc, const double *restrict qin, const unsigned int *restrict facet)
{{
  {tab}
  const unsigned int f = facet[0];
  for (int a = 0; a < {nq}; ++a) {{
    const double p = QP[a];
{facet_pt}
    FD_QGEOM(xc, s, t)
    double nx = K00*rn0 + K10*rn1, ny = K01*rn0 + K11*rn1;
    const double nn = sqrt(nx*nx + ny*ny); nx /= nn; ny /= nn;
    const double tx = (f < 2) ? J01 : J00, ty = (f < 2) ? J11 : J10;
    const double ds = sqrt(tx*tx + ty*ty);
    double qq = 0, u0 = 0, u1 = 0;
    for (int v = 0; v < 4; ++v) {{ qq += qd[v]*N[v]; u0 += ud[2*v]*N[v]; u1 += ud[2*v+1]*N[v]; }}
    const double udn = u0*nx + u1*ny;
    const double flux = (udn < 0.0) ? udn*qin[0] : ((udn > 0.0) ? udn*qq : 0.0);
    const double wq = -dtc[0]*QW[a]*ds*flux;
    for (int i = 0; i < 4; ++i) A[i] += wq * N[i];
  }}
}}
"""
    intf = f"""
{_QUAD_GEOM}
static void dg_adv_int(double *restrict A, const double *restrict xc, const double *restrict qd, const double *restrict ud,
                       const double *restrict dtc, const unsigned int *restrict facet)
{{
  {tab}
  for (int a = 0; a < {nq}; ++a) {{
    const double p = QP[a];
    double Np[4], Nm[4], nx, ny, ds;
    {{ const unsigned int f = facet[0];
{facet_pt}
      FD_QGEOM(xc, s, t)
      nx = K00*rn0 + K10*rn1; ny = K01*rn0 + K11*rn1;
      const double nn = sqrt(nx*nx + ny*ny); nx /= nn; ny /= nn;
      const double tx = (f < 2) ? J01 : J00, ty = (f < 2) ? J11 : J10;
      ds = sqrt(tx*tx + ty*ty);
      for (int v = 0; v < 4; ++v) Np[v] = N[v]; }}
    {{ const unsigned int f = facet[1];
{facet_pt}
      const double N[4] = {{(1-s)*(1-t), (1-s)*t, s*(1-t), s*t}};
      (void)rn0; (void)rn1;
      for (int v = 0; v < 4; ++v) Nm[v] = N[v]; }}
    double qp = 0, qm = 0, u0 = 0, u1 = 0;
    for (int v = 0; v < 4; ++v) {{ qp += qd[v]*Np[v]; qm += qd[4+v]*Nm[v]; u0 += ud[2*v]*Np[v]; u1 += ud[2*v+1]*Np[v]; }}
    const double udn = u0*nx + u1*ny;
    const double unp = 0.5*(udn + fabs(udn)), unm = 0.5*(-udn + fabs(udn));
    const double wq = -dtc[0]*QW[a]*ds*(unp*qp - unm*qm);
    for (int i = 0; i < 4; ++i) {{ A[i] += wq*Np[i]; A[4+i] -= wq*Nm[i]; }}
  }}
}}
"""
    return op2.Kernel(cell, "dg_adv_cell"), op2.Kernel(ext, "dg_adv_ext"), op2.Kernel(intf, "dg_adv_int")


class DGAdvectionProblem:
    """RHS assembly of the DG-advection demo: one 1-form with cell + exterior-facet + interior-facet
    integrals = three parloops into the same Dat inside frozen_halo(INC) (assemble.py:1281-1286;
    SURVEY.md 3.3).  The interior-facet loop has arity-8 maps and a direct (2,)-uint32 facet-number Dat."""

    def __init__(self, qmesh, dt=None):
        import math
        self.mesh = m = qmesh
        pts = m.dq_points
        xq1 = np.array(m.coordinates.data_ro)
        self.u = op2.Dat(m.q1_set ** 2, np.stack([0.5 - xq1[:, 1], xq1[:, 0] - 0.5], axis=1), np.float64, "velocity")
        r_bell = np.minimum(np.sqrt((pts[:, 0] - 0.25) ** 2 + (pts[:, 1] - 0.5) ** 2) / 0.15, 1.0)
        r_cone = np.minimum(np.sqrt((pts[:, 0] - 0.5) ** 2 + (pts[:, 1] - 0.25) ** 2) / 0.15, 1.0)
        self.q = op2.Dat(m.dq_set, 1.0 + 0.25 * (1 + np.cos(math.pi * r_bell)) + 1.0 - r_cone, np.float64, "q")
        self._interp = None
        self.L = op2.Dat(m.dq_set, None, np.float64, "L1")
        self.dtc = op2.Global(1, dt if dt is not None else 2 * math.pi / 600.0, np.float64, "dtc")
        self.q_in = op2.Global(1, 1.0, np.float64, "q_in")
        kc, ke, ki = dg_advection_kernels()
        L, q, u, x = self.L, self.q, self.u, m.coordinates
        self.loops = [
            op2.LegacyParloop(kc, m.cell_set, L(op2.INC, m.cell_dq), x(op2.READ, m.cell_q1), q(op2.READ, m.cell_dq),
                              u(op2.READ, m.cell_q1), self.dtc(op2.READ)),
            op2.LegacyParloop(ke, m.ext_facet_set, L(op2.INC, m.ext_dq), x(op2.READ, m.ext_q1), q(op2.READ, m.ext_dq),
                              u(op2.READ, m.ext_q1), self.dtc(op2.READ), self.q_in(op2.READ), m.ext_local_facet(op2.READ)),
            op2.LegacyParloop(ki, m.int_facet_set, L(op2.INC, m.int_dq), x(op2.READ, m.int_q1), q(op2.READ, m.int_dq),
                              u(op2.READ, m.int_q1), self.dtc(op2.READ), m.int_local_facet(op2.READ)),
        ]

    def assemble_rhs(self):
        self.L.zero()
        with self.L.frozen_halo(op2.INC):
            for loop in self.loops:
                loop()
        return self.L

    # the demo's own statements ``u = Function(W).interpolate(velocity)`` and ``q = Function(V).interpolate(1.0 + bell +
    # cone + slot_cyl)`` (demos/DG_advection/DG_advection.py.rst:128-156) as dual-evaluation parloops on the device
    DEMO_VELOCITY = ("0.5 - X[1]", "X[0] - 0.5")
    DEMO_INITIAL_CONDITION = (
        "1.0 + 0.25*(1.0 + cos(M_PI*fmin(sqrt(pow(X[0] - 0.25, 2) + pow(X[1] - 0.5, 2))/0.15, 1.0)))"
        " + (1.0 - fmin(sqrt(pow(X[0] - 0.5, 2) + pow(X[1] - 0.25, 2))/0.15, 1.0))"
        " + ((sqrt(pow(X[0] - 0.5, 2) + pow(X[1] - 0.75, 2)) < 0.15)"
        " ? (((X[0] > 0.475 && X[0] < 0.525) && X[1] < 0.85) ? 0.0 : 1.0) : 0.0)",)

    def interpolate_demo_fields(self):
        """Set ``u`` and ``q`` to the demo's velocity and bell + cone + slotted-cylinder state on the device."""
        from .interpolation import QUAD_VERTEX_POINTS, Interpolator, Space, q1_quad
        m = self.mesh
        if self._interp is None:
            xs = Space(m.cell_q1, QUAD_VERTEX_POINTS, q1_quad, 2)
            self._interp = (
                Interpolator(self.DEMO_VELOCITY, Space(m.cell_q1, QUAD_VERTEX_POINTS, q1_quad, 2), m.coordinates, xs,
                             name="interpolate_velocity"),
                Interpolator(self.DEMO_INITIAL_CONDITION, Space(m.cell_dq, QUAD_VERTEX_POINTS, q1_quad, 1), m.coordinates, xs,
                             name="interpolate_q0"))
        self._interp[0].interpolate(self.u)
        self._interp[1].interpolate(self.q)


def dg_mass_solve_kernel(nq=2):
    """dq = M_e^{-1} L_e on one DQ1 cell (the DG mass matrix is block diagonal, so the demo's
    ``LinearVariationalSolver(a, L)`` with bjacobi/ilu is an exact cell-local solve).
    Arguments: dq[4] (WRITE), coords[8], L[4]."""
    from numpy.polynomial import legendre as leg
    x, w = leg.leggauss(nq)
    x, w = 0.5 * (x + 1.0), 0.5 * w
    body = f"""
{_QUAD_GEOM}
static void dg_mass_solve(double *restrict dq, const double *restrict xc, const double *restrict L)
{{
  static const double QP[{nq}] = {_c(x)}; static const double QW[{nq}] = {_c(w)};
  double M[4][5];
  for (int i = 0; i < 4; ++i) {{ for (int j = 0; j < 4; ++j) M[i][j] = 0.0; M[i][4] = L[i]; }}
  for (int a = 0; a < {nq}; ++a) for (int b = 0; b < {nq}; ++b) {{
    const double s = QP[a], t = QP[b];
    FD_QGEOM(xc, s, t)
    const double wq = QW[a]*QW[b]*fabs(det);
    (void)K00; (void)K01; (void)K10; (void)K11;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) M[i][j] += wq*N[i]*N[j];
  }}
  for (int c = 0; c < 4; ++c) {{                 /* Gauss-Jordan on the SPD 4x4 block */
    const double ip = 1.0 / M[c][c];
    for (int j = c; j < 5; ++j) M[c][j] *= ip;
    for (int r = 0; r < 4; ++r) if (r != c) {{
      const double fct = M[r][c];
      for (int j = c; j < 5; ++j) M[r][j] -= fct*M[c][j];
    }}
  }}
  for (int i = 0; i < 4; ++i) dq[i] = M[i][4];
}}
"""
    return op2.Kernel(body, "dg_mass_solve")


def dg_integrals_kernel(nq=3):
    """Zero-forms of the regression test (tests/firedrake/regression/test_dg_advection.py:60-72):
    g[0] += int q dx, g[1] += int q^2 dx.  Arguments: g[2] (Global INC), coords[8], q[4]."""
    from numpy.polynomial import legendre as leg
    x, w = leg.leggauss(nq)
    x, w = 0.5 * (x + 1.0), 0.5 * w
    body = f"""
{_QUAD_GEOM}
static void dg_integrals(double *restrict g, const double *restrict xc, const double *restrict qd)
{{
  static const double QP[{nq}] = {_c(x)}; static const double QW[{nq}] = {_c(w)};
  for (int a = 0; a < {nq}; ++a) for (int b = 0; b < {nq}; ++b) {{
    const double s = QP[a], t = QP[b];
    FD_QGEOM(xc, s, t)
    (void)K00; (void)K01; (void)K10; (void)K11;
    double qq = 0.0;
    for (int v = 0; v < 4; ++v) qq += qd[v]*N[v];
    const double wq = QW[a]*QW[b]*fabs(det);
    g[0] += wq*qq; g[1] += wq*qq*qq;
  }}
}}
"""
    return op2.Kernel(body, "dg_integrals")


class DGAdvectionStepper:
    """The demo's three-stage SSP Runge-Kutta loop (demos/DG_advection/DG_advection.py.rst), entirely on the
    device: RHS assembly (three parloops), block-diagonal mass solve (one parloop with an indirect WRITE),
    stage combinations with fd_dat_axpby."""

    def __init__(self, qmesh, dt=None):
        self.prob = p = DGAdvectionProblem(qmesh, dt)
        m = qmesh
        self.dq = op2.Dat(m.dq_set, None, np.float64, "dq")
        self.q0 = op2.Dat(m.dq_set, None, np.float64, "q_n")
        self.solve_loop = op2.LegacyParloop(dg_mass_solve_kernel(), m.cell_set, self.dq(op2.WRITE, m.cell_dq),
                                            m.coordinates(op2.READ, m.cell_q1), p.L(op2.READ, m.cell_dq))
        self.g = op2.Global(2, [0.0, 0.0], np.float64, "integrals")
        self.int_loop = op2.LegacyParloop(dg_integrals_kernel(), m.cell_set, self.g(op2.INC), m.coordinates(op2.READ, m.cell_q1),
                                          p.q(op2.READ, m.cell_dq))

    def _solve(self):
        self.prob.assemble_rhs()       # L(q)
        self.solve_loop()              # dq = M^{-1} L

    def step(self):
        q, q0, dq = self.prob.q, self.q0, self.dq
        q0.assign_dat(q)
        self._solve()
        q.axpby(1.0, dq, 1.0)                              # q1 = q + dq
        self._solve()
        q.axpby(0.25, dq, 0.25)                            # q2 = 0.75 q_n + 0.25 (q1 + dq)
        q.axpby(0.75, q0, 1.0)
        self._solve()
        q.axpby(2.0 / 3.0, dq, 2.0 / 3.0)                  # q_{n+1} = 1/3 q_n + 2/3 (q2 + dq)
        q.axpby(1.0 / 3.0, q0, 1.0)

    def integrals(self):
        """(int q dx, ||q||_L2)"""
        self.g.data[...] = 0.0
        self.int_loop()
        v = self.g.data_ro
        return float(v[0]), float(np.sqrt(v[1]))
