"""Tensor-product (hexahedral Q_k) elements: 1-D tabulations and the kernel descriptor the backend turns into the
matrix-core / sum-factorised wrappers of csrc/fd_tensor.h.

In Firedrake these tables come from FInAT/FIAT (tsfc/fem.py:711-735: ``ctx.basis_evaluation``; Q_k = CG_k (x) CG_k (x)
CG_k with GLL nodes, Gauss-Legendre quadrature from ``dx(degree=...)``, tsfc/fem.py:330-333); neither package is
available here, so the 1-D Lagrange tabulation is computed directly."""
from __future__ import annotations

import re

import numpy as np


def gll_nodes(k):
    """Gauss-Lobatto-Legendre points on [0, 1] (k+1 of them): the nodes of CG_k on an interval."""
    if k == 1:
        return np.array([0.0, 1.0])
    from numpy.polynomial import legendre as leg
    c = np.zeros(k + 1)
    c[k] = 1.0
    return 0.5 * (np.concatenate([[-1.0], np.sort(leg.legroots(leg.legder(c))), [1.0]]) + 1.0)


def gll_gauss_tables(degree=4, nq=5):
    """(L[q][i], DL[q][i], points, weights): values and derivatives of the CG_degree Lagrange basis on the GLL nodes at
    the nq Gauss-Legendre points of [0, 1]."""
    from numpy.polynomial import legendre as leg
    k = degree
    nodes = gll_nodes(k)
    x, w = leg.leggauss(nq)
    x, w = 0.5 * (x + 1.0), 0.5 * w
    L = np.zeros((nq, k + 1))
    DL = np.zeros((nq, k + 1))
    for i in range(k + 1):
        others = [nodes[m] for m in range(k + 1) if m != i]
        denom = np.prod([nodes[i] - o for o in others])
        for q in range(nq):
            L[q, i] = np.prod([x[q] - o for o in others]) / denom
            DL[q, i] = sum(np.prod([x[q] - o for mm, o in enumerate(others) if mm != m]) for m in range(k)) / denom
    return L, DL, x, w


# the point weight of a(u, v) = int alpha grad(u).grad(v) + (b.grad(u)) v + beta u v dx with constant alpha, b, beta:
#     W = w|J| [ alpha K K^T   0 ;  (K b)^T   beta ],   K = J^-1  (K[a][r] = d xi_a / d x_r)
# row l couples component l of the TEST side (d/dxi_1..3, value), column k component k of the TRIAL side; the advection term makes
# W non-symmetric (value of v against the reference gradient of u)
SECOND_ORDER_WEIGHTS = """
static inline void NAME_weights(const double J[3][3], const double X[3], double wq, double W[16])
{
  double K[3][3], det;
  fdt::inv3(J, K, det);
  const double w = wq * fabs(det);
  const double bv[3] = {BX, BY, BZ};
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) W[a*4 + b] = ALPHA * w * (K[a][0]*K[b][0] + K[a][1]*K[b][1] + K[a][2]*K[b][2]);
    W[a*4 + 3] = 0.0;
    W[12 + a] = w * (K[a][0]*bv[0] + K[a][1]*bv[1] + K[a][2]*bv[2]);
  }
  W[15] = BETA * w;
  (void)X;
}
"""


def second_order_weights(name, alpha=1.0, beta=1.0, velocity=(0.0, 0.0, 0.0)):
    """``weights_code`` of a(u, v) = int alpha grad(u).grad(v) + (b.grad(u)) v + beta u v dx, b = ``velocity``."""
    out = SECOND_ORDER_WEIGHTS.replace("ALPHA", repr(float(alpha))).replace("BETA", repr(float(beta)))
    for tag, v in zip(("BX", "BY", "BZ"), velocity):
        out = out.replace(tag, repr(float(v)))
    return out.replace("NAME_weights", name + "_weights")          # the name LAST: it may contain the parameter tokens


# a(u, v) = int kappa grad(u).grad(v) + c u v dx with kappa and c FUNCTIONS of the point and of coefficient fields: the shape of a
# variable-coefficient operator or of the Jacobian of a nonlinear reaction term (F = ... + g(u0) v  ->  c = g'(u0)).  ``KAPPA`` and
# ``REACT`` are C expressions in C[0..ncoef) (the coefficient values at the point) and X[0..2] (the physical point).
COEFFICIENT_WEIGHTS = """
static inline void NAME_weights(const double J[3][3], const double X[3], double wq, const double *C, double W[16])
{
  double K[3][3], det;
  fdt::inv3(J, K, det);
  const double w = wq * fabs(det);
  const double kappa = KAPPA, react = REACT;
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) W[a*4 + b] = kappa * w * (K[a][0]*K[b][0] + K[a][1]*K[b][1] + K[a][2]*K[b][2]);
    W[a*4 + 3] = 0.0;
    W[12 + a] = 0.0;
  }
  W[15] = react * w;
}
"""


def coefficient_weights(name, kappa="1.0", react="1.0"):
    """``weights_code`` of a(u, v) = int kappa grad(u).grad(v) + react u v dx, kappa / react C expressions in C[] and X[]."""
    return COEFFICIENT_WEIGHTS.replace("KAPPA", f"({kappa})").replace("REACT", f"({react})").replace("NAME_weights", name + "_weights")


# The Jacobian of F(u; v) = int (1 + |grad u|^2) grad(u).grad(v) dx at u0 (coefficient 0, with its REFERENCE gradient DC[0..2]):
#     a(du, v) = int (1 + |g|^2) grad(du).grad(v) + 2 (g.grad(du)) (g.grad(v)) dx,     g = grad(u0) = K^T DC
# -- the point weight is kappa K K^T + 2 (K g)(K g)^T on the gradient block.
NONLINEAR_DIFFUSION_WEIGHTS = """
static inline void NAME_weights(const double J[3][3], const double X[3], double wq, const double *C, const double *DC, double W[16])
{
  double K[3][3], det, g[3], Kg[3];
  fdt::inv3(J, K, det);
  const double w = wq * fabs(det);
  for (int r = 0; r < 3; ++r) g[r] = K[0][r]*DC[0] + K[1][r]*DC[1] + K[2][r]*DC[2];
  for (int a = 0; a < 3; ++a) Kg[a] = K[a][0]*g[0] + K[a][1]*g[1] + K[a][2]*g[2];
  const double kappa = 1.0 + g[0]*g[0] + g[1]*g[1] + g[2]*g[2];
  for (int a = 0; a < 3; ++a) {
    for (int b = 0; b < 3; ++b) W[a*4 + b] = w * (kappa * (K[a][0]*K[b][0] + K[a][1]*K[b][1] + K[a][2]*K[b][2]) + 2.0 * Kg[a] * Kg[b]);
    W[a*4 + 3] = 0.0;
    W[12 + a] = 0.0;
  }
  W[15] = 0.0;
  (void)X; (void)C;
}
"""


def nonlinear_diffusion_weights(name):
    """``weights_code`` (coef_gradients=True, ncoef=1) of the Newton Jacobian of int (1 + |grad u|^2) grad(u).grad(v) dx at u0."""
    return NONLINEAR_DIFFUSION_WEIGHTS.replace("NAME_weights", name + "_weights")


# Linear elasticity on (Q_k)^3:  a(u, v) = int 2 mu eps(u):eps(v) + lambda div(u) div(v) + rho u.v dx.  With v = e_p phi_i, u = e_r phi_j:
#     mu (delta_pr grad(phi_i).grad(phi_j) + d_r phi_i d_p phi_j) + lambda d_p phi_i d_r phi_j + rho delta_pr phi_i phi_j
# and d_s phi = sum_l K[l][s] dhat_l phi, so the 12 x 12 point weight is, on the gradient components (l of the test side, k of the trial side),
#     W[(p,l),(r,k)] = w ( mu delta_pr (K K^T)[l][k] + mu K[l][r] K[k][p] + lambda K[l][p] K[k][r] ),     W[(p,3),(r,3)] = w rho delta_pr.
ELASTICITY_WEIGHTS = """
static inline void NAME_weights(const double J[3][3], const double X[3], double wq, double W[144])
{
  double K[3][3], det;
  fdt::inv3(J, K, det);
  const double w = wq * fabs(det);
  for (int p = 0; p < 3; ++p) for (int l = 0; l < 4; ++l) for (int r = 0; r < 3; ++r) for (int k = 0; k < 4; ++k) {
    double v = 0.0;
    if (l < 3 && k < 3)
      v = MU * ((p == r ? K[l][0]*K[k][0] + K[l][1]*K[k][1] + K[l][2]*K[k][2] : 0.0) + K[l][r]*K[k][p]) + LAMBDA * K[l][p]*K[k][r];
    else if (l == 3 && k == 3 && p == r)
      v = RHO;
    W[(p*4 + l)*12 + r*4 + k] = w * v;
  }
  (void)X;
}
"""


def elasticity_weights(name, mu=1.0, lam=1.0, rho=0.0):
    """``weights_code`` (vdim=3) of a(u, v) = int 2 mu eps(u):eps(v) + lam div(u) div(v) + rho u.v dx."""
    # parameters first, the name LAST: a kernel name containing "MU" / "LAMBDA" / "RHO" (Firedrake's kernel names are arbitrary
    # identifiers) must not have its letters replaced by a float literal
    text = ELASTICITY_WEIGHTS
    for token, value in (("LAMBDA", lam), ("MU", mu), ("RHO", rho)):
        text = re.sub(r"\b%s\b" % token, repr(float(value)), text)
    return text.replace("NAME_weights", name + "_weights")
