"""Tensor-product (hexahedral Q_k) elements: 1-D tabulations and the kernel descriptor the backend turns into the
matrix-core / sum-factorised wrappers of csrc/fd_tensor.h.

In Firedrake these tables come from FInAT/FIAT (tsfc/fem.py:711-735: ``ctx.basis_evaluation``; Q_k = CG_k (x) CG_k (x)
CG_k with GLL nodes, Gauss-Legendre quadrature from ``dx(degree=...)``, tsfc/fem.py:330-333); neither package is
available here, so the 1-D Lagrange tabulation is computed directly."""
from __future__ import annotations

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
    out = SECOND_ORDER_WEIGHTS.replace("NAME", name).replace("ALPHA", repr(float(alpha))).replace("BETA", repr(float(beta)))
    for tag, v in zip(("BX", "BY", "BZ"), velocity):
        out = out.replace(tag, repr(float(v)))
    return out


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
    return COEFFICIENT_WEIGHTS.replace("NAME", name).replace("KAPPA", f"({kappa})").replace("REACT", f"({react})")
