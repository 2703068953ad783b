"""Local kernels (C strings) restating the *mathematics* of the reference's PyOP2 test
kernels (tests/pyop2/test_matrices.py:168-352) around the golden DATA in
tests/golden/pyop2_matrices.json.  Written from scratch -- same quadrature tables,
own loop structure -- so both the oracle (gcc) and the HIP backend (hipcc) consume
the same C source, exactly as the reference's CStringLocalKernel route does.
"""
import json
import os

import numpy as np

_G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pyop2_matrices.json")))
GOLD = _G


def _arr(a):
    a = np.asarray(a)
    if a.ndim == 1:
        return "{" + ", ".join(repr(float(x)) for x in a) + "}"
    return "{" + ", ".join(_arr(r) for r in a) + "}"


_Q6 = f"""
  const double PHI[3][6] = {_arr(_G['quad6_basis'])};
  const double DPHI[3][2] = {_arr(_G['quad6_dbasis'])};
  const double WQ[6] = {_arr(_G['quad6_weights'])};
"""

# P1 mass matrix on a triangle, 6-point rule; Jacobian from the tabulated gradients.
MASS_Q6 = f"""
static void mass_q6(double A[9], const double xy[6])
{{
{_Q6}
  double J[2][2] = {{{{0.0, 0.0}}, {{0.0, 0.0}}}};
  for (int v = 0; v < 3; ++v)
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b)
        J[a][b] += xy[2*v + a] * DPHI[v][b];
  const double det = J[0][0]*J[1][1] - J[0][1]*J[1][0];
  for (int q = 0; q < 6; ++q) {{
    const double s = WQ[q] * det;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        A[3*i + j] += PHI[i][q] * PHI[j][q] * s;
  }}
}}
"""

RHS_Q6 = f"""
static void rhs_q6(double *b, const double xy[6], const double *fn)
{{
{_Q6}
  double J[2][2] = {{{{0.0, 0.0}}, {{0.0, 0.0}}}};
  for (int v = 0; v < 3; ++v)
    for (int a = 0; a < 2; ++a)
      for (int c = 0; c < 2; ++c)
        J[a][c] += xy[2*v + a] * DPHI[v][c];
  const double det = J[0][0]*J[1][1] - J[0][1]*J[1][0];
  for (int q = 0; q < 6; ++q) {{
    double fq = 0.0;
    for (int v = 0; v < 3; ++v) fq += fn[v] * PHI[v][q];
    for (int i = 0; i < 3; ++i)
      b[i] += PHI[i][q] * fq * det * WQ[q];
  }}
}}
"""

_Q3 = f"""
  const double B3[3][3] = {_arr(_G['quad3_basis'])};
  const double W3[3] = {_arr(_G['quad3_weights'])};
  const double e1x = xy[2] - xy[0], e2x = xy[4] - xy[0];
  const double e1y = xy[3] - xy[1], e2y = xy[5] - xy[1];
  const double area2 = fabs(e1x*e2y - e2x*e1y);
"""

MASS_AFFINE = f"""
static void mass_affine(double A[9], const double xy[6])
{{
{_Q3}
  for (int q = 0; q < 3; ++q)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        A[3*i + j] += B3[q][i] * B3[q][j] * W3[q] * area2;
}}
"""

RHS_AFFINE = f"""
static void rhs_affine(double *b, const double xy[6], const double *fn)
{{
{_Q3}
  for (int q = 0; q < 3; ++q) {{
    double fq = 0.0;
    for (int v = 0; v < 3; ++v) fq += B3[q][v] * fn[v];
    for (int i = 0; i < 3; ++i) b[i] += B3[q][i] * fq * W3[q] * area2;
  }}
}}
"""

# Vector-valued (cdim 2) variants: element tensor laid out [i][p][j][q] as the MatPack is
# (pyop2/codegen/builder.py:538-548): block (i,j) is mass_ij * I_2.
MASS_VEC_AFFINE = f"""
static void mass_vec_affine(double A[36], const double xy[6])
{{
{_Q3}
  for (int q = 0; q < 3; ++q)
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) {{
        const double m = B3[q][i] * B3[q][j] * W3[q] * area2;
        for (int p = 0; p < 2; ++p)
          A[((i*2 + p)*3 + j)*2 + p] += m;
      }}
}}
"""

RHS_VEC_AFFINE = f"""
static void rhs_vec_affine(double *b, const double xy[6], const double *fn)
{{
{_Q3}
  for (int q = 0; q < 3; ++q)
    for (int p = 0; p < 2; ++p) {{
      double fq = 0.0;
      for (int v = 0; v < 3; ++v) fq += B3[q][v] * fn[2*v + p];
      for (int i = 0; i < 3; ++i) b[2*i + p] += B3[q][i] * fq * W3[q] * area2;
    }}
}}
"""

ELEM_NODE = np.asarray(_G["elem_node_map"], dtype=np.int32)
COORDS = np.asarray(_G["coords"], dtype=np.float64)
F = np.asarray(_G["f"], dtype=np.float64)
F_VEC = np.asarray(_G["f_vec"], dtype=np.float64)
