"""Local kernels written in the C dialect that loopy's C target emits for TSFC kernels (tsfc/loopy.py:216-283,
pyop2/codegen/rep2loopy.py:560-570): the four standard includes (two of them host-only), a non-static function with
``__restrict__`` pointer arguments in TSFC's order (output tensor, coordinates, coefficients --
tsfc/kernel_interface/firedrake_loopy.py:432-522), scalar temporaries declared up front, ``double const tN[...]``
tabulation tables with initialisers, ``int32_t`` loop counters with inclusive bounds, accumulation into a zeroed
output.  loopy itself is not installed here, so these texts are hand-written to that dialect (they are NOT captured
loopy output); what they pin is that the ingestion route compiles and runs such text unchanged."""

_HEADER = """#include <complex.h>
#include <math.h>
#include <petsc.h>
#include <stdint.h>
"""

# u*v*dx on P1 triangles, 3-point degree-2 rule
MASS_P1 = _HEADER + """
void form0_cell_integral(double *__restrict__ A, double const *__restrict__ coords)
{
  double t0;
  double t1;
  double t2;
  double const t3[3] = { 0.16666666666666666, 0.16666666666666666, 0.16666666666666666 };
  double const t4[3 * 3] = { 0.6666666666666667, 0.16666666666666666, 0.16666666666666666, 0.16666666666666666, 0.6666666666666667, 0.16666666666666666, 0.16666666666666666, 0.16666666666666666, 0.6666666666666667 };
  double t5;
  double t6[3];

  t0 = -1.0 * coords[0];
  t1 = -1.0 * coords[1];
  t2 = fabs((t0 + coords[2]) * (t1 + coords[5]) + -1.0 * (t0 + coords[4]) * (t1 + coords[3]));
  for (int32_t ip = 0; ip <= 2; ++ip)
  {
    t5 = t3[ip] * t2;
    for (int32_t k = 0; k <= 2; ++k)
      t6[k] = t4[3 * ip + k] * t5;
    for (int32_t j = 0; j <= 2; ++j)
      for (int32_t k = 0; k <= 2; ++k)
        A[3 * j + k] = A[3 * j + k] + t4[3 * ip + j] * t6[k];
  }
}
"""

# f*v*dx on P1 triangles with a P1 coefficient
RHS_P1 = _HEADER + """
void form1_cell_integral(double *__restrict__ A, double const *__restrict__ coords, double const *__restrict__ w_0)
{
  double t0;
  double t1;
  double t2;
  double const t3[3] = { 0.16666666666666666, 0.16666666666666666, 0.16666666666666666 };
  static double const t4[3][3] = { { 0.6666666666666667, 0.16666666666666666, 0.16666666666666666 }, { 0.16666666666666666, 0.6666666666666667, 0.16666666666666666 }, { 0.16666666666666666, 0.16666666666666666, 0.6666666666666667 } };
  double t5;
  double t6;

  t0 = -1.0 * coords[0];
  t1 = -1.0 * coords[1];
  t2 = fabs((t0 + coords[2]) * (t1 + coords[5]) + -1.0 * (t0 + coords[4]) * (t1 + coords[3]));
  for (int32_t ip = 0; ip <= 2; ++ip)
  {
    t5 = 0.0;
    for (int32_t i = 0; i <= 2; ++i)
      t5 = t5 + t4[ip][i] * w_0[i];
    t6 = t3[ip] * t2 * t5;
    for (int32_t j = 0; j <= 2; ++j)
      A[j] = A[j] + t4[ip][j] * t6;
  }
}
"""
