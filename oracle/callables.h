/* oracle/callables.h -- TEST INFRASTRUCTURE (CPU oracle; also included by the tests' host stand-ins of fd_wrapper.h).
 *
 * The reference's generated kernels may call `inverse(Aout, A, N)` / `solve(out, A, B, N)`: getrf + getri, resp. getrf +
 * getrs('T') on the row-major N x N matrix A (pyop2/codegen/c/inverse.c:20-47, solve.c:18-51; registered by
 * pyop2/codegen/rep2loopy.py:108-199).  LAPACK is not linked here: the same factorisation -- LU with partial pivoting by
 * the largest |entry| of the column -- is restated in plain C.  Pinned against numpy.linalg by tests/test_callables.py, the
 * check the reference's own tests/pyop2/test_callables.py:85-126 makes. */
#ifndef ORACLE_CALLABLES_H
#define ORACLE_CALLABLES_H
#include <math.h>
#include <stdlib.h>
#define FD_LAPACK_MAX 30
#define FD_CALLABLE_FAIL() abort()
static inline void inverse(double *__restrict__ Aout, const double *__restrict__ A, int N)
{
    if (N > FD_LAPACK_MAX) FD_CALLABLE_FAIL();
    double w[FD_LAPACK_MAX * FD_LAPACK_MAX];
    for (int i = 0; i < N; ++i)
        for (int j = 0; j < N; ++j) { w[i * N + j] = A[i * N + j]; Aout[i * N + j] = (i == j) ? 1.0 : 0.0; }
    for (int k = 0; k < N; ++k) {
        for (int i = k + 1; i < N; ++i) {                       /* row k <- the row with the largest |entry| in column k */
            const int sw = fabs(w[i * N + k]) > fabs(w[k * N + k]);
            for (int j = 0; j < N; ++j) {
                const double a = w[k * N + j], b = w[i * N + j]; w[k * N + j] = sw ? b : a; w[i * N + j] = sw ? a : b;
                const double c = Aout[k * N + j], d = Aout[i * N + j]; Aout[k * N + j] = sw ? d : c; Aout[i * N + j] = sw ? c : d;
            }
        }
        const double ip = 1.0 / w[k * N + k];
        for (int j = 0; j < N; ++j) { w[k * N + j] *= ip; Aout[k * N + j] *= ip; }
        for (int i = 0; i < N; ++i) {
            if (i == k) continue;
            const double f = w[i * N + k];
            for (int j = 0; j < N; ++j) { w[i * N + j] -= f * w[k * N + j]; Aout[i * N + j] -= f * Aout[k * N + j]; }
        }
    }
}
static inline void solve(double *__restrict__ out, const double *__restrict__ A, const double *__restrict__ B, int N)
{
    if (N > FD_LAPACK_MAX) FD_CALLABLE_FAIL();
    double w[FD_LAPACK_MAX * FD_LAPACK_MAX];
    for (int i = 0; i < N; ++i) { out[i] = B[i]; for (int j = 0; j < N; ++j) w[i * N + j] = A[i * N + j]; }
    for (int k = 0; k < N; ++k) {
        for (int i = k + 1; i < N; ++i) {
            const int sw = fabs(w[i * N + k]) > fabs(w[k * N + k]);
            for (int j = k; j < N; ++j) { const double a = w[k * N + j], b = w[i * N + j]; w[k * N + j] = sw ? b : a; w[i * N + j] = sw ? a : b; }
            const double c = out[k], d = out[i]; out[k] = sw ? d : c; out[i] = sw ? c : d;
        }
        const double ip = 1.0 / w[k * N + k];
        for (int i = k + 1; i < N; ++i) {
            const double f = w[i * N + k] * ip;
            for (int j = k + 1; j < N; ++j) w[i * N + j] -= f * w[k * N + j];
            out[i] -= f * out[k];
        }
    }
    for (int k = N - 1; k >= 0; --k) {
        double s = out[k];
        for (int j = k + 1; j < N; ++j) s -= w[k * N + j] * out[j];
        out[k] = s / w[k * N + k];
    }
}
#endif
