// fd_callables.h -- device equivalents of the LAPACK callables of the reference's generated kernels.
//
// loopy turns gem.Inverse / gem.Solve (tsfc/loopy.py:368-391; Slate, hybridisation) into calls of `inverse(Aout, A, N)` and
// `solve(out, A, B, N)` (pyop2/codegen/rep2loopy.py:108-199), which the reference implements with getrf + getri / getrs on
// static 30x30 buffers (pyop2/codegen/c/inverse.c:20-47, solve.c:18-51; row-major A: getrs runs with 'T').  Here: per-lane
// Gauss-Jordan / LU with partial pivoting, N <= 30 like those buffers.  The pivot search is a chain of conditional ROW swaps
// with compile-time indices, so for the small constant N of an element kernel everything unrolls into registers.
#pragma once
#define FD_LAPACK_MAX 30
#define FD_LAPACK_UNROLLED 8          /* up to this N: fully unrolled, matrix in registers; above: loops over a private array */

namespace fdw {

template <int NC> __device__ __forceinline__ void inverse_n(double *__restrict__ Aout, const double *__restrict__ A, int Nrt) {
    constexpr int M = NC > 0 ? NC : FD_LAPACK_MAX;              // NC == 0: run-time N, worst-case buffer
    const int N = NC > 0 ? NC : Nrt;
    double w[M * M];
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) { w[i * N + j] = A[i * N + j]; Aout[i * N + j] = (i == j) ? 1.0 : 0.0; }
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = k + 1; i < N; ++i) {                       // row k <- the row with the largest |entry| in column k
            const bool sw = fabs(w[i * N + k]) > fabs(w[k * N + k]);
#pragma unroll
            for (int j = 0; j < N; ++j) {
                const double a = w[k * N + j], b = w[i * N + j]; w[k * N + j] = sw ? b : a; w[i * N + j] = sw ? a : b;
                const double c = Aout[k * N + j], d = Aout[i * N + j]; Aout[k * N + j] = sw ? d : c; Aout[i * N + j] = sw ? c : d;
            }
        }
        const double ip = 1.0 / w[k * N + k];
#pragma unroll
        for (int j = 0; j < N; ++j) { w[k * N + j] *= ip; Aout[k * N + j] *= ip; }
#pragma unroll
        for (int i = 0; i < N; ++i) {
            if (i == k) continue;
            const double f = w[i * N + k];
#pragma unroll
            for (int j = 0; j < N; ++j) { w[i * N + j] -= f * w[k * N + j]; Aout[i * N + j] -= f * Aout[k * N + j]; }
        }
    }
}

template <int NC> __device__ __forceinline__ void solve_n(double *__restrict__ out, const double *__restrict__ A, const double *__restrict__ B, int Nrt) {
    constexpr int M = NC > 0 ? NC : FD_LAPACK_MAX;
    const int N = NC > 0 ? NC : Nrt;
    double w[M * M];
#pragma unroll
    for (int i = 0; i < N; ++i) {
        out[i] = B[i];
#pragma unroll
        for (int j = 0; j < N; ++j) w[i * N + j] = A[i * N + j];
    }
#pragma unroll
    for (int k = 0; k < N; ++k) {
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const bool sw = fabs(w[i * N + k]) > fabs(w[k * N + k]);
#pragma unroll
            for (int j = k; j < N; ++j) { const double a = w[k * N + j], b = w[i * N + j]; w[k * N + j] = sw ? b : a; w[i * N + j] = sw ? a : b; }
            const double c = out[k], d = out[i]; out[k] = sw ? d : c; out[i] = sw ? c : d;
        }
        const double ip = 1.0 / w[k * N + k];
#pragma unroll
        for (int i = k + 1; i < N; ++i) {
            const double f = w[i * N + k] * ip;
#pragma unroll
            for (int j = k + 1; j < N; ++j) w[i * N + j] -= f * w[k * N + j];
            out[i] -= f * out[k];
        }
    }
#pragma unroll
    for (int k = N - 1; k >= 0; --k) {
        double s = out[k];
#pragma unroll
        for (int j = k + 1; j < N; ++j) s -= w[k * N + j] * out[j];
        out[k] = s / w[k * N + k];
    }
}

}  // namespace fdw

// larger systems: one out-of-line copy with run-time loops over a private 30x30 array (kept out of line so that a literal N
// does not unroll N^3 statements into the caller)
__device__ __attribute__((noinline)) static void fd_inverse_rt(double *Aout, const double *A, int N) { fdw::inverse_n<0>(Aout, A, N); }
__device__ __attribute__((noinline)) static void fd_solve_rt(double *out, const double *A, const double *B, int N) { fdw::solve_n<0>(out, A, B, N); }

// N is a literal in generated code, so after inlining the switch folds to one instantiation
__device__ __forceinline__ void inverse(double *__restrict__ Aout, const double *__restrict__ A, int N) {
    switch (N) {
    case 1: fdw::inverse_n<1>(Aout, A, N); break;
    case 2: fdw::inverse_n<2>(Aout, A, N); break;
    case 3: fdw::inverse_n<3>(Aout, A, N); break;
    case 4: fdw::inverse_n<4>(Aout, A, N); break;
    case 5: fdw::inverse_n<5>(Aout, A, N); break;
    case 6: fdw::inverse_n<6>(Aout, A, N); break;
    case 7: fdw::inverse_n<7>(Aout, A, N); break;
    case 8: fdw::inverse_n<8>(Aout, A, N); break;
    default:
        if (N < 1 || N > FD_LAPACK_MAX) __builtin_trap();
        fd_inverse_rt(Aout, A, N);
    }
}
__device__ __forceinline__ void solve(double *__restrict__ out, const double *__restrict__ A, const double *__restrict__ B, int N) {
    switch (N) {
    case 1: fdw::solve_n<1>(out, A, B, N); break;
    case 2: fdw::solve_n<2>(out, A, B, N); break;
    case 3: fdw::solve_n<3>(out, A, B, N); break;
    case 4: fdw::solve_n<4>(out, A, B, N); break;
    case 5: fdw::solve_n<5>(out, A, B, N); break;
    case 6: fdw::solve_n<6>(out, A, B, N); break;
    case 7: fdw::solve_n<7>(out, A, B, N); break;
    case 8: fdw::solve_n<8>(out, A, B, N); break;
    default:
        if (N < 1 || N > FD_LAPACK_MAX) __builtin_trap();
        fd_solve_rt(out, A, B, N);
    }
}
