// fd_pointwise.hip -- pointwise row helpers on Dats (include/fdhip.h: fd_dat_*): boundary-condition rows
// (firedrake/bcs.py:192-221, 404-457) and y = a x + b y.  (The halo pack/unpack kernels live with the exchange in
// fd_comm.hip, typed for every Dat dtype.)
#include "fd_common.h"

namespace {

__global__ void set_rows(double *__restrict__ dat, int cdim, const int32_t *__restrict__ rows, int32_t n, double v) {
    const int64_t total = (int64_t)n * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = t / cdim;
        dat[(int64_t)rows[k] * cdim + (t - k * cdim)] = v;
    }
}

__global__ void copy_rows(double *__restrict__ dst, const double *__restrict__ src, int cdim,
                          const int32_t *__restrict__ rows, int32_t n) {
    const int64_t total = (int64_t)n * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = t / cdim;
        int64_t a = (int64_t)rows[k] * cdim + (t - k * cdim);
        dst[a] = src[a];
    }
}

__global__ void axpby_k(double *__restrict__ y, double a, const double *__restrict__ x, double b, int64_t n) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        y[t] = a * x[t] + b * y[t];
}

inline int grid_for(int64_t n) { int64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 8192) g = 8192; return (int)g; }

}  // namespace

extern "C" {

int fd_dat_set_rows(double *dat, int cdim, const int32_t *rows, int32_t n, double v, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(set_rows, dim3(grid_for((int64_t)n * cdim)), dim3(256), 0, fd::st(s), dat, cdim, rows, n, v);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_dat_axpby(double *y, double a, const double *x, double b, int64_t n, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(axpby_k, dim3(grid_for(n)), dim3(256), 0, fd::st(s), y, a, x, b, n);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_dat_copy_rows(double *dst, const double *src, int cdim, const int32_t *rows, int32_t n, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(copy_rows, dim3(grid_for((int64_t)n * cdim)), dim3(256), 0, fd::st(s), dst, src, cdim, rows, n);
    FD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
