// fd_halo.hip -- device pack/unpack for halo exchange and pointwise row helpers
// (include/fdhip.h: fd_halo_*, fd_dat_*).
//
// Device side of firedrake/halo.py:125-172: the reference lets PetscSF gather the
// owner values into MPI buffers (bcast, MPI.REPLACE) or combine ghost contributions
// into owners (reduce, SUM/MIN/MAX).  Here the gather/scatter are explicit kernels over
// per-neighbour index lists and the wire transfer is RCCL send/recv on the packed buffers.
#include "fd_common.h"

namespace {

__global__ void pack_rows(const double *__restrict__ dat, int cdim, const int32_t *__restrict__ idx, int32_t n,
                          double *__restrict__ buf) {
    const int64_t total = (int64_t)n * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = t / cdim;
        int c = (int)(t - k * cdim);
        buf[t] = dat[(int64_t)idx[k] * cdim + c];
    }
}

// Index lists of one exchange never repeat a node, so no atomics are needed.
__global__ void unpack_rows(double *__restrict__ dat, int cdim, const int32_t *__restrict__ idx, int32_t n,
                            const double *__restrict__ buf, int op) {
    const int64_t total = (int64_t)n * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = t / cdim;
        int c = (int)(t - k * cdim);
        double *p = &dat[(int64_t)idx[k] * cdim + c];
        double v = buf[t];
        switch (op) {
            case 0: *p = v; break;
            case 1: *p += v; break;
            case 2: *p = fmin(*p, v); break;
            default: *p = fmax(*p, v); break;
        }
    }
}

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

int fd_halo_pack(const double *dat, int cdim, const int32_t *idx, int32_t n, double *buf, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(pack_rows, dim3(grid_for((int64_t)n * cdim)), dim3(256), 0, fd::st(s), dat, cdim, idx, n, buf);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_halo_unpack(double *dat, int cdim, const int32_t *idx, int32_t n, const double *buf, int op, fd_stream_t s) {
    if (n <= 0) return 0;
    if (op < 0 || op > 3) FD_FAIL("fd_halo_unpack: op must be 0..3");
    hipLaunchKernelGGL(unpack_rows, dim3(grid_for((int64_t)n * cdim)), dim3(256), 0, fd::st(s), dat, cdim, idx, n, buf, op);
    FD_CHECK_LAUNCH();
    return 0;
}

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
