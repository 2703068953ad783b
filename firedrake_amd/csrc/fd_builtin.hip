// fd_builtin.hip -- wrapper kernels compiled ahead of time into libfdhip.so and reachable
// through fd_kernel_builtin().  (The JIT route -- codegen.py + hipcc --genco -- produces
// the same kind of kernel for arbitrary local kernels.)
#include "fd_common.h"
#include "fd_wrapper.h"

namespace fd { int register_builtin(const char *name, const void *fn); }
#define FD_REGISTER(sym) static int _fd_reg_##sym = fd::register_builtin(#sym, (const void *)sym)

// Smoke kernel: y[i] += a * x[i] over [start, end) -- used by the CPU-side ABI test (symbol
// presence) and the first GPU sanity check of launch marshalling.
extern "C" __global__ void wrap_fd_axpy(int start, int end, double *y, const double *x, const double *a) {
    for (int i = start + blockIdx.x * blockDim.x + threadIdx.x; i < end; i += gridDim.x * blockDim.x)
        y[i] += a[0] * x[i];
}
FD_REGISTER(wrap_fd_axpy);

// ---- PMC calibration kernels (bench.py --traffic, tools/pmc_calibrate.sh): known byte counts in the access widths the
// wrapper kernels use, so the FETCH_SIZE / WRITE_SIZE readings of rocprofv3 can be converted to bytes for THOSE patterns
// (MI355X_MICROARCH.md, HBM section: only 16 B/lane streaming reads are calibrated there).  Element i of [start, end) is
// read (or written) once, consecutive lanes touch consecutive elements; `sink` receives one value per workgroup.
template <class T> __device__ __forceinline__ double calib_val(const T &v) { return (double)v; }
template <> __device__ __forceinline__ double calib_val<uint4>(const uint4 &v) { return (double)(v.x ^ v.y ^ v.z ^ v.w); }
template <> __device__ __forceinline__ double calib_val<uint2>(const uint2 &v) { return (double)(v.x ^ v.y); }

template <class T> __device__ __forceinline__ void calib_read(int start, int end, const T *__restrict__ src, double *__restrict__ sink) {
    double acc = 0.0;
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        acc += calib_val<T>(src[i]);
    acc = fdw::wave_reduce<double, fdw::OpAdd<double>>(acc);
    if ((threadIdx.x & 63) == 0 && acc == 1.2345e300) sink[blockIdx.x] = acc;      // keeps the loads alive, (almost) never stores
}
extern "C" __global__ void wrap_fd_calib_read2(int start, int end, const unsigned short *src, double *sink) { calib_read<unsigned short>(start, end, src, sink); }
extern "C" __global__ void wrap_fd_calib_read4(int start, int end, const unsigned int *src, double *sink) { calib_read<unsigned int>(start, end, src, sink); }
extern "C" __global__ void wrap_fd_calib_read8(int start, int end, const uint2 *src, double *sink) { calib_read<uint2>(start, end, src, sink); }
extern "C" __global__ void wrap_fd_calib_read16(int start, int end, const uint4 *src, double *sink) { calib_read<uint4>(start, end, src, sink); }
// random 8-byte gathers through an index array (the pattern of the node-row staging): idx read coalesced, src gathered
extern "C" __global__ void wrap_fd_calib_gather8(int start, int end, const double *src, const int *idx, double *sink) {
    double acc = 0.0;
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        acc += src[idx[i]];
    acc = fdw::wave_reduce<double, fdw::OpAdd<double>>(acc);
    if ((threadIdx.x & 63) == 0 && acc == 1.2345e300) sink[blockIdx.x] = acc;
}
extern "C" __global__ void wrap_fd_calib_write8(int start, int end, double *dst) {
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        dst[i] = (double)i;
}
extern "C" __global__ void wrap_fd_calib_atomic8(int start, int end, double *dst) {       // streaming fp64 atomics (the staged flush)
    for (long long i = start + (long long)blockIdx.x * blockDim.x + threadIdx.x; i < end; i += (long long)gridDim.x * blockDim.x)
        atomicAdd(&dst[i], 1.0);
}
FD_REGISTER(wrap_fd_calib_read2);
FD_REGISTER(wrap_fd_calib_read4);
FD_REGISTER(wrap_fd_calib_read8);
FD_REGISTER(wrap_fd_calib_read16);
FD_REGISTER(wrap_fd_calib_gather8);
FD_REGISTER(wrap_fd_calib_write8);
FD_REGISTER(wrap_fd_calib_atomic8);

