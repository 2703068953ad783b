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
