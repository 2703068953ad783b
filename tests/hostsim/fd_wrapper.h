// tests/hostsim/fd_wrapper.h -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Host stand-in for firedrake_amd/csrc/fd_wrapper.h: lets g++ compile a DIRECT-mode wrapper kernel
// emitted by firedrake_amd/codegen.py and execute it with ONE sequential "lane" (gridDim = blockDim = 1;
// the wrappers use grid-stride loops, so one lane visits every entity).  tests/hostsim.py uses it to
// check the generated indexing logic (maps, offsets, layers, subsets, lgmaps, CSR search) against the
// oracle on machines without a GPU.  It says nothing about the HIP build -- the -m gpu parity tests do.
#pragma once
#include <stdint.h>
#include <stddef.h>
#include <math.h>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static
#define restrict __restrict__

typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscInt;
typedef long long fd_nnz_t;      // firedrake_amd/csrc/fd_wrapper.h (int64_t)

struct fd_sim_dim3 { int x, y, z; };
static const fd_sim_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0}, blockDim = {1, 1, 1}, gridDim = {1, 1, 1};
static inline void __syncthreads() {}

namespace fdw {
template <class T> inline void atomic_add(T *p, T v) { *p += v; }
template <class T> inline void atomic_min(T *p, T v) { if (v < *p) *p = v; }
template <class T> inline void atomic_max(T *p, T v) { if (v > *p) *p = v; }

inline int wrap_layer(int a, int nl) { return a % nl; }

inline fd_nnz_t csr_find(const fd_nnz_t *rowptr, const int *colidx, int r, int c) {
    for (fd_nnz_t q = rowptr[r]; q < rowptr[r + 1]; ++q)
        if (colidx[q] == c) return q;
    return -1;
}

template <class T> struct OpAdd { static T f(T a, T b) { return a + b; } };
template <class T> struct OpMin { static T f(T a, T b) { return a < b ? a : b; } };
template <class T> struct OpMax { static T f(T a, T b) { return a > b ? a : b; } };
template <class T, class Op> inline T block_reduce(T v, T *) { return v; }
}  // namespace fdw

#include "../../oracle/callables.h"
