// fd_wrapper.h -- device-side building blocks of the wrapper kernels.
//
// The reference generates, per parloop, a sequential C function
//     int wrap_<kernel>(start, end, [layers], [subset], dats..., maps...)
// (pyop2/codegen/builder.py:702-1008, rep2loopy.py:409-593) that packs element data,
// calls the TSFC/loopy local kernel and unpacks (+=, min, max, =, MatSetValues).
// On MI355X the same contract is a HIP kernel assembled from these pieces by
// firedrake_amd/codegen.py (JIT) or instantiated ahead of time in fd_builtin.hip:
//
//   * one workgroup per block of iteration-set entities, one lane per entity
//     (wave64: 64 elements per wavefront);
//   * staged mode: the distinct Dat rows a block touches are gathered once, coalesced
//     over the block's sorted node list, into LDS; element packs are read from LDS;
//     INC contributions are reduced in LDS (ds_add_f64) and leave the CU as ONE global
//     atomic per distinct node;
//   * direct mode: thread-per-entity gather/scatter straight from global memory with
//     hardware fp64 atomics -- the always-available fallback for exotic access modes
//     (RW/WRITE/MIN/MAX through maps, subsets, extruded columns, integer Dats);
//   * workgroups are renumbered so that consecutive blocks (which share nodes) land on
//     the same XCD and hit in its 4 MiB L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscInt;
#define restrict __restrict__   /* C99 keyword used by loopy/TSFC-generated C */
typedef double fd_d4 __attribute__((ext_vector_type(4)));   /* accumulator tile of v_mfma_f64_16x16x4_f64 (fd_tensor.h) */

namespace fdw {

// Hardware places workgroup b on XCD b % 8 (MI355X_MICROARCH: "block b runs on XCD b % 8").
// Map it to a logical block id such that each XCD owns one contiguous range of blocks.
// Bijection for any nb; used for speed only, never for correctness.
__device__ __forceinline__ int xcd_block(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = bid & 7, k = bid >> 3;
    return x * q + (x < r ? x : r) + k;
}

// A value the caller knows to be equal in all lanes of the wavefront (the row index of a sliced instance chunk): moving it
// to a scalar register turns the switch on it into scalar branches -- no divergence bookkeeping, one instantiation executed.
__device__ __forceinline__ int wave_uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }

// ---- global atomics (compiled with -munsafe-fp-atomics => global_atomic_add_f64) ----
template <class T> __device__ __forceinline__ void atomic_add(T *p, T v) { atomicAdd(p, v); }
template <> __device__ __forceinline__ void atomic_add<long long>(long long *p, long long v) {
    atomicAdd((unsigned long long *)p, (unsigned long long)v);
}
template <class T> __device__ __forceinline__ void atomic_min(T *p, T v) { atomicMin(p, v); }
template <class T> __device__ __forceinline__ void atomic_max(T *p, T v) { atomicMax(p, v); }
// int64_t / uint64_t are `long` / `unsigned long` on this ABI; HIP's 64-bit integer atomics are declared for the
// `long long` spellings, which have the same representation
template <> __device__ __forceinline__ void atomic_add<long>(long *p, long v) {
    atomicAdd((unsigned long long *)p, (unsigned long long)v);
}
template <> __device__ __forceinline__ void atomic_add<unsigned long>(unsigned long *p, unsigned long v) {
    atomicAdd((unsigned long long *)p, (unsigned long long)v);
}
template <> __device__ __forceinline__ void atomic_min<long>(long *p, long v) { atomicMin((long long *)p, (long long)v); }
template <> __device__ __forceinline__ void atomic_max<long>(long *p, long v) { atomicMax((long long *)p, (long long)v); }
template <> __device__ __forceinline__ void atomic_min<unsigned long>(unsigned long *p, unsigned long v) {
    atomicMin((unsigned long long *)p, (unsigned long long)v);
}
template <> __device__ __forceinline__ void atomic_max<unsigned long>(unsigned long *p, unsigned long v) {
    atomicMax((unsigned long long *)p, (unsigned long long)v);
}

// ---- periodic extrusion: layer offset modulo the number of cell layers (pyop2/codegen/builder.py:101-123; the
// reference's ad hoc _Remainder subtracts once, which is the same for every offset it can produce when nl >= 2)
__device__ __forceinline__ int wrap_layer(int a, int nl) { return a % nl; }

// ---- CSR position of (row, col): the per-call row search of MatSetValuesLocal ----
__device__ __forceinline__ int csr_find(const int *__restrict__ rowptr, const int *__restrict__ colidx, int r, int c) {
    int lo = rowptr[r], hi = rowptr[r + 1] - 1;
    while (lo <= hi) {
        int mid = lo + ((hi - lo) >> 1);
        int v = colidx[mid];
        if (v == c) return mid;
        if (v < c) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

// ---- block reductions for Global INC/MIN/MAX (GlobalPack, builder.py:292-319) ----
template <class T> struct OpAdd { static __device__ __forceinline__ T f(T a, T b) { return a + b; } };
template <class T> struct OpMin { static __device__ __forceinline__ T f(T a, T b) { return a < b ? a : b; } };
template <class T> struct OpMax { static __device__ __forceinline__ T f(T a, T b) { return a > b ? a : b; } };

template <class T, class Op> __device__ __forceinline__ T wave_reduce(T v) {
    for (int d = 32; d > 0; d >>= 1) {
        T o = __shfl_xor(v, d, 64);
        v = Op::f(v, o);
    }
    return v;
}

// All threads of the block must call.  Result valid on thread 0.
template <class T, class Op> __device__ __forceinline__ T block_reduce(T v, T *scratch /* >= 16 */) {
    v = wave_reduce<T, Op>(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[w] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < nw; ++i) v = Op::f(v, scratch[i]);
    }
    return v;
}

// ---- replicated LDS accumulators: accumulator q has R = 2^rsh copies at s[q*R .. q*R + R-1].
// Sum of the copies; consecutive lanes start at rotated copies so the R reads of a wavefront spread over the banks.
template <class T> __device__ __forceinline__ T rep_sum(const T *s, int q, int rsh) {
    if (rsh == 0) return s[q];
    const int R = 1 << rsh, rot = (rsh < 5) ? (q >> (5 - rsh)) : q;
    const T *p = s + ((size_t)q << rsh);
    T v = 0;
    for (int r = 0; r < R; ++r) v += p[(r + rot) & (R - 1)];
    return v;
}

// ---- packed per-entity index rows (uint16 local maps, uint8/uint16 matrix offsets): N small
// unsigned integers read with the widest loads the row size allows (rows are N*sizeof(T) apart).
template <class T, int N> __device__ __forceinline__ void load_packed(const T *__restrict__ p, int (&out)[N]) {
    constexpr int BYTES = N * (int)sizeof(T);
    constexpr int PER = 4 / (int)sizeof(T);          // entries per 32-bit word
    constexpr unsigned MASK = sizeof(T) == 1 ? 0xffu : 0xffffu;
    constexpr int SH = 8 * (int)sizeof(T);
    if constexpr (BYTES % 16 == 0) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int k = 0; k < BYTES / 16; ++k) {
            uint4 v = q[k];
            unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int c = 0; c < PER; ++c) out[(4 * k + a) * PER + c] = (w[a] >> (SH * c)) & MASK;
        }
    } else if constexpr (BYTES % 8 == 0) {
        const uint2 *q = reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int k = 0; k < BYTES / 8; ++k) {
            uint2 v = q[k];
            unsigned w[2] = {v.x, v.y};
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int c = 0; c < PER; ++c) out[(2 * k + a) * PER + c] = (w[a] >> (SH * c)) & MASK;
        }
    } else if constexpr (BYTES % 4 == 0) {
        const unsigned *q = reinterpret_cast<const unsigned *>(p);
#pragma unroll
        for (int k = 0; k < BYTES / 4; ++k) {
            unsigned v = q[k];
#pragma unroll
            for (int c = 0; c < PER; ++c) out[k * PER + c] = (v >> (SH * c)) & MASK;
        }
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = p[k];
    }
}

template <int ARITY> __device__ __forceinline__ void load_lmap(const uint16_t *__restrict__ p, int (&out)[ARITY]) {
    load_packed<uint16_t, ARITY>(p, out);
}

// ---- bit-packed instance records (fd_ocr_pack_records): W 32-bit words per instance hold every per-instance index of an
// owner-computes-rows loop back to back -- the local-map entries at ceil(log2(nodes per block)) bits, the row-offset entries
// at ceil(log2(longest CSR row)) bits -- instead of uint16 / uint8 arrays: P1 tetrahedra 24 -> 12 bytes per instance.
// Field offsets and widths are compile-time constants, so a field is one v_bfe_u32 (two instructions when it straddles words).
// Fixed-point accumulation (experiment, codegen mode suffix "_x<B>"): see codegen.generate_wrapper.
__device__ __forceinline__ void fx_add(double *p, double x, double S) {
    const double t = __builtin_fma(x, S, 6755399441055744.0);          // 1.5 * 2^52: the sum's unit in the last place is 1
    unsigned long long b;
    __builtin_memcpy(&b, &t, 8);
    atomicAdd((unsigned long long *)p, b);
}
__device__ __forceinline__ double fx_get(double acc, double invS) {
    long long a;
    __builtin_memcpy(&a, &acc, 8);
    a = (long long)((unsigned long long)a << 16) >> 16;                   // low 48 bits, sign-extended
    return (double)a * invS;
}

template <int W> __device__ __forceinline__ void load_rec(const unsigned *__restrict__ p, unsigned (&w)[W]) {
    if constexpr (W % 4 == 0) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll
        for (int k = 0; k < W / 4; ++k) { const uint4 v = q[k]; w[4*k] = v.x; w[4*k+1] = v.y; w[4*k+2] = v.z; w[4*k+3] = v.w; }
    } else if constexpr (W % 2 == 0) {
        const uint2 *q = reinterpret_cast<const uint2 *>(p);
#pragma unroll
        for (int k = 0; k < W / 2; ++k) { const uint2 v = q[k]; w[2*k] = v.x; w[2*k+1] = v.y; }
    } else {
#pragma unroll
        for (int k = 0; k < W; ++k) w[k] = p[k];          // (three adjacent words fuse into one global_load_dwordx3)
    }
}
template <int OFF, int BITS, int W> __device__ __forceinline__ int rec_field(const unsigned (&w)[W]) {
    constexpr int I = OFF >> 5, S = OFF & 31;
    constexpr unsigned M = BITS >= 32 ? 0xffffffffu : ((1u << BITS) - 1u);
    if constexpr (S + BITS <= 32) return (int)((w[I] >> S) & M);
    else return (int)(((w[I] >> S) | (w[I + 1] << (32 - S))) & M);
}

}  // namespace fdw

#include "fd_callables.h"
