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

// row starts of CSR value arrays and of block accumulators (include/fdhip.h: fd_nnz_t; pyop2/datatypes.py:6-10 IntType for nnz
// beyond 2^31): everything INSIDE a row or a block stays a 32-bit offset
typedef int64_t fd_nnz_t;

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
// true when the predicate holds in SOME lane of the wavefront: a scalar condition (paired row-sliced instances pick, per wavefront,
// the instantiation of the local kernel that computes only the rows some lane owns)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

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
__device__ __forceinline__ fd_nnz_t csr_find(const fd_nnz_t *__restrict__ rowptr, const int *__restrict__ colidx, int r, int c) {
    fd_nnz_t lo = rowptr[r], hi = rowptr[r + 1] - 1;
    while (lo <= hi) {
        const fd_nnz_t mid = lo + ((hi - lo) >> 1);
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
// ---- checked fixed-point accumulation (codegen mode suffix "_fx", whole-entity owner-computes-rows loops; OPT-IN since round 6:
// FDHIP_OCR_FIXED_POINT=1 -- the sums below are exact sums of contributions ROUNDED to a quantum of the row block's scale, accurate
// relative to the block's largest contribution, not to each row's own entries; the default adds in fp64 like the reference).
// ds_add_u64 runs at ~6 lanes per clock on scattered addresses where ds_add_f64 runs at ~3.3 (profiles/r1i_microbench_lds.txt)
// and the LDS atomics bind the P1 Jacobian.  A contribution x therefore enters the LDS accumulator as the INTEGER
// k = round(x S): the bit pattern of fma(x, S, 1.5 * 2^52) is 0x4338 << 48 plus k in two's complement (|k| < 2^51), and the low
// word of that constant is zero, so removing it costs one 32-bit add on the high word.  The sums are exact and independent of
// the order of the instances.  Every ROW BLOCK keeps its own scale S = 2^(50 - L) in a 32-byte record next to the window
// [2^(L-6), 2^L) its largest |contribution| must fall into: below the upper end |x S| < 2^50 (2^12 contributions fit 63 bits),
// above the lower end the quantum 2^-(50-L) is at most 2^-44 of that contribution (a sum of n is exact to n/2 quanta).  The high
// words of the contributions are folded into two running maxima per lane (unsigned: the largest negative magnitude, signed: the
// largest positive one; NaN / Inf land above every limit); a block whose maximum leaves the window -- or that has no scale yet --
// redoes ITS rows with fp64 atomics inside the same launch (owner-computes-rows: its rows are nobody else's) and rewrites its
// record for the next launch.  No host round trip, no second kernel.
struct fx_block_t { double S, invS; unsigned lim_hi, low_hi; int L; unsigned pad; };
__device__ __forceinline__ void fx_track(double x, unsigned &mu, int &mi) {
    const int hi = __double2hiint(x);
    mu = max(mu, (unsigned)hi);
    mi = max(mi, hi);
}
__device__ __forceinline__ void fx_acc(double *p, double x, double S, unsigned &mu, int &mi) {
    fx_track(x, mu, mi);
    const double t = __builtin_fma(x, S, 6755399441055744.0);          // 1.5 * 2^52: unit in the last place = 1
    long long b;
    __builtin_memcpy(&b, &t, 8);
    b -= 0x4338000000000000LL;
    atomicAdd((unsigned long long *)p, (unsigned long long)b);
}
__device__ __forceinline__ double fx_get(double acc, double invS) {
    long long a;
    __builtin_memcpy(&a, &acc, 8);
    return (double)a * invS;
}
// fold the lanes' maxima into the block's LDS word (one LDS atomic per wavefront); callers put a barrier behind it
__device__ __forceinline__ void fx_block_max(unsigned *smax, unsigned mu, int mi) {
    unsigned am = max(mu & 0x7fffffffu, (unsigned)max(mi, 0));
    for (int d = 32; d > 0; d >>= 1) am = max(am, (unsigned)__shfl_xor((int)am, d, 64));
    if ((threadIdx.x & 63) == 0 && am) atomicMax(smax, am);
}
// true: the block's largest |contribution| (high word bm; 0 = none) lies outside the window of its scale
__device__ __forceinline__ bool fx_outside(const fx_block_t &r, unsigned bm) { return bm != 0u && (bm >= r.lim_hi || bm < r.low_hi); }
// one lane, end of the block: the record the NEXT launch of this block uses (kept while the maximum stays in the window's
// middle, re-derived with H bits of headroom otherwise); stat[0] counts blocks that fell back from a scale, stat[1] blocks that
// had none
template <int H> __device__ __forceinline__ void fx_update(fx_block_t *rec, const fx_block_t &r, unsigned bm, bool fell, unsigned *stat) {
    if (fell) atomicAdd(stat, 1u);
    if (r.S == 0.0) atomicAdd(stat + 1, 1u);
    if (bm == 0u || bm >= 0x7ff00000u) return;                         // nothing accumulated, or not finite: keep
    const int e = (int)((bm >> 20) & 0x7ffu) - 1023 + 1;               // every |contribution| < 2^e
    if (r.S != 0.0 && e <= r.L - 1 && e >= r.L - 5) return;
    const int L = e + H;
    fx_block_t n = {0.0, 0.0, 0u, 0u, 0, 0u};
    if (L > -900 && L < 900) {
        const unsigned long long sb = (unsigned long long)(1023 + 50 - L) << 52, ib = (unsigned long long)(1023 - 50 + L) << 52;
        __builtin_memcpy(&n.S, &sb, 8);
        __builtin_memcpy(&n.invS, &ib, 8);
        n.lim_hi = (unsigned)(L + 1023) << 20;
        n.low_hi = (unsigned)(L - 6 + 1023) << 20;
        n.L = L;
    }
    *rec = n;
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
// uint16 index row carried as raw words (load_rec<ARITY / 2>): entry 2k in the low half of word k
template <int ARITY> __device__ __forceinline__ void unpack_lmap(const unsigned (&w)[ARITY / 2], int (&out)[ARITY]) {
#pragma unroll
    for (int k = 0; k < ARITY / 2; ++k) { out[2 * k] = (int)(w[k] & 0xffffu); out[2 * k + 1] = (int)(w[k] >> 16); }
}
template <int OFF, int BITS, int W> __device__ __forceinline__ int rec_field(const unsigned (&w)[W]) {
    constexpr int I = OFF >> 5, S = OFF & 31;
    constexpr unsigned M = BITS >= 32 ? 0xffffffffu : ((1u << BITS) - 1u);
    if constexpr (S + BITS <= 32) return (int)((w[I] >> S) & M);
    else return (int)(((w[I] >> S) | (w[I + 1] << (32 - S))) & M);
}

}  // namespace fdw

#include "fd_callables.h"
