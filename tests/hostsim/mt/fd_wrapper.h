// tests/hostsim/mt/fd_wrapper.h -- TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
//
// Multi-threaded host stand-in for firedrake_amd/csrc/fd_wrapper.h: lets g++ compile a STAGED wrapper kernel emitted
// by firedrake_amd/codegen.py and execute it with one OS thread per lane of a workgroup (workgroups one after the
// other).  __syncthreads() is a real barrier, LDS is a shared buffer, LDS/global atomics are CAS loops -- so the
// staging phase, the lane-ordered main loop, the LDS reduction and the flush run with the same structure (and the
// same races to get right) as on the device.  tests/hostsim.py drives it to check the staged code path against the
// oracle on machines without a GPU.  The device helpers below are restated as plain C++ (same contracts as the real
// header); nothing under firedrake_amd/ can reach this file.
#pragma once
#include <stdint.h>
#include <stddef.h>
typedef int64_t fd_nnz_t;        // firedrake_amd/csrc/fd_wrapper.h
#include <math.h>
#include <barrier>
#include <functional>
#include <thread>
#include <type_traits>
#include <vector>

#define __device__
#define __global__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __shared__ static            /* static LDS arrays: one copy shared by all lanes (workgroups run one at a time) */
#define __align__(x)
#define restrict __restrict__

typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscInt;

struct fd_sim_dim3 { int x, y, z; };
static thread_local fd_sim_dim3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
static fd_sim_dim3 blockDim = {1, 1, 1}, gridDim = {1, 1, 1};

namespace fd_sim {
static std::barrier<> *bar = nullptr;
alignas(16) static unsigned char lds[160 * 1024];        // dynamic LDS of the running workgroup
static double red_scratch[2048];

// run `body` as `nblocks` workgroups of `nthreads` lanes: every OS thread is one lane and walks the workgroups in order
inline void run(int nblocks, int nthreads, const std::function<void()> &body) {
    blockDim.x = nthreads; gridDim.x = nblocks;
    std::barrier<> b(nthreads);
    bar = &b;
    std::vector<std::thread> ts;
    for (int t = 0; t < nthreads; ++t)
        ts.emplace_back([&, t] {
            threadIdx.x = t;
            for (int blk = 0; blk < nblocks; ++blk) {
                blockIdx.x = blk;
                body();
                bar->arrive_and_wait();                    // the next workgroup reuses the LDS buffer
            }
        });
    for (auto &t : ts) t.join();
    bar = nullptr;
}

template <class T, class F> inline void cas_update(T *p, F f) {
    typedef typename std::conditional<sizeof(T) == 8, uint64_t, uint32_t>::type U;
    static_assert(sizeof(T) == 4 || sizeof(T) == 8, "32- or 64-bit atomics");
    U o = __atomic_load_n((U *)p, __ATOMIC_RELAXED);
    for (;;) {
        T old;
        __builtin_memcpy(&old, &o, sizeof(T));
        const T neu = f(old);
        U n;
        __builtin_memcpy(&n, &neu, sizeof(T));
        if (__atomic_compare_exchange_n((U *)p, &o, n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) return;
    }
}
}  // namespace fd_sim

static inline void __syncthreads() { fd_sim::bar->arrive_and_wait(); }

// v_mfma_f64_16x16x4_f64 the way csrc/fd_tensor.h uses it: lane l of a wavefront supplies A[l & 15][l >> 4] and
// B[l >> 4][l & 15] and holds D[(l >> 4) + 4 g][l & 15] in accumulator register g.  Every wavefront of the workgroup must
// issue it in step (true of the templates: no divergent control flow around the MFMA loop), so two workgroup barriers
// stand in for the wavefront's lock step.
struct fd_d4 {
    double v[4];
    double &operator[](int i) { return v[i]; }
    double operator[](int i) const { return v[i]; }
};
namespace fd_sim { static double mfma_a[1024], mfma_b[1024]; }
static inline fd_d4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, fd_d4 c, int, int, int) {
    const int t = threadIdx.x, w0 = t & ~63, l = t & 63;
    fd_sim::mfma_a[t] = a;
    fd_sim::mfma_b[t] = b;
    __syncthreads();
    for (int g = 0; g < 4; ++g) {
        const int row = (l >> 4) + 4 * g, col = l & 15;
        double s = c[g];
        for (int k = 0; k < 4; ++k) s += fd_sim::mfma_a[w0 + k * 16 + row] * fd_sim::mfma_b[w0 + k * 16 + col];
        c[g] = s;
    }
    __syncthreads();
    return c;
}
template <class T> inline T atomicAdd(T *p, T v) { fd_sim::cas_update(p, [v](T o) { return o + v; }); return v; }

namespace fdw {
inline int xcd_block(int bid, int nb) {
    const int q = nb >> 3, r = nb & 7;
    const int x = bid & 7, k = bid >> 3;
    return x * q + (x < r ? x : r) + k;
}
inline int wave_uniform(int v) { return v; }
// (one OS thread per lane: "some lane of the wavefront" is answered conservatively -- the per-lane guards decide)
inline bool wave_any(bool) { return true; }
template <class T> inline void atomic_add(T *p, T v) { fd_sim::cas_update(p, [v](T o) { return o + v; }); }
template <class T> inline void atomic_min(T *p, T v) { fd_sim::cas_update(p, [v](T o) { return v < o ? v : o; }); }
template <class T> inline void atomic_max(T *p, T v) { fd_sim::cas_update(p, [v](T o) { return v > o ? v : o; }); }
inline int wrap_layer(int a, int nl) { return a % nl; }

inline fd_nnz_t csr_find(const fd_nnz_t *rowptr, const int *colidx, int r, int c) {
    for (fd_nnz_t q = rowptr[r]; q < rowptr[r + 1]; ++q)
        if (colidx[q] == c) return q;
    return -1;
}

template <class T> struct OpAdd { static T f(T a, T b) { return a + b; } };
template <class T> struct OpMin { static T f(T a, T b) { return a < b ? a : b; } };
template <class T> struct OpMax { static T f(T a, T b) { return a > b ? a : b; } };

// all lanes call; result valid on lane 0 (same contract as the device version)
template <class T, class Op> inline T block_reduce(T v, T *) {
    T *s = (T *)fd_sim::red_scratch;
    __syncthreads();
    s[threadIdx.x] = v;
    __syncthreads();
    if (threadIdx.x == 0)
        for (int i = 1; i < blockDim.x; ++i) v = Op::f(v, s[i]);
    __syncthreads();
    return v;
}

template <class T> inline T rep_sum(const T *s, int q, int rsh) {
    if (rsh == 0) return s[q];
    const int R = 1 << rsh;
    const T *p = s + ((size_t)q << rsh);
    T v = 0;
    for (int r = 0; r < R; ++r) v += p[r];
    return v;
}

template <class T, int N> inline void load_packed(const T *p, int (&out)[N]) {
    for (int k = 0; k < N; ++k) out[k] = p[k];
}
template <int ARITY> inline void load_lmap(const uint16_t *p, int (&out)[ARITY]) { load_packed<uint16_t, ARITY>(p, out); }
// checked fixed-point accumulation (codegen mode suffix "_fx"), as in csrc/fd_wrapper.h
struct fx_block_t { double S, invS; unsigned lim_hi, low_hi; int L; unsigned pad; };
inline void fx_track(double x, unsigned &mu, int &mi) {
    long long b;
    __builtin_memcpy(&b, &x, 8);
    const int hi = (int)(b >> 32);
    if ((unsigned)hi > mu) mu = (unsigned)hi;
    if (hi > mi) mi = hi;
}
inline void fx_acc(double *p, double x, double S, unsigned &mu, int &mi) {
    fx_track(x, mu, mi);
    const double t = __builtin_fma(x, S, 6755399441055744.0);
    long long b;
    __builtin_memcpy(&b, &t, 8);
    b -= 0x4338000000000000LL;
    atomicAdd((unsigned long long *)p, (unsigned long long)b);
}
inline double fx_get(double acc, double invS) {
    long long a;
    __builtin_memcpy(&a, &acc, 8);
    return (double)a * invS;
}
inline void fx_block_max(unsigned *smax, unsigned mu, int mi) {
    unsigned am = mu & 0x7fffffffu;
    if (mi > 0 && (unsigned)mi > am) am = (unsigned)mi;
    if (am) fd_sim::cas_update(smax, [am](unsigned o) { return am > o ? am : o; });
}
inline bool fx_outside(const fx_block_t &r, unsigned bm) { return bm != 0u && (bm >= r.lim_hi || bm < r.low_hi); }
template <int H> inline void fx_update(fx_block_t *rec, const fx_block_t &r, unsigned bm, bool fell, unsigned *stat) {
    if (fell) stat[0] += 1u;                 // (workgroups run one after the other here)
    if (r.S == 0.0) stat[1] += 1u;
    if (bm == 0u || bm >= 0x7ff00000u) return;
    const int e = (int)((bm >> 20) & 0x7ffu) - 1023 + 1;
    if (r.S != 0.0 && e <= r.L - 1 && e >= r.L - 5) return;
    const int L = e + H;
    fx_block_t n = {0.0, 0.0, 0u, 0u, 0, 0u};
    if (L > -900 && L < 900) {
        n.S = ldexp(1.0, 50 - L);
        n.invS = ldexp(1.0, L - 50);
        n.lim_hi = (unsigned)(L + 1023) << 20;
        n.low_hi = (unsigned)(L - 6 + 1023) << 20;
        n.L = L;
    }
    *rec = n;
}
template <int W> inline void load_rec(const unsigned *p, unsigned (&w)[W]) { for (int k = 0; k < W; ++k) w[k] = p[k]; }
template <int ARITY> inline void unpack_lmap(const unsigned (&w)[ARITY / 2], int (&out)[ARITY]) {
    for (int k = 0; k < ARITY / 2; ++k) { out[2 * k] = (int)(w[k] & 0xffffu); out[2 * k + 1] = (int)(w[k] >> 16); }
}
template <int OFF, int BITS, int W> inline int rec_field(const unsigned (&w)[W]) {
    unsigned long long v = w[OFF >> 5];
    if ((OFF >> 5) + 1 < W) v |= (unsigned long long)w[(OFF >> 5) + 1] << 32;
    return (int)((v >> (OFF & 31)) & ((1ull << BITS) - 1ull));
}
}  // namespace fdw
static inline int min(int a, int b) { return a < b ? a : b; }      // (HIP's device-side integer min, used by the generated staging loops)

#include "../../../oracle/callables.h"
