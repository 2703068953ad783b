// fd_plan.hip -- block-localisation plans (include/fdhip.h: fd_plan_*).
//
// One-off, per (Map, iteration range) device-side preprocessing: the iteration range
// is cut into blocks of `epb` consecutive entities; for each block we compute the
// sorted list of distinct target nodes and a uint16 local index for every map entry.
// The staged wrapper kernels (fd_wrapper.h / codegen.py) then touch HBM once per
// distinct node per block instead of once per map entry, and reduce INC
// contributions in LDS before a single global atomic per node.
//
// Reference counterpart: none (the reference runs one sequential C loop per rank,
// pyop2/codegen/builder.py:734-741).  Semantically the plan is a lossless
// re-encoding of Map.values (pyop2/types/map.py:32-45) restricted to [start,end).
#include "fd_common.h"
#include <climits>

struct fd_plan_s {
    int32_t nblocks = 0, max_nd = 0;
    int64_t list_len = 0;
    int32_t *blkoff = nullptr;   // nblocks+1
    int32_t *list = nullptr;     // list_len
    uint16_t *lmap = nullptr;    // (end-start)*arity
    int arity = 0, epb = 0;
    int32_t start = 0, end = 0;
};

namespace {

constexpr int PT = 1024;       // threads per plan-builder workgroup
constexpr int MAXCHUNK = 16;   // PT * MAXCHUNK = 16384 map entries per block at most

__device__ inline void bitonic_sort_lds(int *s, int n2) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= n2; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < n2; i += PT) {
                int ixj = i ^ j;
                if (ixj > i) {
                    int a = s[i], b = s[ixj];
                    bool up = ((i & k) == 0);
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
}

// exclusive scan of one int per thread over the 1024-thread block; returns total via *total
__device__ inline int block_excl_scan(int v, int *total) {
    __shared__ int wsum[PT / 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int x = v;
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(x, d, 64);
        if (lane >= d) x += y;
    }
    if (lane == 63) wsum[w] = x;
    __syncthreads();
    if (w == 0) {
        int t = lane < PT / 64 ? wsum[lane] : 0;
        for (int d = 1; d < PT / 64; d <<= 1) {
            int y = __shfl_up(t, d, 64);
            if (lane >= d) t += y;
        }
        if (lane < PT / 64) wsum[lane] = t;
    }
    __syncthreads();
    int base = w > 0 ? wsum[w - 1] : 0;
    *total = wsum[PT / 64 - 1];
    int r = base + x - v;
    __syncthreads();
    return r;
}

// mode 0: count distinct nodes per block; mode 1: write node list + local map
__global__ __launch_bounds__(PT) void plan_pass(const int32_t *__restrict__ map, int arity, int32_t start,
                                                int32_t end, int epb, int n2, int mode,
                                                int32_t *__restrict__ nuniq, const int32_t *__restrict__ blkoff,
                                                int32_t *__restrict__ list, uint16_t *__restrict__ lmap,
                                                int32_t *__restrict__ maxnd, int32_t *__restrict__ err) {
    extern __shared__ int s[];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t e0 = start + b * epb;
    const int64_t e1 = (e0 + epb < end) ? e0 + epb : end;
    const int cnt = (int)(e1 - e0) * arity;
    const int32_t *src = map + e0 * arity;
    for (int i = tid; i < n2; i += PT) {
        int v = INT_MAX;
        if (i < cnt) {
            v = src[i];
            if (v < 0) { atomicExch(err, 1); v = INT_MAX; }
        }
        s[i] = v;
    }
    __syncthreads();
    bitonic_sort_lds(s, n2);
    const int chunk = n2 / PT;          // n2 >= PT by construction
    int vals[MAXCHUNK];
    int nflag = 0;
    unsigned flags = 0;
    for (int c = 0; c < MAXCHUNK; ++c) {
        if (c < chunk) {
            int i = tid * chunk + c;
            int v = s[i];
            bool f = (v != INT_MAX) && (i == 0 || s[i - 1] != v);
            vals[c] = v;
            if (f) { flags |= 1u << c; ++nflag; }
        }
    }
    int total;
    int pos = block_excl_scan(nflag, &total);   // contains __syncthreads: all reads of s[] done
    if (mode == 0) {
        if (tid == 0) { nuniq[b] = total; atomicMax(maxnd, total); }
        return;
    }
    const int32_t off = blkoff[b];
    for (int c = 0; c < MAXCHUNK; ++c) {
        if (c < chunk && (flags >> c & 1u)) {
            s[pos] = vals[c];
            list[off + pos] = vals[c];
            ++pos;
        }
    }
    __syncthreads();
    uint16_t *dst = lmap + (e0 - start) * arity;
    for (int i = tid; i < cnt; i += PT) {
        int v = src[i];
        int lo = 0, hi = total - 1, r = 0;
        while (lo <= hi) {
            int mid = (lo + hi) >> 1;
            int m = s[mid];
            if (m == v) { r = mid; break; }
            if (m < v) lo = mid + 1; else hi = mid - 1;
        }
        dst[i] = (uint16_t)r;
    }
}

__global__ __launch_bounds__(PT) void plan_scan(const int32_t *__restrict__ in, int32_t *__restrict__ out, int n) {
    __shared__ int carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += PT) {
        int i = base + threadIdx.x;
        int v = i < n ? in[i] : 0;
        int total;
        int ex = block_excl_scan(v, &total);
        int carry = carry_s;
        if (i < n) out[i] = carry + ex;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) out[n] = carry_s;
}

}  // namespace

extern "C" {

int fd_plan_create(const int32_t *map_dev, int arity, int32_t start, int32_t end, int epb,
                   fd_stream_t s_, fd_plan_t *out) {
    hipStream_t s = fd::st(s_);
    if (arity <= 0 || epb <= 0 || end < start) FD_FAIL("fd_plan_create: bad arguments");
    if ((int64_t)epb * arity > (int64_t)PT * MAXCHUNK)
        FD_FAIL("fd_plan_create: ents_per_block*arity exceeds 16384 map entries per block");
    auto *p = new fd_plan_s;
    p->arity = arity; p->epb = epb; p->start = start; p->end = end;
    int64_t n = (int64_t)end - start;
    p->nblocks = (int32_t)((n + epb - 1) / epb);
    if (p->nblocks == 0) { *out = p; return 0; }
    int n2 = PT;
    while (n2 < epb * arity) n2 <<= 1;
    size_t lds = (size_t)n2 * sizeof(int);
    if (lds > 48 * 1024)
        FD_HIP(hipFuncSetAttribute((const void *)plan_pass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int32_t *nuniq = nullptr, *scal = nullptr;
    FD_HIP(hipMalloc(&nuniq, (size_t)p->nblocks * 4));
    FD_HIP(hipMalloc(&scal, 8));
    FD_HIP(hipMemsetAsync(scal, 0, 8, s));
    FD_HIP(hipMalloc(&p->blkoff, ((size_t)p->nblocks + 1) * 4));
    hipLaunchKernelGGL(plan_pass, dim3(p->nblocks), dim3(PT), lds, s, map_dev, arity, start, end, epb, n2, 0,
                       nuniq, nullptr, nullptr, nullptr, scal, scal + 1);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(plan_scan, dim3(1), dim3(PT), 0, s, nuniq, p->blkoff, p->nblocks);
    FD_CHECK_LAUNCH();
    int32_t h[2], total;
    FD_HIP(hipMemcpyAsync(h, scal, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(&total, p->blkoff + p->nblocks, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (h[1]) { (void)hipFree(nuniq); (void)hipFree(scal); (void)hipFree(p->blkoff); delete p;
                FD_FAIL("fd_plan_create: map has negative entries (VALUE_UNDEFINED); use the direct wrapper"); }
    p->max_nd = h[0];
    p->list_len = total;
    FD_HIP(hipMalloc(&p->list, (size_t)(total > 0 ? total : 1) * 4));
    FD_HIP(hipMalloc(&p->lmap, (size_t)n * arity * 2));
    hipLaunchKernelGGL(plan_pass, dim3(p->nblocks), dim3(PT), lds, s, map_dev, arity, start, end, epb, n2, 1,
                       nuniq, p->blkoff, p->list, p->lmap, scal, scal + 1);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(nuniq));
    FD_HIP(hipFree(scal));
    *out = p;
    return 0;
}

int fd_plan_info(fd_plan_t p, int32_t *nblocks, int32_t *max_nd, int64_t *list_len) {
    if (!p) FD_FAIL("fd_plan_info: null plan");
    if (nblocks) *nblocks = p->nblocks;
    if (max_nd) *max_nd = p->max_nd;
    if (list_len) *list_len = p->list_len;
    return 0;
}

int fd_plan_arrays(fd_plan_t p, const int32_t **blkoff, const int32_t **list, const uint16_t **lmap) {
    if (!p) FD_FAIL("fd_plan_arrays: null plan");
    if (blkoff) *blkoff = p->blkoff;
    if (list) *list = p->list;
    if (lmap) *lmap = p->lmap;
    return 0;
}

int fd_plan_free(fd_plan_t p) {
    if (!p) return 0;
    if (p->blkoff) FD_HIP(hipFree(p->blkoff));
    if (p->list) FD_HIP(hipFree(p->list));
    if (p->lmap) FD_HIP(hipFree(p->lmap));
    delete p;
    return 0;
}

}  // extern "C"
