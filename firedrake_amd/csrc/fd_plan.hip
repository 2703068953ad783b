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
#include <vector>
#include <algorithm>
#include <climits>

struct fd_plan_s {
    int32_t nblocks = 0, max_nd = 0;
    int64_t list_len = 0;
    int32_t *blkoff = nullptr;   // nblocks+1
    int32_t *list = nullptr;     // list_len
    uint16_t *lmap = nullptr;    // (end-start)*arity
    int32_t *bstart = nullptr;   // nblocks+1: first entity of every block (blocks may differ in size)
    int arity = 0, epb = 0;      // epb = largest block
    int32_t start = 0, end = 0;
    int lane_threads = 0;        // >0: lmap rows are stored in lane order (fd_plan_set_lane_order)
};

namespace {

constexpr int PT = 1024;       // threads per plan-builder workgroup
constexpr int MAXCHUNK = 32;   // PT * MAXCHUNK = 32768 map entries per block at most (128 KiB of LDS for the sort)

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
                                                const int32_t *__restrict__ bstart, int n2, int mode,
                                                int32_t *__restrict__ nuniq, const int32_t *__restrict__ blkoff,
                                                int32_t *__restrict__ list, uint16_t *__restrict__ lmap,
                                                int32_t *__restrict__ maxnd, int32_t *__restrict__ err) {
    extern __shared__ int s[];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t e0 = bstart[b];
    const int64_t e1 = bstart[b + 1];
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
            int mid = lo + ((hi - lo) >> 1);
            int m = s[mid];
            if (m == v) { r = mid; break; }
            if (m < v) lo = mid + 1; else hi = mid - 1;
        }
        dst[i] = (uint16_t)r;
    }
}

// Single-pass variant: the distinct nodes of a block are found with an LDS hash set (H = 2 * 2^ceil(log2(entries)) slots, linear
// probing) instead of sorting every map entry -- a block of 2400 tetrahedra has 9700 entries but ~400 distinct nodes --, only the
// distinct ones are sorted, and the list goes to a scratch area at the block's entry offset (compacted by plan_compact once the
// counts are scanned).  One O(entries) pass instead of two O(entries log^2 entries) ones.
__global__ __launch_bounds__(PT) void plan_pass_hash(const int32_t *__restrict__ map, int arity, int32_t start,
                                                     const int32_t *__restrict__ bstart, int H, int hshift,
                                                     int32_t *__restrict__ nuniq, int32_t *__restrict__ tmplist,
                                                     uint16_t *__restrict__ lmap, int32_t *__restrict__ maxnd, int32_t *__restrict__ err) {
    extern __shared__ int s[];
    const int tid = threadIdx.x;
    const int64_t b = blockIdx.x;
    const int64_t e0 = bstart[b];
    const int64_t e1 = bstart[b + 1];
    const int cnt = (int)(e1 - e0) * arity;
    const int32_t *src = map + e0 * arity;
    for (int i = tid; i < H; i += PT) s[i] = INT_MAX;
    __syncthreads();
    for (int i = tid; i < cnt; i += PT) {
        const int v = src[i];
        if (v < 0) { atomicExch(err, 1); continue; }
        unsigned h = ((unsigned)v * 2654435761u) >> hshift;
        while (true) {
            const int old = atomicCAS(&s[h], INT_MAX, v);
            if (old == INT_MAX || old == v) break;
            h = (h + 1) & (unsigned)(H - 1);
        }
    }
    __syncthreads();
    const int chunk = H / PT;          // H >= PT by construction
    int vals[MAXCHUNK];
    int nflag = 0;
    unsigned flags = 0;
    for (int c = 0; c < MAXCHUNK; ++c) {
        if (c < chunk) {
            const int v = s[tid * chunk + c];
            vals[c] = v;
            if (v != INT_MAX) { flags |= 1u << c; ++nflag; }
        }
    }
    int total;
    int pos = block_excl_scan(nflag, &total);   // contains __syncthreads: all reads of s[] done
    for (int c = 0; c < MAXCHUNK; ++c)
        if (c < chunk && (flags >> c & 1u)) s[pos++] = vals[c];
    int u2 = 2;
    while (u2 < total) u2 <<= 1;
    __syncthreads();
    for (int i = total + tid; i < u2; i += PT) s[i] = INT_MAX;
    __syncthreads();
    bitonic_sort_lds(s, u2);
    if (tid == 0) { nuniq[b] = total; atomicMax(maxnd, total); }
    int32_t *dstl = tmplist + (e0 - start) * arity;
    for (int i = tid; i < total; i += PT) dstl[i] = s[i];
    uint16_t *dst = lmap + (e0 - start) * arity;
    for (int i = tid; i < cnt; i += PT) {
        const int v = src[i];
        int lo = 0, hi = total - 1, r = 0;
        while (lo <= hi) {
            const int mid = lo + ((hi - lo) >> 1);
            const int m = s[mid];
            if (m == v) { r = mid; break; }
            if (m < v) lo = mid + 1; else hi = mid - 1;
        }
        dst[i] = (uint16_t)r;
    }
}

__global__ void plan_compact(const int32_t *__restrict__ bstart, int arity, int32_t start, const int32_t *__restrict__ blkoff,
                             const int32_t *__restrict__ tmplist, int32_t *__restrict__ list) {
    const int64_t e0 = bstart[blockIdx.x];
    const int32_t off = blkoff[blockIdx.x], n = blkoff[blockIdx.x + 1] - off;
    const int32_t *src = tmplist + (e0 - start) * arity;
    for (int i = threadIdx.x; i < n; i += blockDim.x) list[off + i] = src[i];
}

// Lane order for T lanes: a block of n entities is cut into T contiguous runs (lane t owns run t; the first
// n%T runs are one longer), and the k-th entity of run t is stored at slot k*T + t.  A wavefront that walks the
// slots with stride T therefore handles, in one trip, entities ~n/T apart -- entities that are neighbours in the
// (locality-preserving) numbering, and so share nodes, no longer meet in one LDS atomic -- while every lane's
// index rows stay coalesced.  slot_of_entity is the inverse used by the builders.
__device__ __forceinline__ int lane_slot_of_entity(int c, int n, int T) {
    const int q = n / T, rem = n - q * T;
    int t, k;
    if (c < rem * (q + 1)) { t = c / (q + 1); k = c - t * (q + 1); }
    else { const int c2 = c - rem * (q + 1); const int d = c2 / q; t = rem + d; k = c2 - d * q; }
    return k * T + t;
}

__global__ void plan_lane_order(const int32_t *__restrict__ bstart, int arity, int32_t start, int T,
                                const uint16_t *__restrict__ in, uint16_t *__restrict__ out) {
    const int e0 = bstart[blockIdx.x], n = bstart[blockIdx.x + 1] - e0;
    const size_t base = (size_t)(e0 - start) * arity;
    for (int i = threadIdx.x; i < n * arity; i += blockDim.x) {
        const int c = i / arity, a = i - c * arity;
        out[base + (size_t)lane_slot_of_entity(c, n, T) * arity + a] = in[base + i];
    }
}

__global__ void plan_uniform_blocks(int32_t start, int32_t end, int epb, int32_t nblocks, int32_t *__restrict__ bstart) {
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b <= nblocks; b += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = (int64_t)start + b * epb;
        bstart[b] = (int32_t)(e < end ? e : end);
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

static int plan_build_hash(fd_plan_s *p, const int32_t *map_dev, int H, hipStream_t s) {
    const int arity = p->arity;
    const int64_t n = (int64_t)p->end - p->start;
    int hshift = 32;
    for (int h = H; h > 1; h >>= 1) --hshift;
    const size_t lds = (size_t)H * sizeof(int);
    if (lds > 48 * 1024)
        FD_HIP(hipFuncSetAttribute((const void *)plan_pass_hash, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int32_t *nuniq = nullptr, *scal = nullptr, *tmpl = nullptr;
    FD_HIP(hipMalloc(&nuniq, (size_t)p->nblocks * 4));
    FD_HIP(hipMalloc(&scal, 8));
    FD_HIP(hipMemsetAsync(scal, 0, 8, s));
    FD_HIP(hipMalloc(&tmpl, (size_t)n * arity * 4));
    FD_HIP(hipMalloc(&p->blkoff, ((size_t)p->nblocks + 1) * 4));
    FD_HIP(hipMalloc(&p->lmap, (size_t)n * arity * 2));
    hipLaunchKernelGGL(plan_pass_hash, dim3(p->nblocks), dim3(PT), lds, s, map_dev, arity, p->start, p->bstart, H, hshift,
                       nuniq, tmpl, p->lmap, scal, scal + 1);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(plan_scan, dim3(1), dim3(PT), 0, s, nuniq, p->blkoff, p->nblocks);
    FD_CHECK_LAUNCH();
    int32_t h[2], total;
    FD_HIP(hipMemcpyAsync(h, scal, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(&total, p->blkoff + p->nblocks, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (h[1]) { (void)hipFree(nuniq); (void)hipFree(scal); (void)hipFree(tmpl);
                FD_FAIL("fd_plan_create: map has negative entries (VALUE_UNDEFINED); use the direct wrapper"); }
    p->max_nd = h[0];
    p->list_len = total;
    FD_HIP(hipMalloc(&p->list, (size_t)(total > 0 ? total : 1) * 4));
    hipLaunchKernelGGL(plan_compact, dim3(p->nblocks), dim3(256), 0, s, p->bstart, arity, p->start, p->blkoff, tmpl, p->list);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(nuniq)); FD_HIP(hipFree(scal)); FD_HIP(hipFree(tmpl));
    return 0;
}

static int plan_build(fd_plan_s *p, const int32_t *map_dev, hipStream_t s) {
    const int arity = p->arity;
    const int64_t n = (int64_t)p->end - p->start;
    int n2 = PT;
    while (n2 < p->epb * arity) n2 <<= 1;
    // hash-set builder when twice the entries of the largest block fit the LDS, else the two-pass sort
    if (2 * n2 <= PT * MAXCHUNK) return plan_build_hash(p, map_dev, 2 * n2, s);
    size_t lds = (size_t)n2 * sizeof(int);
    if (lds > 48 * 1024)
        FD_HIP(hipFuncSetAttribute((const void *)plan_pass, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int32_t *nuniq = nullptr, *scal = nullptr;
    FD_HIP(hipMalloc(&nuniq, (size_t)p->nblocks * 4));
    FD_HIP(hipMalloc(&scal, 8));
    FD_HIP(hipMemsetAsync(scal, 0, 8, s));
    FD_HIP(hipMalloc(&p->blkoff, ((size_t)p->nblocks + 1) * 4));
    hipLaunchKernelGGL(plan_pass, dim3(p->nblocks), dim3(PT), lds, s, map_dev, arity, p->start, p->bstart, n2, 0,
                       nuniq, nullptr, nullptr, nullptr, scal, scal + 1);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(plan_scan, dim3(1), dim3(PT), 0, s, nuniq, p->blkoff, p->nblocks);
    FD_CHECK_LAUNCH();
    int32_t h[2], total;
    FD_HIP(hipMemcpyAsync(h, scal, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(&total, p->blkoff + p->nblocks, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (h[1]) { (void)hipFree(nuniq); (void)hipFree(scal);
                FD_FAIL("fd_plan_create: map has negative entries (VALUE_UNDEFINED); use the direct wrapper"); }
    p->max_nd = h[0];
    p->list_len = total;
    FD_HIP(hipMalloc(&p->list, (size_t)(total > 0 ? total : 1) * 4));
    FD_HIP(hipMalloc(&p->lmap, (size_t)n * arity * 2));
    hipLaunchKernelGGL(plan_pass, dim3(p->nblocks), dim3(PT), lds, s, map_dev, arity, p->start, p->bstart, n2, 1,
                       nuniq, p->blkoff, p->list, p->lmap, scal, scal + 1);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(nuniq));
    FD_HIP(hipFree(scal));
    return 0;
}

static void plan_release(fd_plan_s *p) {
    if (p->blkoff) (void)fd::release(p->blkoff);
    if (p->list) (void)fd::release(p->list);
    if (p->lmap) (void)fd::release(p->lmap);
    if (p->bstart) (void)fd::release(p->bstart);
    delete p;
}

int fd_plan_create(const int32_t *map_dev, int arity, int32_t start, int32_t end, int epb,
                   fd_stream_t s_, fd_plan_t *out) {
    hipStream_t s = fd::st(s_);
    if (arity <= 0 || epb <= 0 || end < start) FD_FAIL("fd_plan_create: bad arguments");
    if ((int64_t)epb * arity > (int64_t)PT * MAXCHUNK)
        FD_FAIL("fd_plan_create: ents_per_block*arity exceeds 32768 map entries per block");
    auto *p = new fd_plan_s;
    p->arity = arity; p->epb = epb; p->start = start; p->end = end;
    int64_t n = (int64_t)end - start;
    p->nblocks = (int32_t)((n + epb - 1) / epb);
    FD_HIP(hipMalloc(&p->bstart, ((size_t)p->nblocks + 1) * 4));
    hipLaunchKernelGGL(plan_uniform_blocks, dim3((p->nblocks + 256) / 256), dim3(256), 0, s, start, end, epb, p->nblocks, p->bstart);
    FD_CHECK_LAUNCH();
    if (p->nblocks == 0) { *out = p; return 0; }
    int rc = plan_build(p, map_dev, s);
    if (rc) { plan_release(p); return rc; }
    *out = p;
    return 0;
}

int fd_plan_create_blocks(const int32_t *map_dev, int arity, const int32_t *block_starts_host, int32_t nblocks,
                          fd_stream_t s_, fd_plan_t *out) {
    hipStream_t s = fd::st(s_);
    if (arity <= 0 || nblocks < 0 || !block_starts_host) FD_FAIL("fd_plan_create_blocks: bad arguments");
    auto *p = new fd_plan_s;
    p->arity = arity; p->nblocks = nblocks;
    p->start = block_starts_host[0]; p->end = block_starts_host[nblocks];
    int mx = 0;
    for (int32_t b = 0; b < nblocks; ++b) {
        int d = block_starts_host[b + 1] - block_starts_host[b];
        if (d < 0) { delete p; FD_FAIL("fd_plan_create_blocks: block starts must be non-decreasing"); }
        if (d > mx) mx = d;
    }
    p->epb = mx > 0 ? mx : 1;
    if ((int64_t)p->epb * arity > (int64_t)PT * MAXCHUNK) { delete p;
        FD_FAIL("fd_plan_create_blocks: a block exceeds 32768 map entries"); }
    FD_HIP(hipMalloc(&p->bstart, ((size_t)nblocks + 1) * 4));
    FD_HIP(hipMemcpyAsync(p->bstart, block_starts_host, ((size_t)nblocks + 1) * 4, hipMemcpyHostToDevice, s));
    FD_HIP(hipStreamSynchronize(s));
    if (nblocks == 0 || p->end == p->start) { *out = p; return 0; }
    int rc = plan_build(p, map_dev, s);
    if (rc) { plan_release(p); return rc; }
    *out = p;
    return 0;
}

int fd_plan_set_lane_order(fd_plan_t p, int lane_threads, fd_stream_t s_) {
    if (!p || lane_threads <= 0) FD_FAIL("fd_plan_set_lane_order: bad arguments");
    if (p->lane_threads) FD_FAIL("fd_plan_set_lane_order: the plan is already in lane order");
    p->lane_threads = lane_threads;
    const int64_t n = (int64_t)p->end - p->start;
    if (p->nblocks == 0 || n == 0) return 0;
    hipStream_t s = fd::st(s_);
    uint16_t *tmp = nullptr;
    FD_HIP(hipMalloc(&tmp, (size_t)n * p->arity * 2));
    hipLaunchKernelGGL(plan_lane_order, dim3(p->nblocks), dim3(256), 0, s, p->bstart, p->arity, p->start, lane_threads, p->lmap, tmp);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(p->lmap));
    p->lmap = tmp;
    return 0;
}

int fd_plan_block_starts(fd_plan_t p, const int32_t **block_starts, int32_t *max_ents_per_block) {
    if (!p) FD_FAIL("fd_plan_block_starts: null plan");
    if (block_starts) *block_starts = p->bstart;
    if (max_ents_per_block) *max_ents_per_block = p->epb;
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
    plan_release(p);
    return 0;
}

}  // extern "C"

// =====================================================================================
// Matrix plans: block-local sparsity for LDS-staged MatPack scatter (fd_matplan_*).
//
// For every plan block, the distinct (local row, local col) pairs touched by the block's
// entities, sorted by (row, col) -- i.e. in CSR order -- with their global CSR position.
// The staged wrapper accumulates the block's element matrices into an LDS array indexed by
// this block-local numbering (ds_add_f64) and then issues one global atomic per distinct
// nonzero of the block instead of one per element-matrix entry (MatSetValuesLocal with
// ADD_VALUES, pyop2/codegen/builder.py:573-625, at block granularity).
// =====================================================================================
#include <hipcub/hipcub.hpp>

struct fd_matplan_s {
    int32_t nblocks = 0, max_nnz = 0, max_rowlen = 0, kbytes = 1;
    int64_t total = 0;
    int32_t *mb_off = nullptr;   // nblocks+1
    int32_t *gpos = nullptr;     // total
    int32_t *lrp = nullptr;      // sum_b (ndr_b + 1), block b starts at blkoff_r[b] + b
    void *kidx = nullptr;        // (end-start)*ar*ac entries of kbytes each
    int32_t *zero_list = nullptr;  // CSR positions NOT written exclusively by one block (shared or untouched)
    int64_t n_zero = 0, n_exclusive = 0;
};

namespace {

__global__ void mp_ent_block(const int32_t *__restrict__ bstart, int32_t start, int32_t *__restrict__ eb) {
    const int32_t b = blockIdx.x;
    for (int32_t e = bstart[b] + threadIdx.x; e < bstart[b + 1]; e += blockDim.x) eb[e - start] = b;
}

__global__ void mp_emit_keys(const uint16_t *__restrict__ lmr, const uint16_t *__restrict__ lmc, int ar, int ac,
                             int64_t nent, const int32_t *__restrict__ eb, uint64_t *__restrict__ keys) {
    const int64_t per = (int64_t)ar * ac, total = nent * per;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = t / per;
        int ij = (int)(t - e * per);
        int i = ij / ac, j = ij - i * ac;
        uint64_t b = (uint64_t)eb[e];
        keys[t] = (b << 32) | ((uint64_t)lmr[e * ar + i] << 16) | (uint64_t)lmc[e * ac + j];
    }
}

__global__ void mp_block_offsets(const uint64_t *__restrict__ keys, int64_t n, int32_t nblocks, int32_t *__restrict__ off,
                                 int32_t *__restrict__ maxnnz) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t <= n; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t b = t < n ? (int32_t)(keys[t] >> 32) : nblocks;
        int32_t bp = t > 0 ? (int32_t)(keys[t - 1] >> 32) : -1;
        for (int32_t bb = bp + 1; bb <= b; ++bb) off[bb] = (int32_t)t;
    }
}

__device__ inline int64_t lower_bound_u64(const uint64_t *__restrict__ a, int64_t lo, int64_t hi, uint64_t key) {
    while (lo < hi) {
        int64_t mid = lo + ((hi - lo) >> 1);
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one workgroup per block: local row pointers, global positions, statistics
__global__ void mp_rows_and_gpos(const uint64_t *__restrict__ keys, const int32_t *__restrict__ mb_off,
                                 const int32_t *__restrict__ blkoff_r, const int32_t *__restrict__ list_r,
                                 const int32_t *__restrict__ blkoff_c, const int32_t *__restrict__ list_c,
                                 const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                                 int32_t *__restrict__ lrp, int32_t *__restrict__ gpos, int32_t *__restrict__ stats) {
    const int64_t b = blockIdx.x;
    const int64_t o0 = mb_off[b], o1 = mb_off[b + 1];
    const int32_t r0 = blkoff_r[b], ndr = blkoff_r[b + 1] - r0, c0 = blkoff_c[b];
    int32_t *mylrp = lrp + r0 + b;
    for (int lr = threadIdx.x; lr <= ndr; lr += blockDim.x) {
        uint64_t key = ((uint64_t)b << 32) | ((uint64_t)lr << 16);
        mylrp[lr] = (int32_t)(lower_bound_u64(keys, o0, o1, key) - o0);
    }
    for (int64_t t = o0 + threadIdx.x; t < o1; t += blockDim.x) {
        uint64_t k = keys[t];
        int lr = (int)((k >> 16) & 0xffff), lc = (int)(k & 0xffff);
        int r = list_r[r0 + lr], c = list_c[c0 + lc];
        const fd_nnz_t pos = fd_csr_find(rowptr, colidx, r, c);
        if (pos < 0) atomicExch(&stats[2], 1);
        gpos[t] = (int32_t)pos;                                // (fd_matplan_create: patterns below 2^31 entries)
    }
    if (threadIdx.x == 0) atomicMax(&stats[0], (int32_t)(o1 - o0));
    __syncthreads();
    for (int lr = threadIdx.x; lr < ndr; lr += blockDim.x) atomicMax(&stats[1], mylrp[lr + 1] - mylrp[lr]);
}

template <class KT>
__global__ void mp_kidx(const uint64_t *__restrict__ keys, const int32_t *__restrict__ mb_off,
                        const int32_t *__restrict__ blkoff_r, const int32_t *__restrict__ lrp,
                        const uint16_t *__restrict__ lmr, const uint16_t *__restrict__ lmc, int ar, int ac, int64_t nent,
                        const int32_t *__restrict__ eb, KT *__restrict__ kidx) {
    const int64_t per = (int64_t)ar * ac, total = nent * per;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = t / per;
        int ij = (int)(t - e * per);
        int i = ij / ac, j = ij - i * ac;
        int64_t b = eb[e];
        int lr = lmr[e * ar + i], lc = lmc[e * ac + j];
        uint64_t key = ((uint64_t)b << 32) | ((uint64_t)lr << 16) | (uint64_t)lc;
        const int32_t *mylrp = lrp + blkoff_r[b] + b;
        int64_t o0 = mb_off[b];
        int64_t p = lower_bound_u64(keys, o0 + mylrp[lr], o0 + mylrp[lr + 1], key);
        kidx[t] = (KT)(p - o0 - mylrp[lr]);
    }
}

__global__ void mp_count(const int32_t *__restrict__ gpos, int64_t n, int32_t *__restrict__ cnt) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[gpos[t]], 1);
}

// exclusive entries (touched by exactly one block) are encoded as ~pos (negative)
__global__ void mp_encode(int32_t *__restrict__ gpos, int64_t n, const int32_t *__restrict__ cnt) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t g = gpos[t];
        if (cnt[g] == 1) gpos[t] = ~g;
    }
}

__global__ void mp_flags(const int32_t *__restrict__ cnt, int64_t nnz, uint8_t *__restrict__ flags) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < nnz; t += (int64_t)gridDim.x * blockDim.x)
        flags[t] = cnt[t] != 1;
}

inline int mp_grid(int64_t n) { int64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 256 * 64) g = 256 * 64; return (int)g; }

}  // namespace

extern "C" {

int fd_matplan_create(fd_plan_t rp, fd_plan_t cp, const fd_nnz_t *rowptr, const int32_t *colidx, int64_t nnz,
                      fd_stream_t s_, fd_matplan_t *out) {
    hipStream_t s = fd::st(s_);
    if (!rp || !cp) FD_FAIL("fd_matplan_create: null plan");
    if (rp->start != cp->start || rp->end != cp->end || rp->nblocks != cp->nblocks)
        FD_FAIL("fd_matplan_create: row and column plans must cover the same blocks");
    if (nnz > 2147483647ll) FD_FAIL("fd_matplan_create: staged matrix plans address the value array with 32-bit places (nnz >= 2^31: use the owner-computes-rows shapes)");
    auto *m = new fd_matplan_s;
    m->nblocks = rp->nblocks;
    const int64_t nent = (int64_t)rp->end - rp->start;
    const int ar = rp->arity, ac = cp->arity;
    const int64_t nkeys = nent * ar * ac;
    if (m->nblocks == 0 || nkeys == 0) { *out = m; return 0; }
    uint64_t *k1 = nullptr, *k2 = nullptr;
    FD_HIP(hipMalloc(&k1, (size_t)nkeys * 8));
    FD_HIP(hipMalloc(&k2, (size_t)nkeys * 8));
    int32_t *eb = nullptr;
    FD_HIP(hipMalloc(&eb, (size_t)nent * 4));
    hipLaunchKernelGGL(mp_ent_block, dim3(m->nblocks), dim3(256), 0, s, rp->bstart, rp->start, eb);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(mp_emit_keys, dim3(mp_grid(nkeys)), dim3(256), 0, s, rp->lmap, cp->lmap, ar, ac, nent, eb, k1);
    FD_CHECK_LAUNCH();
    int bbits = 1; while ((1ll << bbits) < m->nblocks) ++bbits;
    size_t tb = 0;
    hipcub::DoubleBuffer<uint64_t> db(k1, k2);
    FD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, db, nkeys, 0, 32 + bbits, s));
    void *tmp = nullptr;
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tb, db, nkeys, 0, 32 + bbits, s));
    uint64_t *sorted = db.Current(), *uniq = (sorted == k1) ? k2 : k1;
    int64_t *nsel = nullptr;
    FD_HIP(hipMalloc(&nsel, 8));
    size_t tb2 = 0;
    FD_HIP(hipcub::DeviceSelect::Unique(nullptr, tb2, sorted, uniq, nsel, nkeys, s));
    if (tb2 > tb) { FD_HIP(hipFree(tmp)); FD_HIP(hipMalloc(&tmp, tb2)); }
    FD_HIP(hipcub::DeviceSelect::Unique(tmp, tb2, sorted, uniq, nsel, nkeys, s));
    int64_t nu = 0;
    FD_HIP(hipMemcpyAsync(&nu, nsel, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (nu > 2147483647ll) FD_FAIL("fd_matplan_create: too many block-local nonzeros for int32");
    m->total = nu;
    int32_t *stats = nullptr;
    FD_HIP(hipMalloc(&stats, 16));
    FD_HIP(hipMemsetAsync(stats, 0, 16, s));
    FD_HIP(hipMalloc(&m->mb_off, ((size_t)m->nblocks + 1) * 4));
    FD_HIP(hipMalloc(&m->gpos, (size_t)nu * 4));
    FD_HIP(hipMalloc(&m->lrp, ((size_t)rp->list_len + m->nblocks) * 4));
    hipLaunchKernelGGL(mp_block_offsets, dim3(mp_grid(nu + 1)), dim3(256), 0, s, uniq, nu, m->nblocks, m->mb_off, stats);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(mp_rows_and_gpos, dim3(m->nblocks), dim3(256), 0, s, uniq, m->mb_off, rp->blkoff, rp->list, cp->blkoff,
                       cp->list, rowptr, colidx, m->lrp, m->gpos, stats);
    FD_CHECK_LAUNCH();
    int32_t h[4];
    FD_HIP(hipMemcpyAsync(h, stats, 16, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (h[2]) FD_FAIL("fd_matplan_create: an element-matrix entry is not in the sparsity pattern");
    m->max_nnz = h[0];
    m->max_rowlen = h[1];
    m->kbytes = m->max_rowlen <= 255 ? 1 : 2;
    FD_HIP(hipMalloc(&m->kidx, (size_t)nkeys * m->kbytes));
    if (m->kbytes == 1)
        hipLaunchKernelGGL(mp_kidx<uint8_t>, dim3(mp_grid(nkeys)), dim3(256), 0, s, uniq, m->mb_off, rp->blkoff, m->lrp, rp->lmap,
                           cp->lmap, ar, ac, nent, eb, (uint8_t *)m->kidx);
    else
        hipLaunchKernelGGL(mp_kidx<uint16_t>, dim3(mp_grid(nkeys)), dim3(256), 0, s, uniq, m->mb_off, rp->blkoff, m->lrp, rp->lmap,
                           cp->lmap, ar, ac, nent, eb, (uint16_t *)m->kidx);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(k1)); FD_HIP(hipFree(k2)); FD_HIP(hipFree(stats)); FD_HIP(hipFree(eb));
    // exclusivity: a CSR position touched by exactly one block can be written without an atomic; everything
    // else (shared between blocks, or not touched by this loop at all) is listed for the fused zeroing pass
    if (nnz > 0) {
        int32_t *cnt = nullptr;
        uint8_t *flags = nullptr;
        FD_HIP(hipMalloc(&cnt, (size_t)nnz * 4));
        FD_HIP(hipMemsetAsync(cnt, 0, (size_t)nnz * 4, s));
        FD_HIP(hipMalloc(&flags, (size_t)nnz));
        hipLaunchKernelGGL(mp_count, dim3(mp_grid(nu)), dim3(256), 0, s, m->gpos, nu, cnt);
        FD_CHECK_LAUNCH();
        hipLaunchKernelGGL(mp_encode, dim3(mp_grid(nu)), dim3(256), 0, s, m->gpos, nu, cnt);
        FD_CHECK_LAUNCH();
        hipLaunchKernelGGL(mp_flags, dim3(mp_grid(nnz)), dim3(256), 0, s, cnt, nnz, flags);
        FD_CHECK_LAUNCH();
        FD_HIP(hipMalloc(&m->zero_list, (size_t)nnz * 4));
        size_t tb3 = 0;
        hipcub::CountingInputIterator<int32_t> it(0);
        FD_HIP(hipcub::DeviceSelect::Flagged(nullptr, tb3, it, flags, m->zero_list, nsel, nnz, s));
        FD_HIP(hipFree(tmp));
        FD_HIP(hipMalloc(&tmp, tb3 ? tb3 : 8));
        FD_HIP(hipcub::DeviceSelect::Flagged(tmp, tb3, it, flags, m->zero_list, nsel, nnz, s));
        int64_t nz = 0;
        FD_HIP(hipMemcpyAsync(&nz, nsel, 8, hipMemcpyDeviceToHost, s));
        FD_HIP(hipStreamSynchronize(s));
        m->n_zero = nz;
        m->n_exclusive = nnz - nz;
        FD_HIP(hipFree(cnt)); FD_HIP(hipFree(flags));
    }
    FD_HIP(hipFree(tmp)); FD_HIP(hipFree(nsel));
    *out = m;
    return 0;
}

int fd_matplan_zero_list(fd_matplan_t m, const int32_t **zero_list, int64_t *n_zero, int64_t *n_exclusive) {
    if (!m) FD_FAIL("fd_matplan_zero_list: null plan");
    if (zero_list) *zero_list = m->zero_list;
    if (n_zero) *n_zero = m->n_zero;
    if (n_exclusive) *n_exclusive = m->n_exclusive;
    return 0;
}

int fd_matplan_info(fd_matplan_t m, int32_t *max_nnz, int32_t *max_rowlen, int32_t *kbytes, int64_t *total) {
    if (!m) FD_FAIL("fd_matplan_info: null plan");
    if (max_nnz) *max_nnz = m->max_nnz;
    if (max_rowlen) *max_rowlen = m->max_rowlen;
    if (kbytes) *kbytes = m->kbytes;
    if (total) *total = m->total;
    return 0;
}

int fd_matplan_arrays(fd_matplan_t m, const int32_t **mb_off, const int32_t **gpos, const int32_t **lrp, const void **kidx) {
    if (!m) FD_FAIL("fd_matplan_arrays: null plan");
    if (mb_off) *mb_off = m->mb_off;
    if (gpos) *gpos = m->gpos;
    if (lrp) *lrp = m->lrp;
    if (kidx) *kidx = m->kidx;
    return 0;
}

int fd_matplan_free(fd_matplan_t m) {
    if (!m) return 0;
    if (m->mb_off) FD_HIP(fd::release(m->mb_off));
    if (m->gpos) FD_HIP(fd::release(m->gpos));
    if (m->lrp) FD_HIP(fd::release(m->lrp));
    if (m->kidx) FD_HIP(fd::release(m->kidx));
    if (m->zero_list) FD_HIP(fd::release(m->zero_list));
    delete m;
    return 0;
}

}  // extern "C"

// =====================================================================================
// Owner-computes-rows plans (fd_ocr_*): matrix assembly without global atomics.
//
// The row nodes are cut into contiguous blocks; block b assembles the COMPLETE rows of its nodes by
// visiting every entity that touches one of them (entities on block borders are visited by several
// blocks -- redundant local-kernel work, ~1.4-1.8x for P1 tets).  The rows of a block are contiguous in
// the CSR value array, so after the LDS reduction the block writes them with plain coalesced stores:
// no global atomics, and a pending Mat.zero() needs no memset at all.  This is the block-granular form
// of the "redundant ghost-cell compute, owner computes rows" option of SURVEY.md 8e, applied inside one
// GPU; it replaces MatSetValuesLocal(ADD_VALUES) + MatAssembly (builder.py:573-625, mat.py:940-954).
// =====================================================================================
struct fd_ocrplan_s {
    int32_t nblocks = 0, max_inst = 0;
    int64_t ninst = 0;
    int32_t *inst_off = nullptr;     // nblocks+1 (device)
    int32_t *inst_off_host = nullptr;
    int32_t *inst_ent = nullptr;     // ninst (device): entity of every instance
    int32_t *rblk = nullptr;         // nblocks+1 (device): first row node of every block
    // optional backend-derived row order (fd_first_touch_order): the blocks are then ranges of row POSITIONS;
    // pinv[node] = position for node < npos (borrowed, caller keeps it alive), prowptr = CSR row starts in position order
    const int32_t *pinv = nullptr;
    const fd_nnz_t *prowptr = nullptr;
    int32_t npos = 0;
    // row-sliced plans (fd_ocrplan_create_sliced): an instance is (entity, local row); the lists are padded so that every
    // 64 consecutive slots hold ONE local row index (chunk_role), valid[t] = 0 marks the padding slots
    uint8_t *chunk_role = nullptr;
    uint8_t *valid = nullptr;
    int64_t nreal = 0;
    int sliced_ar = 0;
    // row GROUPS of a sliced plan: an instance is (entity, group); a group names one local row (fd_ocrplan_create_sliced: group i =
    // row i) or two (fd_ocrplan_create_paired: both rows share one evaluation of the local kernel).  groles[2g], groles[2g+1]
    // (device; 255 = none); rows_per_inst = 1 or 2 = rows of the per-instance tables
    int ngroups = 0, rows_per_inst = 1;
    uint8_t *groles = nullptr;
    int32_t *chunk_block = nullptr;  // row block of every 64-slot chunk (the table builders look a block up per chunk, not per entry)
};

namespace {

// row position of a node under an optional row order (identity when pinv == nullptr; -1 = not an owned row)
__device__ __forceinline__ int32_t row_position(const int32_t *__restrict__ pinv, int32_t npos, int32_t node) {
    if (!pinv) return node;
    return (node >= 0 && node < npos) ? pinv[node] : -1;
}

__device__ inline int32_t block_of_node(const int32_t *__restrict__ rblk, int32_t nblocks, int32_t node) {
    // largest b with rblk[b] <= node ; node outside [rblk[0], rblk[nblocks]) -> -1
    if (node < rblk[0] || node >= rblk[nblocks]) return -1;
    int lo = 0, hi = nblocks - 1;
    while (lo < hi) {
        int mid = lo + ((hi - lo + 1) >> 1);
        if (rblk[mid] <= node) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void ocr_emit(const int32_t *__restrict__ rmap, int ar, int32_t start, int32_t end,
                         const int32_t *__restrict__ rblk, int32_t nblocks, uint64_t *__restrict__ keys,
                         const int32_t *__restrict__ pinv, int32_t npos) {
    const int64_t total = ((int64_t)end - start) * ar;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = start + t / ar;
        int32_t r = row_position(pinv, npos, rmap[e * ar + (t % ar)]);
        int32_t b = r >= 0 ? block_of_node(rblk, nblocks, r) : -1;
        keys[t] = b >= 0 ? (((uint64_t)b << 32) | (uint64_t)(uint32_t)e) : ~0ull;
    }
}

__global__ void ocr_split(const uint64_t *__restrict__ keys, int64_t n, int32_t nblocks, int32_t *__restrict__ off,
                          int32_t *__restrict__ ent) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t <= n; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t b = t < n ? (int32_t)(keys[t] >> 32) : nblocks;
        int32_t bp = t > 0 ? (int32_t)(keys[t - 1] >> 32) : -1;
        for (int32_t bb = bp + 1; bb <= b; ++bb) off[bb] = (int32_t)t;
        if (t < n) ent[t] = (int32_t)(keys[t] & 0xffffffffu);
    }
}

// Within every block, reorder the instances by a multiplicative permutation i -> (i*P) mod n so that the
// lanes of a wavefront work on entities ~P apart: neighbouring entities share nodes, and lanes that add into
// the SAME LDS accumulator in one ds_add_f64 are serialised by the hardware.
__global__ void ocr_interleave(const int32_t *__restrict__ off, const int32_t *__restrict__ in, int32_t *__restrict__ out, int stride) {
    const int b = blockIdx.x;
    const int o = off[b], n = off[b + 1] - o;
    __shared__ int P;
    if (threadIdx.x == 0) {
        int p = stride;
        for (;; ++p) {
            int a = p, c = n;
            while (c) { int t = a % c; a = c; c = t; }
            if (a == 1 || n <= 1) break;
        }
        P = p;
    }
    __syncthreads();
    const long long pp = P;
    for (int j = threadIdx.x; j < n; j += blockDim.x) out[o + j] = in[o + (int)((j * pp) % n)];
}

// Stencil order: sort the instances of every block by (ownership pattern, signature, first owned row), where the signature hashes which
// of the entity's rows the block owns and the offsets of all its row nodes from the first owned one.  Instances with
// equal signatures are the "same kind of entity at another place" (a tet type of the structured cube split, in the
// same position relative to the tile border): consecutive ones touch consecutive rows, so the 16 lanes of an LDS
// conflict window add into distinct banks, and no two of them share an accumulator in one instruction.
__global__ void ocr_stencil_keys(const int32_t *__restrict__ rmap, int ar, const int32_t *__restrict__ inst_off,
                                 const int32_t *__restrict__ inst_ent, const int32_t *__restrict__ rblk, int32_t nblocks,
                                 int64_t ninst, uint64_t *__restrict__ keys, const int32_t *__restrict__ pinv, int32_t npos) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < ninst; t += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = nblocks - 1;                      // block of instance t: largest b with inst_off[b] <= t
        while (lo < hi) {
            int mid = lo + ((hi - lo + 1) >> 1);
            if (inst_off[mid] <= t) lo = mid; else hi = mid - 1;
        }
        const int32_t n0 = rblk[lo], n1 = rblk[lo + 1];
        const int32_t *row = rmap + (int64_t)inst_ent[t] * ar;
        int32_t first = -1;
        for (int i = 0; i < ar; ++i) { int32_t r = row_position(pinv, npos, row[i]); if (first < 0 && r >= n0 && r < n1) first = r; }
        uint32_t h = 2166136261u;
        for (int i = 0; i < ar; ++i) {
            int32_t r = row_position(pinv, npos, row[i]);
            uint32_t own = (r >= n0 && r < n1) ? 1u : 0u;
            uint32_t d = (uint32_t)(r - first);
            h = (h ^ own) * 16777619u;
            h = (h ^ (d & 0xffffu)) * 16777619u;
            h = (h ^ (d >> 16)) * 16777619u;
        }
        h ^= h >> 16;
        // the ownership pattern is the MAJOR key inside a block (arity <= 8, < 2^24 blocks): the wrapper skips the LDS atomics
        // of a row no lane of the wavefront owns (s_cbranch_execz), and 64 consecutive instances are one wavefront's trip --
        // with the patterns mixed every trip issued all ar*ac atomics (16.1 of 16 measured on C2), grouped it issues the owned
        // rows' only (12.1).  The full 16-bit signature stays the next key: it is what keeps a conflict window regular
        // (with 8 bits of it the bank conflicts rose by 43 %, profiles/r3k_pmc_mask_order.txt)
        if (ar <= 8 && nblocks < (1 << 24)) {
            uint32_t mask = 0;
            for (int i = 0; i < ar; ++i) { const int32_t r = row_position(pinv, npos, row[i]); if (r >= n0 && r < n1) mask |= 1u << i; }
            keys[t] = ((uint64_t)(uint32_t)lo << 40) | ((uint64_t)(mask ^ ((1u << ar) - 1u)) << 32) | ((uint64_t)(h & 0xffffu) << 16)
                      | (uint64_t)((uint32_t)(first - n0) & 0xffffu);                 // fully owned entities first
        } else {
            keys[t] = ((uint64_t)(uint32_t)lo << 32) | ((uint64_t)(h & 0xffffu) << 16) | (uint64_t)((uint32_t)(first - n0) & 0xffffu);
        }
    }
}


__global__ void iota_k(int32_t *__restrict__ out, int64_t n) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) out[t] = (int32_t)t;
}

__global__ void gather_rows_k(const int32_t *__restrict__ src, int arity, const int32_t *__restrict__ idx, int64_t n,
                              int32_t *__restrict__ dst) {
    const int64_t total = n * arity;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t k = t / arity;
        const int32_t r = idx[k];                        // a negative (undefined) index gives an undefined row
        dst[t] = r < 0 ? -1 : src[(int64_t)r * arity + (t - k * arity)];
    }
}

template <class KT>
__global__ void row_offsets_k(const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                              const int32_t *__restrict__ rmap, const int32_t *__restrict__ cmap, int64_t nent, int ar, int ac,
                              KT *__restrict__ out, int32_t *__restrict__ err) {
    const int64_t per = (int64_t)ar * ac, total = nent * per;
    const KT SKIP = (KT)~(KT)0;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = t / per;
        int ij = (int)(t - e * per);
        int i = ij / ac, j = ij - i * ac;
        int r = rmap[e * ar + i], c = cmap[e * ac + j];
        KT v = SKIP;
        if (r >= 0 && c >= 0) {
            const fd_nnz_t pos = fd_csr_find(rowptr, colidx, r, c);
            if (pos >= 0) v = (KT)(pos - rowptr[r]); else atomicExch(err, 1);
        }
        out[t] = v;
    }
}

// ---- row-sliced owner-computes-rows (fd_ocrplan_create_sliced) --------------------------------------------------
// Large element matrices (P2 tets: 10x10, 28-entry rows) leave a 64 KiB row block ~250 rows, i.e. a brick of ~30
// vertices: most entities touching it are border entities computed again by the neighbouring blocks (x2.2-3.4), and
// the 100 accumulator registers per lane cap the occupancy.  The sliced form makes the unit of work (entity, local
// row i): the wrapper instantiates the local kernel once per i and uses row i of its output only, so the compiler
// strips the other rows' arithmetic from that instantiation.  Every instance is then useful whatever the block size
// (only the entity's geometry is recomputed per row), blocks can be small (several resident per CU), and the lane
// needs ~60 registers.  The instances of a block are grouped by i and every group is padded to whole wavefronts, so
// a wavefront executes exactly one instantiation.
__global__ void ocrs_emit(const int32_t *__restrict__ rmap, int ar, int32_t start, int32_t end, const int32_t *__restrict__ rblk,
                          int32_t nblocks, uint64_t *__restrict__ keys, const int32_t *__restrict__ pinv, int32_t npos) {
    const int64_t total = ((int64_t)end - start) * ar;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = start + t / ar;
        const int i = (int)(t % ar);
        const int32_t r = row_position(pinv, npos, rmap[e * ar + i]);
        const int32_t b = r >= 0 ? block_of_node(rblk, nblocks, r) : -1;
        keys[t] = b >= 0 ? (((uint64_t)b << 39) | ((uint64_t)i << 31) | (uint64_t)(uint32_t)e) : ~0ull;
    }
}

// paired groups: one thread per (entity, group) writes TWO key slots -- the instance of the block that owns the group's first row,
// and a second instance when the other row belongs to a different block (a row that no block owns gives none)
__global__ void ocrs_emit_pairs(const int32_t *__restrict__ rmap, int ar, int32_t start, int32_t end, const int32_t *__restrict__ rblk,
                                int32_t nblocks, uint64_t *__restrict__ keys, const int32_t *__restrict__ pinv, int32_t npos,
                                const uint8_t *__restrict__ groles, int ng) {
    const int64_t total = ((int64_t)end - start) * ng;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t e = start + t / ng;
        const int g = (int)(t % ng);
        const int ra = groles[2 * g], rb = groles[2 * g + 1];
        int32_t ba = -1, bb = -1;
        { const int32_t r = row_position(pinv, npos, rmap[e * ar + ra]); if (r >= 0) ba = block_of_node(rblk, nblocks, r); }
        if (rb != 255) { const int32_t r = row_position(pinv, npos, rmap[e * ar + rb]); if (r >= 0) bb = block_of_node(rblk, nblocks, r); }
        // ownership class of an instance (bits 31..32, between group and entity): 0 = the block owns both rows, 1 = the first only,
        // 2 = the second only.  Sorting by it makes the wavefronts of a group (nearly) uniform in what they have to compute.
        uint64_t ca = 1, cb = 2;
        if (ba >= 0 && bb == ba) { ca = 0; bb = -1; }
        else if (ba < 0) { ba = bb; bb = -1; ca = 2; }
        keys[2 * t] = ba >= 0 ? (((uint64_t)ba << 39) | ((uint64_t)g << 33) | (ca << 31) | (uint64_t)(uint32_t)e) : ~0ull;
        keys[2 * t + 1] = bb >= 0 ? (((uint64_t)bb << 39) | ((uint64_t)g << 33) | (cb << 31) | (uint64_t)(uint32_t)e) : ~0ull;
    }
}

// co-ownership counts of the local rows: cnt[a*ar + b] = entities whose rows a and b (a < b) fall into the same row block
__global__ void ocrs_pair_counts_k(const int32_t *__restrict__ rmap, int ar, int32_t start, int32_t end, const int32_t *__restrict__ rblk,
                                   int32_t nblocks, const int32_t *__restrict__ pinv, int32_t npos, unsigned long long *__restrict__ cnt) {
    __shared__ unsigned int sc[32 * 32];
    for (int q = threadIdx.x; q < ar * ar; q += blockDim.x) sc[q] = 0;
    __syncthreads();
    for (int64_t e = start + blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < end; e += (int64_t)gridDim.x * blockDim.x) {
        int32_t blk[32];
        for (int i = 0; i < ar; ++i) {
            const int32_t r = row_position(pinv, npos, rmap[e * ar + i]);
            blk[i] = r >= 0 ? block_of_node(rblk, nblocks, r) : -1;
        }
        for (int a = 0; a < ar; ++a)
            for (int b = a + 1; b < ar; ++b)
                if (blk[a] >= 0 && blk[a] == blk[b]) atomicAdd(&sc[a * ar + b], 1u);
    }
    __syncthreads();
    for (int q = threadIdx.x; q < ar * ar; q += blockDim.x) if (sc[q]) atomicAdd(&cnt[q], (unsigned long long)sc[q]);
}

// segment (block, role) boundaries of the sorted keys: sstart/send per dense segment id b*ar + role; nvalid = number of real keys
// (shift = first bit of the role / group field: 31 for (entity, row) keys, 33 for paired keys whose bits 31..32 hold the ownership class)
__global__ void ocrs_bounds(const uint64_t *__restrict__ keys, int64_t n, int ar, int32_t *__restrict__ sstart, int32_t *__restrict__ send,
                            int64_t *__restrict__ nvalid, int shift) {
    const uint64_t gmask = (1ull << (39 - shift)) - 1ull;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t k = keys[t];
        if (k == ~0ull) continue;
        const uint64_t seg = k >> shift;
        const int64_t id = (int64_t)(seg >> (39 - shift)) * ar + (int64_t)(seg & gmask);
        if (t == 0 || (keys[t - 1] >> shift) != seg) sstart[id] = (int32_t)t;
        if (t + 1 == n || keys[t + 1] == ~0ull || (keys[t + 1] >> shift) != seg) send[id] = (int32_t)(t + 1);
        if (t + 1 == n || keys[t + 1] == ~0ull) *nvalid = t + 1;
    }
}

__global__ void ocrs_padded_counts(const int32_t *__restrict__ sstart, const int32_t *__restrict__ send, int64_t nseg, int64_t *__restrict__ pc) {
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q <= nseg; q += (int64_t)gridDim.x * blockDim.x)
        pc[q] = q < nseg ? (((int64_t)(send[q] - sstart[q]) + 63) & ~(int64_t)63) : 0;
}

// one thread per segment: copy the entities, pad with the last one (valid = 0), name the role of every 64-slot chunk
// interleave > 1: the real instances of a group are stored in the order j -> (j*P) mod cnt, P = the smallest integer >= interleave
// coprime with cnt -- neighbouring entities share rows and columns, and lanes that add into the SAME accumulator in one
// ds_add_f64 are serialised
__device__ inline long long ocrs_stride(int interleave, int32_t cnt) {
    long long P = 1;
    if (interleave > 1 && cnt > 1) {
        for (P = interleave;; ++P) {
            long long a = P, c = cnt;
            while (c) { long long t = a % c; a = c; c = t; }
            if (a == 1) break;
        }
    }
    return P;
}

__global__ void ocrs_fill(const uint64_t *__restrict__ keys, const int32_t *__restrict__ sstart, const int32_t *__restrict__ send,
                          const int64_t *__restrict__ pstart, int64_t nseg, int ar, int32_t *__restrict__ ent, uint8_t *__restrict__ valid,
                          uint8_t *__restrict__ chunk_role, int interleave, int classes) {
    // 64 lanes per segment
    const int lane = threadIdx.x & 63;
    for (int64_t q = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; q < nseg; q += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const int32_t s0 = sstart[q], cnt = send[q] - s0;
        if (cnt <= 0) continue;
        const int64_t o = pstart[q];
        const int32_t pc = (cnt + 63) & ~63;
        const int32_t last = (int32_t)(keys[s0 + cnt - 1] & 0x7fffffffu);
        if (classes) {
            // paired keys: the segment is sorted by ownership class (bits 31..32) first; the stride permutation stays inside a class
            int32_t c1 = cnt, c2 = cnt;                    // first instance of class >= 1 / >= 2
            { int32_t lo = 0, hi = cnt; while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (((keys[s0 + mid] >> 31) & 3u) >= 1u) hi = mid; else lo = mid + 1; } c1 = lo; }
            { int32_t lo = c1, hi = cnt; while (lo < hi) { const int32_t mid = (lo + hi) >> 1; if (((keys[s0 + mid] >> 31) & 3u) >= 2u) hi = mid; else lo = mid + 1; } c2 = lo; }
            const long long P0 = ocrs_stride(interleave, c1), P1 = ocrs_stride(interleave, c2 - c1), P2 = ocrs_stride(interleave, cnt - c2);
            for (int32_t j = lane; j < pc; j += 64) {
                int32_t src = 0;
                if (j < c1) src = (int32_t)((j * P0) % c1);
                else if (j < c2) src = c1 + (int32_t)(((j - c1) * P1) % (c2 - c1));
                else if (j < cnt) src = c2 + (int32_t)(((j - c2) * P2) % (cnt - c2));
                ent[o + j] = j < cnt ? (int32_t)(keys[s0 + src] & 0x7fffffffu) : last;
                valid[o + j] = j < cnt ? 1 : 0;
                if ((j & 63) == 0) chunk_role[(o + j) >> 6] = (uint8_t)(q % ar);
            }
            continue;
        }
        const long long P = ocrs_stride(interleave, cnt);
        for (int32_t j = lane; j < pc; j += 64) {
            ent[o + j] = j < cnt ? (int32_t)(keys[s0 + (int32_t)((j * P) % cnt)] & 0x7fffffffu) : last;
            valid[o + j] = j < cnt ? 1 : 0;
            if ((j & 63) == 0) chunk_role[(o + j) >> 6] = (uint8_t)(q % ar);
        }
    }
}

__global__ void ocrs_chunk_blocks(const int32_t *__restrict__ inst_off, int32_t nblocks, int32_t *__restrict__ chunk_block) {
    // 64 lanes per block: its chunks [inst_off[b] / 64, inst_off[b+1] / 64)
    const int lane = threadIdx.x & 63;
    for (int64_t b = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; b < nblocks; b += ((int64_t)gridDim.x * blockDim.x) >> 6)
        for (int32_t c = (inst_off[b] >> 6) + lane; c < (inst_off[b + 1] >> 6); c += 64) chunk_block[c] = (int32_t)b;
}

__global__ void ocrs_block_offsets(const int64_t *__restrict__ pstart, int32_t nblocks, int ar, int32_t *__restrict__ off) {
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b <= nblocks; b += (int64_t)gridDim.x * blockDim.x)
        off[b] = (int32_t)pstart[b * ar];
}

// per-instance tables of a sliced plan for one pair of lgmaps: slot[t] = offset of the instance's row inside the block's
// accumulator (0xffff: padding slot, or row dropped by the row lgmap), kk[t][j] = position of column cmap[e][j] inside that
// CSR row (all-ones: negative map entry, or column dropped by the column lgmap -- MatSetValuesLocal ignores negative indices)
template <class KT>
__global__ void ocrs_tables_k(const int32_t *__restrict__ inst_off, int32_t nblocks, const int32_t *__restrict__ ent,
                              const uint8_t *__restrict__ valid, const uint8_t *__restrict__ chunk_role, int64_t ninst,
                              const int32_t *__restrict__ rmap, int ar, const int32_t *__restrict__ cmap, int ac,
                              const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                              const fd_nnz_t *__restrict__ acc_by_node, const fd_nnz_t *__restrict__ acc_by_pos,
                              const int32_t *__restrict__ rblk, const int32_t *__restrict__ rlg, const int32_t *__restrict__ clg,
                              uint16_t *__restrict__ slot, uint16_t *__restrict__ rowlen, KT *__restrict__ kk, int32_t *__restrict__ err,
                              int rbs, int cbs, uint8_t *__restrict__ rmask, unsigned long long *__restrict__ cmask,
                              const uint8_t *__restrict__ groles, int NR, const int32_t *__restrict__ pinv, int32_t npos,
                              const int32_t *__restrict__ chunk_block) {
    // rmask != nullptr: per-DOF lgmaps (``unroll``): rlg / clg are indexed by node*bs + component; a node row (column) is
    // dropped as a whole only when all its components are, the per-component bits go to rmask[t] / cmask[t]
    // NR = rows per instance (fd_ocrplan_create_paired: 2): the tables hold NR rows per instance, (t, s) at t*NR + s; a row of the
    // group that another block owns (or that the group does not have) is a dropped row of THIS instance
    const KT SKIP = (KT)~(KT)0;
    const int64_t total = ninst * NR * ac;
    for (int64_t u = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; u < total; u += (int64_t)gridDim.x * blockDim.x) {
        const int64_t ts = u / ac;                    // table row (t, s)
        const int j = (int)(u - ts * ac);
        const int64_t t = ts / NR;
        const int sidx = (int)(ts - t * NR);
        const int role = groles[2 * chunk_role[t >> 6] + sidx];
        const int32_t e = ent[t];
        const int32_t r = role != 255 ? rmap[(int64_t)e * ar + role] : -1;
        unsigned rm = 0;
        if (rmask && r >= 0) { for (int p = 0; p < rbs; ++p) if (!rlg || rlg[(int64_t)r * rbs + p] >= 0) rm |= 1u << p; }
        bool live = valid[t] && r >= 0 && (rmask ? rm != 0 : !(rlg && rlg[r] < 0));
        int lo = 0;
        if (live && (j == 0 || NR > 1)) {
            lo = chunk_block[t >> 6];                  // block of instance t
            if (NR > 1) {
                const int32_t pos = row_position(pinv, npos, r);
                live = pos >= rblk[lo] && pos < rblk[lo + 1];
            }
        }
        if (rmask && j == 0) {
            rmask[t] = live ? (uint8_t)rm : (uint8_t)0;
            unsigned long long cmk = 0;
            for (int jj = 0; jj < ac; ++jj) {
                const int32_t cc = cmap[(int64_t)e * ac + jj];
                for (int q = 0; q < cbs; ++q)
                    if (cc >= 0 && (!clg || clg[(int64_t)cc * cbs + q] >= 0)) cmk |= 1ull << (jj * cbs + q);
            }
            cmask[t] = live ? cmk : 0ull;
        }
        if (j == 0) {
            uint16_t sl = 0xffffu;
            if (live) {
                const fd_nnz_t d = acc_by_node[r] - acc_by_pos[rblk[lo]];
                if (d < 0 || d >= 0xffff) atomicExch(err, 2); else sl = (uint16_t)d;
            }
            slot[ts] = sl;
            if (rowlen) {
                const fd_nnz_t rl = r >= 0 ? rowptr[r + 1] - rowptr[r] : 0;
                if (rl > 0xffff) atomicExch(err, 2);
                rowlen[ts] = (uint16_t)rl;
            }
        }
        KT v = SKIP;
        const int32_t c = cmap[(int64_t)e * ac + j];
        if (live && c >= 0 && (rmask || !(clg && clg[c] < 0))) {
            const fd_nnz_t pos = fd_csr_find(rowptr, colidx, r, c);
            if (pos >= 0) v = (KT)(pos - rowptr[r]); else atomicExch(err, 1);
        }
        kk[u] = v;
    }
}

// ---- bit-packed instance records of an owner-computes-rows plan (fd_ocr_pack_records; read by fdw::rec_field)
struct RecDesc {
    const uint16_t *lmap[8]; int ar[8]; int lbits[8]; int nmaps;
    const void *kidx; int kbytes, nr, nc, kbits, skipdiag, words;
    const uint16_t *extra; int ebits, sentinel, nextra;
};
__global__ void ocr_pack_records_k(RecDesc d, int64_t ninst, uint32_t *__restrict__ out, int32_t *__restrict__ err) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < ninst; t += (int64_t)gridDim.x * blockDim.x) {
        uint64_t acc = 0; int nb = 0, wi = 0;
        uint32_t *o = out + t * d.words;
        auto put = [&](uint32_t v, int bits) {
            if (bits < 32 && (v >> bits)) atomicOr(err, 1);
            acc |= (uint64_t)v << nb; nb += bits;
            if (nb >= 32) { o[wi++] = (uint32_t)acc; acc >>= 32; nb -= 32; }
        };
        for (int m = 0; m < d.nmaps; ++m)
            for (int i = 0; i < d.ar[m]; ++i) put(d.lmap[m][t * d.ar[m] + i], d.lbits[m]);
        for (int i = 0; i < d.nr; ++i)
            for (int j = 0; j < d.nc; ++j) {
                if (d.skipdiag && i == j) continue;
                const int64_t q = (t * d.nr + i) * d.nc + j;
                uint32_t v = d.kbytes == 1 ? (uint32_t)((const uint8_t *)d.kidx)[q] : (uint32_t)((const uint16_t *)d.kidx)[q];
                if (d.sentinel) {                                   // "dropped" (all ones of the source type) -> all ones of the field
                    const uint32_t all = (1u << d.kbits) - 1u;
                    if (v == (d.kbytes == 1 ? 0xffu : 0xffffu)) v = all; else if (v >= all) atomicOr(err, 1);
                }
                put(v, d.kbits);
            }
        for (int x = 0; d.extra && x < d.nextra; ++x) {
            uint32_t v = d.extra[t * d.nextra + x];
            if (d.sentinel) {
                const uint32_t all = (1u << d.ebits) - 1u;
                if (v == 0xffffu) v = all; else if (v >= all) atomicOr(err, 1);
            }
            put(v, d.ebits);
        }
        if (nb > 0) o[wi++] = (uint32_t)acc;
        while (wi < d.words) o[wi++] = 0;
    }
}

// words[i] |= (position of the diagonal entry in the CSR row of node list[i]) << 20  (8 bits; err |= 1 above 255 or without a diagonal)
__global__ void ocr_node_diag_k(const int32_t *__restrict__ list, int64_t n, int32_t nrows, const fd_nnz_t *__restrict__ rowptr,
                                const int32_t *__restrict__ colidx, uint32_t *__restrict__ words, int32_t *__restrict__ err) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t g = list[i];
        if (g < 0 || g >= nrows) continue;                    // not a row of this matrix (never owned: its word is 0)
        const fd_nnz_t at = fd_csr_find(rowptr, colidx, g, g);
        const int pos = at < 0 ? -1 : (int)(at - rowptr[g]);
        if (pos < 0 || pos > 255) { if (words[i] & 0xfffffu) atomicOr(err, 1); continue; }
        words[i] |= (uint32_t)pos << 20;
    }
}

// One word per (block, staged node): what the owner-computes-rows wrapper keeps in LDS for a node of its block -- bits 0..29 =
// 1 + offset of the node's row inside the block's accumulator (0 = row not owned by this block, or masked by the row lgmap),
// bit 31 = the node's column is masked by the column lgmap.  Precomputed in PLAN order so that the wrapper's staging phase
// streams it instead of gathering a row start and two lgmap entries per node by node id.
__global__ void ocr_node_words_k(const int32_t *__restrict__ blkoff, const int32_t *__restrict__ list, int32_t nblocks,
                                 const int32_t *__restrict__ rblk, const fd_nnz_t *__restrict__ base_by_node,
                                 const fd_nnz_t *__restrict__ start_by_pos, int by_offset, int32_t npos, const int32_t *__restrict__ rlg,
                                 const int32_t *__restrict__ clg, uint32_t *__restrict__ out) {
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int32_t n0 = rblk[b], nown = rblk[b + 1] - n0;
        const fd_nnz_t r0 = start_by_pos[n0], nnzb = start_by_pos[n0 + nown] - r0;
        for (int32_t i = blkoff[b] + threadIdx.x; i < blkoff[b + 1]; i += blockDim.x) {
            const int32_t g = list[i];
            fd_nnz_t p = -1;
            if (by_offset) { if (g >= 0 && g < npos) { p = base_by_node[g] - r0; if (p < 0 || p >= nnzb) p = -1; } }
            else if (g >= n0 && g < n0 + nown) p = base_by_node[g] - r0;
            uint32_t w = (p >= 0 && !(rlg && rlg[g] < 0)) ? (uint32_t)(p + 1) : 0u;
            if (clg && clg[g] < 0) w |= 0x80000000u;
            out[i] = w;
        }
    }
}

}  // namespace

extern "C" {

int fd_ocr_node_words(const int32_t *blkoff_dev, const int32_t *list_dev, int32_t nblocks, const int32_t *rblk_dev,
                      const fd_nnz_t *base_by_node_dev, const fd_nnz_t *start_by_pos_dev, int by_offset, int32_t npos,
                      const int32_t *row_lgmap_dev, const int32_t *col_lgmap_dev, uint32_t *out_dev, fd_stream_t s) {
    if (!blkoff_dev || !list_dev || !rblk_dev || !base_by_node_dev || !start_by_pos_dev || !out_dev || nblocks < 0)
        FD_FAIL("fd_ocr_node_words: bad arguments");
    if (nblocks == 0) return 0;
    hipLaunchKernelGGL(ocr_node_words_k, dim3(nblocks < 65536 ? nblocks : 65536), dim3(256), 0, fd::st(s), blkoff_dev, list_dev, nblocks,
                       rblk_dev, base_by_node_dev, start_by_pos_dev, by_offset, npos, row_lgmap_dev, col_lgmap_dev, out_dev);
    FD_CHECK_LAUNCH();
    return 0;
}

static int pack_records(int64_t ninst, int nmaps, const uint16_t *const *lmaps_dev, const int32_t *arities, const int32_t *lbits,
                        const void *kidx_dev, int kbytes, int nr, int nc, int kbits, int skipdiag, const uint16_t *extra_dev, int ebits,
                        int nextra, int sentinel, int words, uint32_t *out_dev, fd_stream_t s_);

int fd_ocr_pack_records(int64_t ninst, int nmaps, const uint16_t *const *lmaps_dev, const int32_t *arities, const int32_t *lbits,
                        const void *kidx_dev, int kbytes, int nr, int nc, int kbits, int skipdiag, const uint16_t *extra_dev, int ebits,
                        int sentinel, int words, uint32_t *out_dev, fd_stream_t s_) {
    return pack_records(ninst, nmaps, lmaps_dev, arities, lbits, kidx_dev, kbytes, nr, nc, kbits, skipdiag, extra_dev, ebits, 1, sentinel,
                        words, out_dev, s_);
}

int fd_ocr_pack_records_rows(int64_t ninst, int nmaps, const uint16_t *const *lmaps_dev, const int32_t *arities, const int32_t *lbits,
                             const void *kidx_dev, int kbytes, int nr, int nc, int kbits, const uint16_t *extra_dev, int ebits,
                             int words, uint32_t *out_dev, fd_stream_t s_) {
    if (!extra_dev) FD_FAIL("fd_ocr_pack_records_rows: the per-row slots are required");
    return pack_records(ninst, nmaps, lmaps_dev, arities, lbits, kidx_dev, kbytes, nr, nc, kbits, 0, extra_dev, ebits, nr, 1, words, out_dev, s_);
}

static int pack_records(int64_t ninst, int nmaps, const uint16_t *const *lmaps_dev, const int32_t *arities, const int32_t *lbits,
                        const void *kidx_dev, int kbytes, int nr, int nc, int kbits, int skipdiag, const uint16_t *extra_dev, int ebits,
                        int nextra, int sentinel, int words, uint32_t *out_dev, fd_stream_t s_) {
    if (ninst < 0 || nextra < 1 || nmaps < 0 || nmaps > 8 || (nmaps && (!lmaps_dev || !arities || !lbits)) || !kidx_dev || !out_dev ||
        (kbytes != 1 && kbytes != 2) || nr <= 0 || nc <= 0 || kbits <= 0 || kbits > 16 || words <= 0 || (extra_dev && (ebits <= 0 || ebits > 16)))
        FD_FAIL("fd_ocr_pack_records: bad arguments");
    RecDesc d{};
    int64_t bits = 0;
    d.nmaps = nmaps;
    for (int m = 0; m < nmaps; ++m) {
        if (!lmaps_dev[m] || arities[m] <= 0 || lbits[m] <= 0 || lbits[m] > 16) FD_FAIL("fd_ocr_pack_records: bad local map");
        d.lmap[m] = lmaps_dev[m]; d.ar[m] = arities[m]; d.lbits[m] = lbits[m];
        bits += (int64_t)arities[m] * lbits[m];
    }
    bits += (int64_t)(nr * nc - (skipdiag ? (nr < nc ? nr : nc) : 0)) * kbits + (extra_dev ? ebits * nextra : 0);
    d.extra = extra_dev; d.ebits = ebits; d.sentinel = sentinel; d.nextra = nextra;
    if ((bits + 31) / 32 != words) FD_FAIL("fd_ocr_pack_records: the fields do not fill the stated number of words");
    d.kidx = kidx_dev; d.kbytes = kbytes; d.nr = nr; d.nc = nc; d.kbits = kbits; d.skipdiag = skipdiag; d.words = words;
    if (ninst == 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *err = nullptr, e = 0;
    FD_HIP(hipMalloc(&err, 4));
    FD_HIP(hipMemsetAsync(err, 0, 4, s));
    hipLaunchKernelGGL(ocr_pack_records_k, dim3(mp_grid(ninst)), dim3(256), 0, s, d, ninst, out_dev, err);
    FD_CHECK_LAUNCH();
    FD_HIP(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    (void)hipFree(err);
    if (e) FD_FAIL("fd_ocr_pack_records: an index does not fit its field");
    return 0;
}

int fd_ocr_node_diag(const int32_t *list_dev, int64_t n, int32_t nrows, const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev,
                     uint32_t *words_dev, fd_stream_t s_) {
    if (n < 0 || (n && (!list_dev || !rowptr_dev || !colidx_dev || !words_dev))) FD_FAIL("fd_ocr_node_diag: bad arguments");
    if (n == 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *err = nullptr, e = 0;
    FD_HIP(hipMalloc(&err, 4));
    FD_HIP(hipMemsetAsync(err, 0, 4, s));
    hipLaunchKernelGGL(ocr_node_diag_k, dim3(mp_grid(n)), dim3(256), 0, s, list_dev, n, nrows, rowptr_dev, colidx_dev, words_dev, err);
    FD_CHECK_LAUNCH();
    FD_HIP(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    (void)hipFree(err);
    if (e) FD_FAIL("fd_ocr_node_diag: an owned row has no diagonal entry within its first 256 columns");
    return 0;
}

static int ocrplan_build(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *row_block_starts_host,
                         int32_t nblocks, int interleave, const int32_t *pinv_dev, int32_t npos, const fd_nnz_t *prowptr_dev,
                         fd_stream_t s_, fd_ocrplan_t *out);

int fd_ocrplan_create(const int32_t *rmap_dev, int ar, int32_t start, int32_t end,
                      const int32_t *row_block_starts_host, int32_t nblocks, int interleave, fd_stream_t s_, fd_ocrplan_t *out) {
    return ocrplan_build(rmap_dev, ar, start, end, row_block_starts_host, nblocks, interleave, nullptr, 0, nullptr, s_, out);
}

int fd_ocrplan_create_ordered(const int32_t *rmap_dev, int ar, int32_t start, int32_t end,
                              const int32_t *pos_block_starts_host, int32_t nblocks, int interleave,
                              const int32_t *pinv_dev, int32_t npos, const fd_nnz_t *prowptr_dev, fd_stream_t s_, fd_ocrplan_t *out) {
    if (!pinv_dev || !prowptr_dev || npos < 0) FD_FAIL("fd_ocrplan_create_ordered: bad arguments");
    return ocrplan_build(rmap_dev, ar, start, end, pos_block_starts_host, nblocks, interleave, pinv_dev, npos, prowptr_dev, s_, out);
}

static int ocrplan_build(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *row_block_starts_host,
                         int32_t nblocks, int interleave, const int32_t *pinv_dev, int32_t npos, const fd_nnz_t *prowptr_dev,
                         fd_stream_t s_, fd_ocrplan_t *out) {
    hipStream_t s = fd::st(s_);
    if (ar <= 0 || nblocks < 0 || end < start || !row_block_starts_host) FD_FAIL("fd_ocrplan_create: bad arguments");
    auto *p = new fd_ocrplan_s;
    p->nblocks = nblocks;
    p->pinv = pinv_dev; p->npos = npos; p->prowptr = prowptr_dev;
    FD_HIP(hipMalloc(&p->rblk, ((size_t)nblocks + 1) * 4));
    FD_HIP(hipMemcpyAsync(p->rblk, row_block_starts_host, ((size_t)nblocks + 1) * 4, hipMemcpyHostToDevice, s));
    FD_HIP(hipMalloc(&p->inst_off, ((size_t)nblocks + 1) * 4));
    p->inst_off_host = (int32_t *)malloc(((size_t)nblocks + 1) * 4);
    const int64_t nkeys = ((int64_t)end - start) * ar;
    if (nblocks == 0 || nkeys == 0) {
        FD_HIP(hipMemsetAsync(p->inst_off, 0, ((size_t)nblocks + 1) * 4, s));
        memset(p->inst_off_host, 0, ((size_t)nblocks + 1) * 4);
        *out = p; return 0;
    }
    uint64_t *k1 = nullptr, *k2 = nullptr;
    FD_HIP(hipMalloc(&k1, (size_t)nkeys * 8));
    FD_HIP(hipMalloc(&k2, (size_t)nkeys * 8));
    hipLaunchKernelGGL(ocr_emit, dim3(mp_grid(nkeys)), dim3(256), 0, s, rmap_dev, ar, start, end, p->rblk, nblocks, k1, p->pinv, p->npos);
    FD_CHECK_LAUNCH();
    size_t tb = 0;
    hipcub::DoubleBuffer<uint64_t> db(k1, k2);
    FD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, db, nkeys, 0, 64, s));
    void *tmp = nullptr;
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tb, db, nkeys, 0, 64, s));
    uint64_t *sorted = db.Current(), *uniq = (sorted == k1) ? k2 : k1;
    int64_t *nsel = nullptr;
    FD_HIP(hipMalloc(&nsel, 8));
    size_t tb2 = 0;
    FD_HIP(hipcub::DeviceSelect::Unique(nullptr, tb2, sorted, uniq, nsel, nkeys, s));
    if (tb2 > tb) { FD_HIP(hipFree(tmp)); FD_HIP(hipMalloc(&tmp, tb2)); }
    FD_HIP(hipcub::DeviceSelect::Unique(tmp, tb2, sorted, uniq, nsel, nkeys, s));
    int64_t nu = 0;
    FD_HIP(hipMemcpyAsync(&nu, nsel, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (nu > 0) {   // drop the "no owner block" sentinel
        uint64_t last;
        FD_HIP(hipMemcpy(&last, uniq + nu - 1, 8, hipMemcpyDeviceToHost));
        if (last == ~0ull) --nu;
    }
    if (nu > 2147483647ll) FD_FAIL("fd_ocrplan_create: too many instances");
    p->ninst = nu;
    FD_HIP(hipMalloc(&p->inst_ent, (size_t)(nu > 0 ? nu : 1) * 4));
    hipLaunchKernelGGL(ocr_split, dim3(mp_grid(nu + 1)), dim3(256), 0, s, uniq, nu, nblocks, p->inst_off, p->inst_ent);
    FD_CHECK_LAUNCH();
    if (interleave == 1 && nu > 0) {
        // stencil order (-1: groups by shape, not by ownership pattern): (block | signature | first owned row) keys, one stable radix sort of (key, entity) pairs
        uint64_t *ka = nullptr, *kb = nullptr;
        int32_t *vb = nullptr;
        FD_HIP(hipMalloc(&ka, (size_t)nu * 8));
        FD_HIP(hipMalloc(&kb, (size_t)nu * 8));
        FD_HIP(hipMalloc(&vb, (size_t)nu * 4));
        hipLaunchKernelGGL(ocr_stencil_keys, dim3(mp_grid(nu)), dim3(256), 0, s, rmap_dev, ar, p->inst_off, p->inst_ent, p->rblk,
                           nblocks, nu, ka, p->pinv, p->npos);
        FD_CHECK_LAUNCH();
        hipcub::DoubleBuffer<uint64_t> dk(ka, kb);
        hipcub::DoubleBuffer<int32_t> dv(p->inst_ent, vb);
        size_t tb3 = 0;
        FD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb3, dk, dv, (int)nu, 0, 64, s));
        void *tmp3 = nullptr;
        FD_HIP(hipMalloc(&tmp3, tb3 ? tb3 : 8));
        FD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp3, tb3, dk, dv, (int)nu, 0, 64, s));
        FD_HIP(hipStreamSynchronize(s));
        int32_t *sorted_ent = dv.Current(), *other = (sorted_ent == p->inst_ent) ? vb : p->inst_ent;
        p->inst_ent = sorted_ent;
        FD_HIP(hipFree(other)); FD_HIP(hipFree(ka)); FD_HIP(hipFree(kb)); FD_HIP(hipFree(tmp3));
    }
    if (interleave > 1 && nu > 0) {
        int32_t *perm = nullptr;
        FD_HIP(hipMalloc(&perm, (size_t)nu * 4));
        hipLaunchKernelGGL(ocr_interleave, dim3(nblocks), dim3(256), 0, s, p->inst_off, p->inst_ent, perm, interleave);
        FD_CHECK_LAUNCH();
        FD_HIP(hipStreamSynchronize(s));
        FD_HIP(hipFree(p->inst_ent));
        p->inst_ent = perm;
    }
    FD_HIP(hipMemcpyAsync(p->inst_off_host, p->inst_off, ((size_t)nblocks + 1) * 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    for (int32_t b = 0; b < nblocks; ++b) {
        int d = p->inst_off_host[b + 1] - p->inst_off_host[b];
        if (d > p->max_inst) p->max_inst = d;
    }
    FD_HIP(hipFree(k1)); FD_HIP(hipFree(k2)); FD_HIP(hipFree(tmp)); FD_HIP(hipFree(nsel));
    *out = p;
    return 0;
}

int fd_ocrplan_info(fd_ocrplan_t p, int64_t *ninst, int32_t *max_inst_per_block) {
    if (!p) FD_FAIL("fd_ocrplan_info: null plan");
    if (ninst) *ninst = p->ninst;
    if (max_inst_per_block) *max_inst_per_block = p->max_inst;
    return 0;
}

int fd_ocrplan_arrays(fd_ocrplan_t p, const int32_t **inst_off_dev, const int32_t **inst_off_host,
                      const int32_t **inst_entity_dev, const int32_t **row_block_starts_dev) {
    if (!p) FD_FAIL("fd_ocrplan_arrays: null plan");
    if (inst_off_dev) *inst_off_dev = p->inst_off;
    if (inst_off_host) *inst_off_host = p->inst_off_host;
    if (inst_entity_dev) *inst_entity_dev = p->inst_ent;
    if (row_block_starts_dev) *row_block_starts_dev = p->rblk;
    return 0;
}

int fd_ocrplan_free(fd_ocrplan_t p) {
    if (!p) return 0;
    if (p->inst_off) FD_HIP(fd::release(p->inst_off));
    if (p->inst_ent) FD_HIP(fd::release(p->inst_ent));
    if (p->rblk) FD_HIP(fd::release(p->rblk));
    if (p->chunk_role) FD_HIP(fd::release(p->chunk_role));
    if (p->groles) FD_HIP(fd::release(p->groles));
    if (p->chunk_block) FD_HIP(fd::release(p->chunk_block));
    if (p->valid) FD_HIP(fd::release(p->valid));
    free(p->inst_off_host);
    delete p;
    return 0;
}

static int create_sliced(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *block_starts_host,
                         int32_t nblocks, const int32_t *pinv_dev, int32_t npos, int interleave, int ngroups, const uint8_t *groles_host,
                         fd_stream_t s_, fd_ocrplan_t *out);

int fd_ocrplan_create_sliced(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *block_starts_host,
                             int32_t nblocks, const int32_t *pinv_dev, int32_t npos, int interleave, fd_stream_t s_, fd_ocrplan_t *out) {
    if (ar <= 0 || ar > 255) FD_FAIL("fd_ocrplan_create_sliced: bad arguments");
    return create_sliced(rmap_dev, ar, start, end, block_starts_host, nblocks, pinv_dev, npos, interleave, 0, nullptr, s_, out);
}

int fd_ocrplan_create_paired(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *block_starts_host,
                             int32_t nblocks, const int32_t *pinv_dev, int32_t npos, int interleave, int ngroups,
                             const uint8_t *group_rows_host, fd_stream_t s_, fd_ocrplan_t *out) {
    if (ar <= 0 || ar > 254 || ngroups <= 0 || ngroups > ar || ngroups > 63 || !group_rows_host)
        FD_FAIL("fd_ocrplan_create_paired: bad arguments (at most 63 groups)");
    // every local row in exactly one group, first row of a group present
    int seen[256] = {0};
    for (int g = 0; g < ngroups; ++g)
        for (int x = 0; x < 2; ++x) {
            const int r = group_rows_host[2 * g + x];
            if (r == 255 && x == 1) continue;
            if (r >= ar || seen[r]++) FD_FAIL("fd_ocrplan_create_paired: the groups are not a partition of the local rows");
        }
    for (int r = 0; r < ar; ++r) if (!seen[r]) FD_FAIL("fd_ocrplan_create_paired: the groups are not a partition of the local rows");
    return create_sliced(rmap_dev, ar, start, end, block_starts_host, nblocks, pinv_dev, npos, interleave, ngroups, group_rows_host, s_, out);
}

int fd_ocrplan_pair_counts(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *block_starts_host, int32_t nblocks,
                           const int32_t *pinv_dev, int32_t npos, int64_t *counts_host, fd_stream_t s_) {
    if (!rmap_dev || ar <= 0 || ar > 32 || nblocks < 0 || end < start || !block_starts_host || !counts_host)
        FD_FAIL("fd_ocrplan_pair_counts: bad arguments (at most 32 local rows)");
    for (int q = 0; q < ar * ar; ++q) counts_host[q] = 0;
    if (nblocks == 0 || end == start) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *rblk = nullptr;
    unsigned long long *cnt = nullptr;
    FD_HIP(hipMalloc(&rblk, ((size_t)nblocks + 1) * 4));
    FD_HIP(hipMalloc(&cnt, (size_t)ar * ar * 8));
    FD_HIP(hipMemcpyAsync(rblk, block_starts_host, ((size_t)nblocks + 1) * 4, hipMemcpyHostToDevice, s));
    FD_HIP(hipMemsetAsync(cnt, 0, (size_t)ar * ar * 8, s));
    hipLaunchKernelGGL(ocrs_pair_counts_k, dim3(mp_grid((int64_t)end - start)), dim3(256), 0, s, rmap_dev, ar, start, end, rblk, nblocks,
                       pinv_dev, npos, cnt);
    FD_CHECK_LAUNCH();
    FD_HIP(hipMemcpyAsync(counts_host, cnt, (size_t)ar * ar * 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(rblk)); FD_HIP(hipFree(cnt));
    return 0;
}

static int create_sliced(const int32_t *rmap_dev, int ar, int32_t start, int32_t end, const int32_t *block_starts_host,
                         int32_t nblocks, const int32_t *pinv_dev, int32_t npos, int interleave, int ngroups, const uint8_t *groles_host,
                         fd_stream_t s_, fd_ocrplan_t *out) {
    hipStream_t s = fd::st(s_);
    if (!rmap_dev || !out || ar <= 0 || ar > 255 || nblocks < 0 || nblocks >= (1 << 24) || end < start || !block_starts_host)
        FD_FAIL("fd_ocrplan_create_sliced: bad arguments");
    const bool paired = groles_host != nullptr;
    const int ng = paired ? ngroups : ar;
    auto *p = new fd_ocrplan_s;
    p->nblocks = nblocks;
    p->pinv = pinv_dev; p->npos = npos; p->sliced_ar = ar;
    p->ngroups = ng; p->rows_per_inst = paired ? 2 : 1;
    {
        uint8_t roles[512];
        for (int g = 0; g < ng; ++g) { roles[2 * g] = paired ? groles_host[2 * g] : (uint8_t)g; roles[2 * g + 1] = paired ? groles_host[2 * g + 1] : (uint8_t)255; }
        FD_HIP(hipMalloc(&p->groles, (size_t)2 * ng));
        FD_HIP(hipMemcpyAsync(p->groles, roles, (size_t)2 * ng, hipMemcpyHostToDevice, s));
        FD_HIP(hipStreamSynchronize(s));            // (the host copy is a local)
    }
    FD_HIP(hipMalloc(&p->rblk, ((size_t)nblocks + 1) * 4));
    FD_HIP(hipMemcpyAsync(p->rblk, block_starts_host, ((size_t)nblocks + 1) * 4, hipMemcpyHostToDevice, s));
    FD_HIP(hipMalloc(&p->inst_off, ((size_t)nblocks + 1) * 4));
    p->inst_off_host = (int32_t *)calloc((size_t)nblocks + 1, 4);
    const int64_t nkeys = ((int64_t)end - start) * (paired ? 2 * ng : ar);
    if (nblocks == 0 || nkeys == 0) {
        FD_HIP(hipMemsetAsync(p->inst_off, 0, ((size_t)nblocks + 1) * 4, s));
        *out = p; return 0;
    }
    if (nkeys > 2147483647ll || (int64_t)nblocks * ng >= 2147483647ll) FD_FAIL("fd_ocrplan_create_sliced: too many (entity, row) pairs");
    uint64_t *k1 = nullptr, *k2 = nullptr;
    FD_HIP(hipMalloc(&k1, (size_t)nkeys * 8));
    FD_HIP(hipMalloc(&k2, (size_t)nkeys * 8));
    if (paired)
        hipLaunchKernelGGL(ocrs_emit_pairs, dim3(mp_grid(nkeys / 2)), dim3(256), 0, s, rmap_dev, ar, start, end, p->rblk, nblocks, k1, p->pinv,
                           p->npos, p->groles, ng);
    else
        hipLaunchKernelGGL(ocrs_emit, dim3(mp_grid(nkeys)), dim3(256), 0, s, rmap_dev, ar, start, end, p->rblk, nblocks, k1, p->pinv, p->npos);
    FD_CHECK_LAUNCH();
    size_t tb = 0;
    hipcub::DoubleBuffer<uint64_t> db(k1, k2);
    FD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, db, nkeys, 0, 64, s));
    void *tmp = nullptr;
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceRadixSort::SortKeys(tmp, tb, db, nkeys, 0, 64, s));
    const uint64_t *sorted = db.Current();
    const int64_t nseg = (int64_t)nblocks * ng;
    int32_t *sstart = nullptr, *send = nullptr;
    int64_t *pc = nullptr, *pstart = nullptr, *nvalid = nullptr;
    FD_HIP(hipMalloc(&sstart, (size_t)nseg * 4));
    FD_HIP(hipMalloc(&send, (size_t)nseg * 4));
    FD_HIP(hipMalloc(&pc, (size_t)(nseg + 1) * 8));
    FD_HIP(hipMalloc(&pstart, (size_t)(nseg + 1) * 8));
    FD_HIP(hipMalloc(&nvalid, 8));
    FD_HIP(hipMemsetAsync(sstart, 0, (size_t)nseg * 4, s));
    FD_HIP(hipMemsetAsync(send, 0, (size_t)nseg * 4, s));
    FD_HIP(hipMemsetAsync(nvalid, 0, 8, s));
    hipLaunchKernelGGL(ocrs_bounds, dim3(mp_grid(nkeys)), dim3(256), 0, s, sorted, nkeys, ng, sstart, send, nvalid, paired ? 33 : 31);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(ocrs_padded_counts, dim3(mp_grid(nseg + 1)), dim3(256), 0, s, sstart, send, nseg, pc);
    FD_CHECK_LAUNCH();
    size_t tb2 = 0;
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb2, pc, pstart, (int)(nseg + 1), s));
    if (tb2 > tb) { FD_HIP(hipFree(tmp)); FD_HIP(hipMalloc(&tmp, tb2)); }
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb2, pc, pstart, (int)(nseg + 1), s));
    int64_t np_ = 0, nv = 0;
    FD_HIP(hipMemcpyAsync(&np_, pstart + nseg, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(&nv, nvalid, 8, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    if (np_ > 2147483647ll) FD_FAIL("fd_ocrplan_create_sliced: too many instances");
    p->ninst = np_; p->nreal = nv;
    FD_HIP(hipMalloc(&p->inst_ent, (size_t)(np_ > 0 ? np_ : 1) * 4));
    FD_HIP(hipMalloc(&p->valid, (size_t)(np_ > 0 ? np_ : 1)));
    FD_HIP(hipMalloc(&p->chunk_role, (size_t)(np_ / 64 + 1)));
    if (np_ > 0) {
        hipLaunchKernelGGL(ocrs_fill, dim3(mp_grid(nseg * 64)), dim3(256), 0, s, sorted, sstart, send, pstart, nseg, ng, p->inst_ent,
                           p->valid, p->chunk_role, interleave, paired ? 1 : 0);
        FD_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(ocrs_block_offsets, dim3(mp_grid((int64_t)nblocks + 1)), dim3(256), 0, s, pstart, nblocks, ng, p->inst_off);
    FD_CHECK_LAUNCH();
    FD_HIP(hipMalloc(&p->chunk_block, (size_t)(np_ / 64 + 1) * 4));
    hipLaunchKernelGGL(ocrs_chunk_blocks, dim3(mp_grid((int64_t)nblocks * 64)), dim3(256), 0, s, p->inst_off, nblocks, p->chunk_block);
    FD_CHECK_LAUNCH();
    FD_HIP(hipMemcpyAsync(p->inst_off_host, p->inst_off, ((size_t)nblocks + 1) * 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    for (int32_t b = 0; b < nblocks; ++b) {
        int d = p->inst_off_host[b + 1] - p->inst_off_host[b];
        if (d > p->max_inst) p->max_inst = d;
    }
    FD_HIP(hipFree(k1)); FD_HIP(hipFree(k2)); FD_HIP(hipFree(tmp)); FD_HIP(hipFree(sstart)); FD_HIP(hipFree(send));
    FD_HIP(hipFree(pc)); FD_HIP(hipFree(pstart)); FD_HIP(hipFree(nvalid));
    *out = p;
    return 0;
}

int fd_ocrplan_sliced_arrays(fd_ocrplan_t p, const uint8_t **chunk_role_dev, const uint8_t **valid_dev, int64_t *nreal) {
    if (!p || !p->sliced_ar) FD_FAIL("fd_ocrplan_sliced_arrays: not a sliced plan");
    if (chunk_role_dev) *chunk_role_dev = p->chunk_role;
    if (valid_dev) *valid_dev = p->valid;
    if (nreal) *nreal = p->nreal;
    return 0;
}

int fd_ocrplan_sliced_tables(fd_ocrplan_t p, const int32_t *rmap_dev, const int32_t *cmap_dev, int ac, const fd_nnz_t *rowptr_dev,
                             const int32_t *colidx_dev, const fd_nnz_t *acc_by_node_dev, const fd_nnz_t *acc_by_pos_dev,
                             const int32_t *row_lgmap_dev, const int32_t *col_lgmap_dev, int kbytes, uint16_t *slot_out_dev,
                             uint16_t *rowlen_out_dev, void *kk_out_dev, int rbs, int cbs, uint8_t *rowmask_out_dev,
                             uint64_t *colmask_out_dev, fd_stream_t s_) {
    if (!p || !p->sliced_ar || !rmap_dev || !cmap_dev || ac <= 0 || !rowptr_dev || !colidx_dev || !acc_by_node_dev || !acc_by_pos_dev ||
        !slot_out_dev || !kk_out_dev || (kbytes != 1 && kbytes != 2))
        FD_FAIL("fd_ocrplan_sliced_tables: bad arguments");
    if ((rowmask_out_dev != nullptr) != (colmask_out_dev != nullptr) || (rowmask_out_dev && (rbs < 1 || rbs > 8 || cbs < 1 || ac * cbs > 64)))
        FD_FAIL("fd_ocrplan_sliced_tables: per-dof masks need both outputs, rbs <= 8 and carity*cbs <= 64");
    if (p->ninst == 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *err = nullptr;
    FD_HIP(hipMalloc(&err, 4));
    FD_HIP(hipMemsetAsync(err, 0, 4, s));
    const int NR = p->rows_per_inst;
    if (NR > 1 && (rowlen_out_dev || rowmask_out_dev)) FD_FAIL("fd_ocrplan_sliced_tables: paired plans serve scalar matrices with node lgmaps");
    const int64_t total = p->ninst * NR * ac;
    if (kbytes == 1)
        hipLaunchKernelGGL(ocrs_tables_k<uint8_t>, dim3(mp_grid(total)), dim3(256), 0, s, p->inst_off, p->nblocks, p->inst_ent, p->valid,
                           p->chunk_role, p->ninst, rmap_dev, p->sliced_ar, cmap_dev, ac, rowptr_dev, colidx_dev, acc_by_node_dev,
                           acc_by_pos_dev, p->rblk, row_lgmap_dev, col_lgmap_dev, slot_out_dev, rowlen_out_dev, (uint8_t *)kk_out_dev, err,
                           rbs, cbs, rowmask_out_dev, (unsigned long long *)colmask_out_dev, p->groles, NR, p->pinv, p->npos, p->chunk_block);
    else
        hipLaunchKernelGGL(ocrs_tables_k<uint16_t>, dim3(mp_grid(total)), dim3(256), 0, s, p->inst_off, p->nblocks, p->inst_ent, p->valid,
                           p->chunk_role, p->ninst, rmap_dev, p->sliced_ar, cmap_dev, ac, rowptr_dev, colidx_dev, acc_by_node_dev,
                           acc_by_pos_dev, p->rblk, row_lgmap_dev, col_lgmap_dev, slot_out_dev, rowlen_out_dev, (uint16_t *)kk_out_dev, err,
                           rbs, cbs, rowmask_out_dev, (unsigned long long *)colmask_out_dev, p->groles, NR, p->pinv, p->npos, p->chunk_block);
    FD_CHECK_LAUNCH();
    int32_t h = 0;
    FD_HIP(hipMemcpyAsync(&h, err, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(err));
    if (h == 1) FD_FAIL("fd_ocrplan_sliced_tables: an element-matrix entry is not in the sparsity pattern");
    if (h == 2) FD_FAIL("fd_ocrplan_sliced_tables: a row block holds more than 65534 matrix entries");
    return 0;
}

int fd_gather_rows(const int32_t *src_dev, int arity, const int32_t *idx_dev, int64_t n, int32_t *dst_dev, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_k, dim3(mp_grid(n * arity)), dim3(256), 0, fd::st(s), src_dev, arity, idx_dev, n, dst_dev);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_elem_row_offsets(const fd_nnz_t *rowptr, const int32_t *colidx, const int32_t *rmap, const int32_t *cmap,
                            int32_t nent, int ar, int ac, int kbytes, void *out, fd_stream_t s_) {
    if (nent <= 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *err = nullptr;
    FD_HIP(hipMalloc(&err, 4));
    FD_HIP(hipMemsetAsync(err, 0, 4, s));
    const int64_t total = (int64_t)nent * ar * ac;
    if (kbytes == 1)
        hipLaunchKernelGGL(row_offsets_k<uint8_t>, dim3(mp_grid(total)), dim3(256), 0, s, rowptr, colidx, rmap, cmap, (int64_t)nent, ar, ac, (uint8_t *)out, err);
    else if (kbytes == 2)
        hipLaunchKernelGGL(row_offsets_k<uint16_t>, dim3(mp_grid(total)), dim3(256), 0, s, rowptr, colidx, rmap, cmap, (int64_t)nent, ar, ac, (uint16_t *)out, err);
    else FD_FAIL("fd_csr_elem_row_offsets: kbytes must be 1 or 2");
    FD_CHECK_LAUNCH();
    int32_t h = 0;
    FD_HIP(hipMemcpyAsync(&h, err, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(err));
    if (h) FD_FAIL("fd_csr_elem_row_offsets: an element-matrix entry is not in the sparsity pattern");
    return 0;
}

}  // extern "C"
