// fd_locality.hip -- backend-derived locality orders (include/fdhip.h: fd_locality_order, fd_first_touch_order).
//
// The staged and owner-computes-rows wrappers want blocks of entities / rows that are compact in space.  The reference
// gets its locality from DMPlex: cells reordered by reverse Cuthill-McKee, DoFs numbered in order of first appearance
// while walking the cells' closures (firedrake/mesh.py:1214-1228, firedrake/cython/dmcommon.pyx:2599-2729) -- good enough
// for a cache, not for a workgroup that must hold its working set in 64 KiB of LDS.  Execution order is free under the
// wrapper's semantics (builder.py:734-741 is a plain loop over independent entities; INC is commutative), so the
// backend derives its own:
//
//   * entities: sorted by the Morton key of the centroid of their nodes, taken from a position field -- the coordinate
//     argument every TSFC kernel receives (tsfc/kernel_interface/firedrake_loopy.py:432-522);
//   * rows: the reference's own first-touch rule applied to THAT entity order.
//
// Both are private re-encodings: Dats, Maps and the CSR keep the caller's numbering.
#include "fd_common.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>

namespace {

inline int lo_grid(int64_t n) { int64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 256 * 64) g = 256 * 64; return (int)g; }

// bounding box of the positions the entities reference: box[0..2] = min, box[3..5] = max
__global__ void lo_bbox(const int32_t *__restrict__ map, int arity, int64_t e0, int64_t n, const double *__restrict__ pos, int pdim,
                        double *__restrict__ box) {
    double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n * arity; t += (int64_t)gridDim.x * blockDim.x) {
        const int32_t g = map[e0 * arity + t];
        if (g < 0) continue;
        for (int c = 0; c < pdim; ++c) { const double v = pos[(int64_t)g * pdim + c]; lo[c] = fmin(lo[c], v); hi[c] = fmax(hi[c], v); }
    }
    for (int c = 0; c < pdim; ++c) {
        for (int d = 32; d > 0; d >>= 1) { lo[c] = fmin(lo[c], __shfl_xor(lo[c], d, 64)); hi[c] = fmax(hi[c], __shfl_xor(hi[c], d, 64)); }
        if ((threadIdx.x & 63) == 0) { atomicMin(&box[c], lo[c]); atomicMax(&box[3 + c], hi[c]); }
    }
}

__device__ __forceinline__ uint64_t spread16(uint32_t v, int pdim) {       // bit i of v -> bit i*pdim
    uint64_t r = 0;
    for (int i = 0; i < 16; ++i) r |= (uint64_t)((v >> i) & 1u) << (i * pdim);
    return r;
}

__global__ void lo_keys(const int32_t *__restrict__ map, int arity, int64_t e0, int64_t n, const double *__restrict__ pos, int pdim,
                        const double *__restrict__ box, uint64_t *__restrict__ keys, int32_t *__restrict__ ents) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        double c[3] = {0.0, 0.0, 0.0};
        int cnt = 0;
        for (int i = 0; i < arity; ++i) {
            const int32_t g = map[(e0 + t) * arity + i];
            if (g < 0) continue;
            for (int k = 0; k < pdim; ++k) c[k] += pos[(int64_t)g * pdim + k];
            ++cnt;
        }
        uint64_t key = 0;
        for (int k = 0; k < pdim; ++k) {
            const double w = box[3 + k] - box[k];
            double u = (cnt > 0 && w > 0.0) ? (c[k] / cnt - box[k]) / w : 0.0;
            u = fmin(fmax(u, 0.0), 1.0);
            const uint32_t q = (uint32_t)(u * 65535.0);
            key |= spread16(q, pdim) << k;
        }
        keys[t] = key;
        ents[t] = (int32_t)(e0 + t);
    }
}

// node_key[g] = min over the entities (in `order`) touching node g of their rank
__global__ void ft_min_rank(const int32_t *__restrict__ map, int arity, const int32_t *__restrict__ order, int64_t n,
                            int32_t nnodes, uint32_t *__restrict__ node_key) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n * arity; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = t / arity;
        const int32_t g = map[(int64_t)order[r] * arity + (t - r * arity)];
        if (g >= 0 && g < nnodes) atomicMin(&node_key[g], (uint32_t)r);
    }
}

// 64-bit sort key: (first-touch rank, position of the node inside that entity's row is not needed: ties by node id)
__global__ void ft_keys(const uint32_t *__restrict__ node_key, int32_t nnodes, uint64_t *__restrict__ keys, int32_t *__restrict__ ids) {
    for (int64_t g = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; g < nnodes; g += (int64_t)gridDim.x * blockDim.x) {
        keys[g] = ((uint64_t)node_key[g] << 32) | (uint32_t)g;
        ids[g] = (int32_t)g;
    }
}

__global__ void ft_invert(const int32_t *__restrict__ plist, int32_t nnodes, int32_t *__restrict__ pinv) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnodes; p += (int64_t)gridDim.x * blockDim.x)
        pinv[plist[p]] = (int32_t)p;
}

template <class K, class V> int sort_pairs(K *keys, V *vals, int64_t n, int end_bit, hipStream_t s) {
    K *k2 = nullptr; V *v2 = nullptr; void *tmp = nullptr;
    FD_HIP(hipMalloc(&k2, (size_t)n * sizeof(K)));
    FD_HIP(hipMalloc(&v2, (size_t)n * sizeof(V)));
    hipcub::DoubleBuffer<K> dk(keys, k2);
    hipcub::DoubleBuffer<V> dv(vals, v2);
    size_t tb = 0;
    FD_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, tb, dk, dv, n, 0, end_bit, s));
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceRadixSort::SortPairs(tmp, tb, dk, dv, n, 0, end_bit, s));
    FD_HIP(hipStreamSynchronize(s));
    if (dv.Current() != vals) FD_HIP(hipMemcpy(vals, dv.Current(), (size_t)n * sizeof(V), hipMemcpyDeviceToDevice));
    FD_HIP(hipFree(k2)); FD_HIP(hipFree(v2)); FD_HIP(hipFree(tmp));
    return 0;
}

}  // namespace

extern "C" {

int fd_locality_order(const int32_t *map_dev, int arity, int32_t start, int32_t end, const double *pos_dev, int pdim,
                      int32_t *order_dev, fd_stream_t s_) {
    if (!map_dev || !pos_dev || !order_dev || arity <= 0 || end < start || pdim < 1 || pdim > 3)
        FD_FAIL("fd_locality_order: bad arguments");
    const int64_t n = (int64_t)end - start;
    if (n == 0) return 0;
    hipStream_t s = fd::st(s_);
    double *box = nullptr;
    uint64_t *keys = nullptr;
    FD_HIP(hipMalloc(&box, 6 * sizeof(double)));
    const double init[6] = {DBL_MAX, DBL_MAX, DBL_MAX, -DBL_MAX, -DBL_MAX, -DBL_MAX};
    FD_HIP(hipMemcpyAsync(box, init, sizeof init, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lo_bbox, dim3(lo_grid(n * arity)), dim3(256), 0, s, map_dev, arity, (int64_t)start, n, pos_dev, pdim, box);
    FD_CHECK_LAUNCH();
    FD_HIP(hipMalloc(&keys, (size_t)n * 8));
    hipLaunchKernelGGL(lo_keys, dim3(lo_grid(n)), dim3(256), 0, s, map_dev, arity, (int64_t)start, n, pos_dev, pdim, box, keys, order_dev);
    FD_CHECK_LAUNCH();
    int rc = sort_pairs<uint64_t, int32_t>(keys, order_dev, n, 16 * pdim, s);
    (void)hipFree(box); (void)hipFree(keys);
    return rc;
}

int fd_first_touch_order(const int32_t *map_dev, int arity, const int32_t *order_dev, int64_t n, int32_t nnodes,
                         int32_t *pinv_dev, int32_t *plist_dev, fd_stream_t s_) {
    if (!map_dev || !order_dev || !pinv_dev || !plist_dev || arity <= 0 || n < 0 || nnodes < 0)
        FD_FAIL("fd_first_touch_order: bad arguments");
    if (nnodes == 0) return 0;
    hipStream_t s = fd::st(s_);
    uint32_t *nk = nullptr;
    uint64_t *keys = nullptr;
    FD_HIP(hipMalloc(&nk, (size_t)nnodes * 4));
    FD_HIP(hipMemsetAsync(nk, 0xff, (size_t)nnodes * 4, s));          // never touched: rank 2^32 - 1 -> sorted last, by node id
    if (n > 0) {
        hipLaunchKernelGGL(ft_min_rank, dim3(lo_grid(n * arity)), dim3(256), 0, s, map_dev, arity, order_dev, n, nnodes, nk);
        FD_CHECK_LAUNCH();
    }
    FD_HIP(hipMalloc(&keys, (size_t)nnodes * 8));
    hipLaunchKernelGGL(ft_keys, dim3(lo_grid(nnodes)), dim3(256), 0, s, nk, nnodes, keys, plist_dev);
    FD_CHECK_LAUNCH();
    int rc = sort_pairs<uint64_t, int32_t>(keys, plist_dev, (int64_t)nnodes, 64, s);
    if (!rc) {
        hipLaunchKernelGGL(ft_invert, dim3(lo_grid(nnodes)), dim3(256), 0, s, plist_dev, nnodes, pinv_dev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { fd::set_error(hipGetErrorString(e)); rc = (int)e; }
        else FD_HIP(hipStreamSynchronize(s));
    }
    (void)hipFree(nk); (void)hipFree(keys);
    return rc;
}

}  // extern "C"
