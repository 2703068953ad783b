// fd_locality.hip -- backend-derived locality orders (include/fdhip.h: fd_locality_order, fd_first_touch_order).
//
// The staged and owner-computes-rows wrappers want blocks of entities / rows that are compact in space.  The reference
// gets its locality from DMPlex: cells reordered by reverse Cuthill-McKee, DoFs numbered in order of first appearance
// while walking the cells' closures (firedrake/mesh.py:1214-1228, firedrake/cython/dmcommon.pyx:2599-2729) -- good enough
// for a cache, not for a workgroup that must hold its working set in 64 KiB of LDS.  Execution order is free under the
// wrapper's semantics (builder.py:734-741 is a plain loop over independent entities; INC is commutative), so the
// backend derives its own:
//
//   * a k-d partition (fd_kd_order) of a point set into leaves of EQUAL population: recursive splits at the population
//     median (in leaf units) along the longest axis of each segment's bounding box.  Leaves are boxes -- the smallest
//     surface, i.e. the fewest nodes per entity and the fewest redundant instances per owned row -- and hold the same
//     number of points whatever the mesh grading (a uniform grid of tiles does not: on a lattice mesh a tile of 6.3 cells
//     per axis holds 6 or 7 node planes, +-30 % rows, profiles/r3c_tile_binning.txt);
//   * the NODES of a position field -- the coordinate argument every TSFC kernel receives
//     (tsfc/kernel_interface/firedrake_loopy.py:432-522) -- are partitioned; an entity joins the lowest leaf among its
//     nodes (fd_group_entities): one leaf's entities = one staged block.  Partitioning the nodes rather than the entity
//     centroids keeps the entities around a node together -- what a producer's mesh tiles do -- so a block touches its
//     leaf plus one layer (0.27 nodes per tetrahedron on the C2 mesh against 0.30 for leaves of centroids, 0.26 for the
//     producer's 8 x 8 x 4 tiles);
//   * rows that are the position field's nodes take the same leaves as owner-computes-rows blocks; rows of another space
//     take the first-touch rule under the entity order (fd_first_touch_order), cut where the entity leaf changes.
// Both are private re-encodings: Dats, Maps and the CSR keep the caller's numbering.
#include "fd_common.h"
#include <hipcub/hipcub.hpp>
#include <cfloat>
#include <vector>
#include <algorithm>

namespace {

inline int lo_grid(int64_t n) { int64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 256 * 64) g = 256 * 64; return (int)g; }

// key[t] = smallest label among the nodes of entity e0 + t (nlabels - 1 if it has no labelled node); counts[key] += 1
__global__ void lo_min_label(const int32_t *__restrict__ map, int arity, int64_t e0, int64_t n, const int32_t *__restrict__ label,
                             int32_t nnodes, int32_t nlabels, uint32_t *__restrict__ keys, int32_t *__restrict__ ents,
                             int32_t *__restrict__ counts) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t best = nlabels - 1;
        for (int i = 0; i < arity; ++i) {
            const int32_t g = map[(e0 + t) * arity + i];
            if (g >= 0 && g < nnodes) { const int32_t l = label[g]; if (l < best) best = l; }
        }
        keys[t] = (uint32_t)best;
        ents[t] = (int32_t)(e0 + t);
        atomicAdd(&counts[best], 1);
    }
}

// ---- k-d partition ----------------------------------------------------------------------------------------------------
// One level: every segment that still holds more than one leaf is sorted along the longest axis of its bounding box and cut
// at the boundary between its two groups of leaves.  Points stay where they are; `idx` (position -> point) is permuted.
struct KdSeg { int32_t start, cleft, child, split; };      // split = 1: two children (child, child + 1), else one (child)

__global__ void kd_bbox(const double *__restrict__ pts, int pdim, const int32_t *__restrict__ idx, const int32_t *__restrict__ seg,
                        int64_t n, double *__restrict__ box /* nseg x 6 */) {
    // segments are contiguous position ranges: a lane walks a run of 16 positions and keeps a private box while the segment
    // stays the same; a wavefront whose lanes all ended in one segment combines its boxes before touching memory
    constexpr int RUN = 16;
    const int64_t nrun = (n + RUN - 1) / RUN;
    for (int64_t r0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x - (threadIdx.x & 63); r0 < nrun; r0 += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = r0 + (threadIdx.x & 63);
        double lo[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, hi[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
        int32_t cur = -1;
        auto flush = [&](int32_t sg) {
            for (int k = 0; k < pdim; ++k) { atomicMin(&box[(int64_t)sg * 6 + k], lo[k]); atomicMax(&box[(int64_t)sg * 6 + 3 + k], hi[k]); }
        };
        if (r < nrun) {
            const int64_t p1 = (r + 1) * RUN < n ? (r + 1) * RUN : n;
            for (int64_t p = r * RUN; p < p1; ++p) {
                const int32_t sg = seg[p];
                if (sg != cur) {
                    if (cur >= 0) flush(cur);
                    for (int k = 0; k < 3; ++k) { lo[k] = DBL_MAX; hi[k] = -DBL_MAX; }
                    cur = sg;
                }
                const double *x = pts + (int64_t)idx[p] * pdim;
                for (int k = 0; k < pdim; ++k) { lo[k] = fmin(lo[k], x[k]); hi[k] = fmax(hi[k], x[k]); }
            }
        }
        const int32_t first = __shfl(cur, 0, 64);
        if (__all(cur == first || cur < 0) && first >= 0) {
            for (int k = 0; k < pdim; ++k)
                for (int d = 32; d > 0; d >>= 1) { lo[k] = fmin(lo[k], __shfl_xor(lo[k], d, 64)); hi[k] = fmax(hi[k], __shfl_xor(hi[k], d, 64)); }
            if ((threadIdx.x & 63) == 0) flush(first);
        } else if (cur >= 0) {
            flush(cur);
        }
    }
}

// key = segment << 32 | position along the segment's split axis quantised to 32 bits (explicitly rounded operations: the
// numpy restatement in tests/helpers.py reproduces the keys bit for bit); segments that are already leaves keep their order
__global__ void kd_keys(const double *__restrict__ pts, int pdim, const int32_t *__restrict__ idx, const int32_t *__restrict__ seg,
                        int64_t n, const double *__restrict__ box, const KdSeg *__restrict__ tab, uint64_t *__restrict__ keys) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = seg[p];
        uint32_t q = 0;
        if (tab[s].split) {
            const double *b = box + (int64_t)s * 6;
            int ax = 0;
            double best = __dsub_rn(b[3], b[0]);
            for (int k = 1; k < pdim; ++k) { const double w = __dsub_rn(b[3 + k], b[k]); if (w > best) { best = w; ax = k; } }
            if (best > 0.0) {
                double u = __ddiv_rn(__dsub_rn(pts[(int64_t)idx[p] * pdim + ax], b[ax]), best);
                u = fmin(fmax(u, 0.0), 1.0);
                q = (uint32_t)__dmul_rn(u, 4294967295.0);
            }
        }
        keys[p] = ((uint64_t)(uint32_t)s << 32) | q;
    }
}

__global__ void kd_relabel(const uint64_t *__restrict__ keys, int64_t n, const KdSeg *__restrict__ tab, int32_t *__restrict__ seg) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) {
        const int32_t s = (int32_t)(keys[p] >> 32);
        const KdSeg t = tab[s];
        seg[p] = t.child + ((t.split && (int32_t)(p - t.start) >= t.cleft) ? 1 : 0);
    }
}

__global__ void kd_iota(int32_t *__restrict__ idx, int32_t *__restrict__ seg, int64_t n) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) { idx[p] = (int32_t)p; seg[p] = 0; }
}

__global__ void kd_leaf_keys(const int32_t *__restrict__ idx, const int32_t *__restrict__ seg, int64_t n, uint64_t *__restrict__ keys) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x)
        keys[p] = ((uint64_t)(uint32_t)seg[p] << 32) | (uint32_t)idx[p];
}

__global__ void kd_finish(const int32_t *__restrict__ idx, int32_t base, int64_t n, int32_t *__restrict__ order) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < n; p += (int64_t)gridDim.x * blockDim.x) order[p] = base + idx[p];
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

__global__ void ft_ranks(const uint64_t *__restrict__ keys, int32_t nnodes, int32_t *__restrict__ rank) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnodes; p += (int64_t)gridDim.x * blockDim.x)
        rank[p] = (int32_t)(uint32_t)(keys[p] >> 32);            // untouched rows: 0xffffffff = -1
}

__global__ void ft_invert(const int32_t *__restrict__ plist, int32_t nnodes, int32_t *__restrict__ pinv) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnodes; p += (int64_t)gridDim.x * blockDim.x)
        pinv[plist[p]] = (int32_t)p;
}

// gpos[prowptr[p] + k] = gstart[p] + k: 16 lanes per row
// (gpos holds 32-bit places: the whole-entity row flush serves patterns below 2^31 entries, fd_row_entry_positions checks)
__global__ void row_entry_positions(int32_t npos, const fd_nnz_t *__restrict__ prowptr, const fd_nnz_t *__restrict__ gstart,
                                    int32_t *__restrict__ gpos) {
    const int sub = threadIdx.x & 15;
    for (int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4; p < npos; p += ((int64_t)gridDim.x * blockDim.x) >> 4) {
        const fd_nnz_t a = prowptr[p], g = gstart[p];
        const int len = (int)(prowptr[p + 1] - a);
        for (int k = sub; k < len; k += 16) gpos[a + k] = (int32_t)(g + k);
    }
}

// the same table with the COLUMN lgmap folded in: an entry whose column is masked (clg[col] < 0: MatSetValuesLocal drops it,
// builder.py:573-625) reads -2 - place -- the flush leaves it alone when it accumulates and stores 0.0 when it overwrites its rows
__global__ void row_entry_positions_masked(int32_t npos, const fd_nnz_t *__restrict__ prowptr, const fd_nnz_t *__restrict__ gstart,
                                           const int32_t *__restrict__ colidx, const int32_t *__restrict__ clg, int32_t *__restrict__ gpos) {
    const int sub = threadIdx.x & 15;
    for (int64_t p = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 4; p < npos; p += ((int64_t)gridDim.x * blockDim.x) >> 4) {
        const fd_nnz_t a = prowptr[p], g = gstart[p];
        const int len = (int)(prowptr[p + 1] - a);
        for (int k = sub; k < len; k += 16) {
            const int32_t place = (int32_t)(g + k);
            gpos[a + k] = clg[colidx[g + k]] < 0 ? -2 - place : place;
        }
    }
}

// ---- run-coded flush tables of a derived row order (fd_ocr_row_runs)
// A row position p starts a RUN when it is the first row of its block or when its displacement (CSR start - accumulator start)
// differs from the previous position's: inside a run, place = accumulator index + one displacement.
__global__ void rr_flags(int32_t npos, const fd_nnz_t *__restrict__ prowptr, const fd_nnz_t *__restrict__ gstart,
                         const int32_t *__restrict__ rblk, int32_t nblocks, int32_t *__restrict__ flag) {
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = t0; p < npos; p += st)
        flag[p] = (p == 0 || gstart[p] - prowptr[p] != gstart[p - 1] - prowptr[p - 1]) ? 1 : 0;
}
__global__ void rr_block_starts(const int32_t *__restrict__ rblk, int32_t nblocks, int32_t npos, int32_t *__restrict__ flag) {
    for (int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * blockDim.x)
        if (rblk[b] < npos) flag[rblk[b]] = 1;
}
// runidx = inclusive scan of flag (1-based run number of every position)
__global__ void rr_tables(int32_t npos, const fd_nnz_t *__restrict__ prowptr, const fd_nnz_t *__restrict__ gstart,
                          const int32_t *__restrict__ flag, const int32_t *__restrict__ runidx, const int32_t *__restrict__ rblk,
                          int32_t nblocks, int32_t *__restrict__ brun, fd_nnz_t *__restrict__ rdelta) {
    const int64_t t0 = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, st = (int64_t)gridDim.x * blockDim.x;
    for (int64_t p = t0; p < npos; p += st)
        if (flag[p]) rdelta[runidx[p] - 1] = gstart[p] - prowptr[p];
    for (int64_t b = t0; b <= nblocks; b += st) {
        const int32_t p = rblk[b];
        brun[b] = p < npos ? runidx[p] - 1 : (npos > 0 ? runidx[npos - 1] : 0);
    }
}
// one workgroup per block: grun[entry] = run of the entry's row, counted from the block's first run; err |= 1 above 255
__global__ void rr_entries(const int32_t *__restrict__ rblk, int32_t nblocks, int32_t npos, const fd_nnz_t *__restrict__ prowptr,
                           const int32_t *__restrict__ runidx, const int32_t *__restrict__ brun, uint8_t *__restrict__ grun,
                           int32_t *__restrict__ err) {
    for (int b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const int32_t n0 = rblk[b], n1 = rblk[b + 1] < npos ? rblk[b + 1] : npos, r0 = brun[b];
        for (int32_t p = n0 + (threadIdx.x >> 4); p < n1; p += blockDim.x >> 4) {
            const fd_nnz_t a = prowptr[p];
            const int32_t len = (int32_t)(prowptr[p + 1] - a), r = runidx[p] - 1 - r0;
            if (r > 255 && (threadIdx.x & 15) == 0) atomicOr(err, 1);
            for (int k = threadIdx.x & 15; k < len; k += 16) grun[a + k] = (uint8_t)r;
        }
    }
}

// ---- tables of a row order (fd_row_order_tables): lengths and CSR starts of the rows in position order, accumulator starts by node
__global__ void ro_gather(int32_t npos, const int32_t *__restrict__ plist, const fd_nnz_t *__restrict__ rowptr,
                          fd_nnz_t *__restrict__ len, fd_nnz_t *__restrict__ gstart) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p <= npos; p += (int64_t)gridDim.x * blockDim.x) {
        if (p == npos) { len[p] = 0; continue; }
        const int32_t r = plist[p];
        const fd_nnz_t a = rowptr[r];
        len[p] = rowptr[r + 1] - a;
        gstart[p] = a;
    }
}
__global__ void ro_nstart(int32_t npos, const int32_t *__restrict__ plist, const fd_nnz_t *__restrict__ prowptr, fd_nnz_t *__restrict__ nstart) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < npos; p += (int64_t)gridDim.x * blockDim.x)
        nstart[plist[p]] = prowptr[p];
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
    if (dk.Current() != keys) FD_HIP(hipMemcpy(keys, dk.Current(), (size_t)n * sizeof(K), hipMemcpyDeviceToDevice));
    FD_HIP(hipFree(k2)); FD_HIP(hipFree(v2)); FD_HIP(hipFree(tmp));
    return 0;
}

}  // namespace

// label[order[i]] = leaf of position i (starts[l] <= i < starts[l + 1]): the node labels fd_group_entities groups entities by
__global__ void lo_leaf_labels(const int32_t *__restrict__ order, int64_t n, const int32_t *__restrict__ starts, int32_t nleaves,
                               int32_t *__restrict__ label) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lo = 0, hi = nleaves - 1;
        while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (starts[mid] <= i) lo = mid; else hi = mid - 1; }
        label[order[i]] = lo;
    }
}

extern "C" {

int fd_leaf_labels(const int32_t *order_dev, int64_t n, const int32_t *leaf_starts_host, int32_t nleaves, int32_t *label_dev, fd_stream_t s_) {
    if (!order_dev || !leaf_starts_host || !label_dev || n < 0 || nleaves < 1) FD_FAIL("fd_leaf_labels: bad arguments");
    if (n == 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *starts = nullptr;
    FD_HIP(hipMalloc(&starts, ((size_t)nleaves + 1) * 4));
    FD_HIP(hipMemcpyAsync(starts, leaf_starts_host, ((size_t)nleaves + 1) * 4, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(lo_leaf_labels, dim3(lo_grid(n)), dim3(256), 0, s, order_dev, n, starts, nleaves, label_dev);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(starts));
    return 0;
}

int fd_group_entities(const int32_t *map_dev, int arity, int32_t start, int32_t end, const int32_t *label_dev, int32_t nnodes,
                      int32_t nlabels, int32_t *order_dev, int32_t *counts_host, fd_stream_t s_) {
    if (!map_dev || !label_dev || !order_dev || !counts_host || arity <= 0 || end < start || nlabels < 1 || nnodes < 0)
        FD_FAIL("fd_group_entities: bad arguments");
    const int64_t n = (int64_t)end - start;
    for (int32_t k = 0; k < nlabels; ++k) counts_host[k] = 0;
    if (n == 0) return 0;
    hipStream_t s = fd::st(s_);
    uint32_t *keys = nullptr;
    int32_t *counts = nullptr;
    FD_HIP(hipMalloc(&keys, (size_t)n * 4));
    FD_HIP(hipMalloc(&counts, (size_t)nlabels * 4));
    FD_HIP(hipMemsetAsync(counts, 0, (size_t)nlabels * 4, s));
    hipLaunchKernelGGL(lo_min_label, dim3(lo_grid(n)), dim3(256), 0, s, map_dev, arity, (int64_t)start, n, label_dev, nnodes, nlabels, keys,
                       order_dev, counts);
    FD_CHECK_LAUNCH();
    int bits = 1;
    while ((1ll << bits) < nlabels) ++bits;
    int rc = sort_pairs<uint32_t, int32_t>(keys, order_dev, n, bits, s);
    if (!rc) {
        hipError_t e = hipMemcpy(counts_host, counts, (size_t)nlabels * 4, hipMemcpyDeviceToHost);
        if (e != hipSuccess) { fd::set_error(hipGetErrorString(e)); rc = (int)e; }
    }
    (void)hipFree(keys); (void)hipFree(counts);
    return rc;
}

int fd_kd_order(const double *pts_dev, int pdim, int64_t n, int32_t base, int32_t leaf_size, int32_t *order_dev,
                int32_t *leaf_starts_host, int32_t max_leaves, int32_t *nleaves_out, fd_stream_t s_) {
    if (!pts_dev || !order_dev || !leaf_starts_host || !nleaves_out || pdim < 1 || pdim > 3 || n < 0 || n > 2147483647ll || leaf_size < 1
        || max_leaves < 1)
        FD_FAIL("fd_kd_order: bad arguments");
    *nleaves_out = 0;
    if (n == 0) { leaf_starts_host[0] = 0; return 0; }
    hipStream_t s = fd::st(s_);
    int64_t L = (n + leaf_size - 1) / leaf_size;
    if (L > max_leaves) L = max_leaves;
    struct Seg { int64_t start, count, leaves; };
    std::vector<Seg> segs{{0, n, L}};
    int32_t *idx = nullptr, *seg = nullptr;
    uint64_t *keys = nullptr;
    double *box = nullptr;
    KdSeg *tab = nullptr;
    FD_HIP(hipMalloc(&idx, (size_t)n * 4)); FD_HIP(hipMalloc(&seg, (size_t)n * 4)); FD_HIP(hipMalloc(&keys, (size_t)n * 8));
    FD_HIP(hipMalloc(&box, (size_t)L * 6 * 8)); FD_HIP(hipMalloc(&tab, (size_t)L * sizeof(KdSeg)));
    hipLaunchKernelGGL(kd_iota, dim3(lo_grid(n)), dim3(256), 0, s, idx, seg, n);
    int rc = 0;
    std::vector<KdSeg> htab;
    std::vector<double> hbox;
    while (!rc) {
        bool any = false;
        for (const Seg &g : segs) if (g.leaves > 1) { any = true; break; }
        if (!any) break;
        const size_t ns = segs.size();
        std::vector<Seg> next;
        next.reserve(2 * ns);
        htab.assign(ns, KdSeg{0, 0, 0, 0});
        for (size_t k = 0; k < ns; ++k) {
            const Seg &g = segs[k];
            htab[k].start = (int32_t)g.start;
            htab[k].child = (int32_t)next.size();
            if (g.leaves > 1) {
                const int64_t ll = g.leaves / 2;
                const int64_t cl = (g.count * ll + g.leaves / 2) / g.leaves;        // the left child's share, rounded
                htab[k].cleft = (int32_t)cl; htab[k].split = 1;
                next.push_back({g.start, cl, ll});
                next.push_back({g.start + cl, g.count - cl, g.leaves - ll});
            } else {
                next.push_back(g);
            }
        }
        hbox.resize(ns * 6);
        for (size_t k = 0; k < ns; ++k) for (int c = 0; c < 3; ++c) { hbox[k * 6 + c] = DBL_MAX; hbox[k * 6 + 3 + c] = -DBL_MAX; }
        FD_HIP(hipMemcpyAsync(box, hbox.data(), ns * 6 * 8, hipMemcpyHostToDevice, s));
        FD_HIP(hipMemcpyAsync(tab, htab.data(), ns * sizeof(KdSeg), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kd_bbox, dim3(lo_grid(n)), dim3(256), 0, s, pts_dev, pdim, idx, seg, n, box);
        hipLaunchKernelGGL(kd_keys, dim3(lo_grid(n)), dim3(256), 0, s, pts_dev, pdim, idx, seg, n, box, tab, keys);
        FD_CHECK_LAUNCH();
        int sbits = 1;
        while ((1ull << sbits) < ns) ++sbits;
        rc = sort_pairs<uint64_t, int32_t>(keys, idx, n, 32 + sbits, s);           // (synchronises: htab / hbox may be reused)
        if (rc) break;
        hipLaunchKernelGGL(kd_relabel, dim3(lo_grid(n)), dim3(256), 0, s, keys, n, tab, seg);
        FD_CHECK_LAUNCH();
        FD_HIP(hipStreamSynchronize(s));
        segs.swap(next);
    }
    if (!rc && segs.size() > 1) {
        // inside a leaf the points go in index order: whatever structure the caller's numbering has (lattice lines of a
        // structured mesh, first-touch runs of a DMPlex one) survives, so neighbouring rows of a block stay neighbours in
        // its accumulator and the instance scheduler finds conflict-free windows as it does for producer tiles
        hipLaunchKernelGGL(kd_leaf_keys, dim3(lo_grid(n)), dim3(256), 0, s, idx, seg, n, keys);
        int sbits = 1;
        while ((1ull << sbits) < segs.size()) ++sbits;
        rc = sort_pairs<uint64_t, int32_t>(keys, idx, n, 32 + sbits, s);
    }
    if (!rc) {
        hipLaunchKernelGGL(kd_finish, dim3(lo_grid(n)), dim3(256), 0, s, idx, base, n, order_dev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { fd::set_error(hipGetErrorString(e)); rc = (int)e; }
        else FD_HIP(hipStreamSynchronize(s));
        for (size_t k = 0; k < segs.size(); ++k) leaf_starts_host[k] = (int32_t)segs[k].start;
        leaf_starts_host[segs.size()] = (int32_t)n;
        *nleaves_out = (int32_t)segs.size();
    }
    (void)hipFree(idx); (void)hipFree(seg); (void)hipFree(keys); (void)hipFree(box); (void)hipFree(tab);
    return rc;
}

int fd_first_touch_order(const int32_t *map_dev, int arity, const int32_t *order_dev, int64_t n, int32_t nnodes,
                         int32_t *pinv_dev, int32_t *plist_dev, int32_t *rank_dev, fd_stream_t s_) {
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
    if (!rc && rank_dev) {
        hipLaunchKernelGGL(ft_ranks, dim3(lo_grid(nnodes)), dim3(256), 0, s, keys, nnodes, rank_dev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { fd::set_error(hipGetErrorString(e)); rc = (int)e; }
    }
    if (!rc) {
        hipLaunchKernelGGL(ft_invert, dim3(lo_grid(nnodes)), dim3(256), 0, s, plist_dev, nnodes, pinv_dev);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { fd::set_error(hipGetErrorString(e)); rc = (int)e; }
        else FD_HIP(hipStreamSynchronize(s));
    }
    (void)hipFree(nk); (void)hipFree(keys);
    return rc;
}

int fd_row_entry_positions(int32_t npos, const fd_nnz_t *prowptr_dev, const fd_nnz_t *gstart_dev, int32_t *gpos_dev, fd_stream_t s) {
    if (npos <= 0) return 0;
    hipLaunchKernelGGL(row_entry_positions, dim3(lo_grid((int64_t)npos * 16)), dim3(256), 0, fd::st(s), npos, prowptr_dev, gstart_dev, gpos_dev);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_row_entry_positions_masked(int32_t npos, const fd_nnz_t *prowptr_dev, const fd_nnz_t *gstart_dev, const int32_t *colidx_dev,
                                  const int32_t *col_lgmap_dev, int32_t *gpos_dev, fd_stream_t s) {
    if (npos <= 0) return 0;
    if (!prowptr_dev || !gstart_dev || !colidx_dev || !col_lgmap_dev || !gpos_dev) FD_FAIL("fd_row_entry_positions_masked: bad arguments");
    hipLaunchKernelGGL(row_entry_positions_masked, dim3(lo_grid((int64_t)npos * 16)), dim3(256), 0, fd::st(s), npos, prowptr_dev, gstart_dev,
                       colidx_dev, col_lgmap_dev, gpos_dev);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_ocr_row_runs(int32_t npos, const fd_nnz_t *prowptr_dev, const fd_nnz_t *gstart_dev, const int32_t *rblk_dev, int32_t nblocks,
                    uint8_t *grun_dev, int32_t *brun_dev, fd_nnz_t *rdelta_dev, int32_t *nruns_out, int32_t *max_runs_out, fd_stream_t s_) {
    if (!prowptr_dev || !gstart_dev || !rblk_dev || !grun_dev || !brun_dev || !rdelta_dev || !nruns_out || !max_runs_out || npos < 0 || nblocks < 0)
        FD_FAIL("fd_ocr_row_runs: bad arguments");
    *nruns_out = 0; *max_runs_out = 0;
    if (npos == 0 || nblocks == 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *flag = nullptr, *runidx = nullptr, *err = nullptr; void *tmp = nullptr;
    FD_HIP(hipMalloc(&flag, (size_t)npos * 4));
    FD_HIP(hipMalloc(&runidx, (size_t)npos * 4));
    FD_HIP(hipMalloc(&err, 4));
    FD_HIP(hipMemsetAsync(err, 0, 4, s));
    hipLaunchKernelGGL(rr_flags, dim3(lo_grid(npos)), dim3(256), 0, s, npos, prowptr_dev, gstart_dev, rblk_dev, nblocks, flag);
    hipLaunchKernelGGL(rr_block_starts, dim3(lo_grid(nblocks)), dim3(256), 0, s, rblk_dev, nblocks, npos, flag);
    size_t tb = 0;
    FD_HIP(hipcub::DeviceScan::InclusiveSum(nullptr, tb, flag, runidx, npos, s));
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceScan::InclusiveSum(tmp, tb, flag, runidx, npos, s));
    hipLaunchKernelGGL(rr_tables, dim3(lo_grid(npos)), dim3(256), 0, s, npos, prowptr_dev, gstart_dev, flag, runidx, rblk_dev, nblocks,
                       brun_dev, rdelta_dev);
    hipLaunchKernelGGL(rr_entries, dim3(nblocks < 65536 ? nblocks : 65536), dim3(256), 0, s, rblk_dev, nblocks, npos, prowptr_dev, runidx,
                       brun_dev, grun_dev, err);
    FD_CHECK_LAUNCH();
    std::vector<int32_t> br((size_t)nblocks + 1);
    int32_t e = 0;
    FD_HIP(hipMemcpyAsync(br.data(), brun_dev, ((size_t)nblocks + 1) * 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(nruns_out, runidx + (npos - 1), 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    int32_t mx = 0;
    for (int32_t b = 0; b < nblocks; ++b) mx = std::max(mx, br[b + 1] - br[b]);
    // (the last block's runs end at nruns: brun[nblocks] is the run of position npos, i.e. nruns when rblk[nblocks] == npos)
    *max_runs_out = mx;
    (void)hipFree(flag); (void)hipFree(runidx); (void)hipFree(err); (void)hipFree(tmp);
    return 0;
}

int fd_row_order_tables(int32_t npos, const int32_t *plist_dev, const fd_nnz_t *rowptr_dev, fd_nnz_t *prowptr_dev, fd_nnz_t *nstart_dev,
                        fd_nnz_t *gstart_dev, fd_stream_t s_) {
    if (npos < 0 || !prowptr_dev || (npos && (!plist_dev || !rowptr_dev || !nstart_dev || !gstart_dev))) FD_FAIL("fd_row_order_tables: bad arguments");
    hipStream_t s = fd::st(s_);
    if (npos == 0) { FD_HIP(hipMemsetAsync(prowptr_dev, 0, sizeof(fd_nnz_t), s)); return 0; }
    fd_nnz_t *len = nullptr; void *tmp = nullptr;
    FD_HIP(hipMalloc(&len, ((size_t)npos + 1) * sizeof(fd_nnz_t)));
    hipLaunchKernelGGL(ro_gather, dim3(lo_grid((int64_t)npos + 1)), dim3(256), 0, s, npos, plist_dev, rowptr_dev, len, gstart_dev);
    FD_CHECK_LAUNCH();
    size_t tb = 0;
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, len, prowptr_dev, npos + 1, s));
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, len, prowptr_dev, npos + 1, s));
    hipLaunchKernelGGL(ro_nstart, dim3(lo_grid(npos)), dim3(256), 0, s, npos, plist_dev, prowptr_dev, nstart_dev);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    (void)hipFree(len); (void)hipFree(tmp);
    return 0;
}

int fd_invert_permutation(const int32_t *plist_dev, int32_t n, int32_t *pinv_dev, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(ft_invert, dim3(lo_grid(n)), dim3(256), 0, fd::st(s), plist_dev, n, pinv_dev);
    FD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
