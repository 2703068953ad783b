// fd_csr.hip -- device-side sparsity construction and CSR services (include/fdhip.h: fd_csr_*).
//
// Native replacement of pyop2/sparsity.pyx:105-159 (build_sparsity) and :162-389
// (fill_with_zeros), which walk the maps calling MatSetValuesBlockedLocal on a PETSc
// MATPREALLOCATOR.  Here: emit one 64-bit key (row<<32 | col) per candidate entry,
// radix-sort, unique, and derive rowptr/colidx -- all on the GPU.  One-off per
// function-space pair, excluded from the DoFs/s metric.
#include "fd_common.h"
#include <hipcub/hipcub.hpp>
#include <algorithm>
#include <cstdlib>

namespace {

// One candidate key per (entity, layer, stacked cell x row entry, stacked cell x col entry).  nf = 2 for the
// ON_INTERIOR_FACETS region (two vertically stacked cells per facet, sparsity.pyx:298-303); layers l0 .. l0+nli-1 of
// the nl cell layers are visited (sparsity.pyx:331-346).  Node of entry i in layer l (sparsity.pyx:357-368):
//   map[e][i % arity] + off[i % arity] * ((l + i / arity + quot) % nl - quot % nl)      (quot = 0 unless periodic)
// Variable layers (`layers` = per-entity [bottom, top) rows, set.py:326-337): every entity gets nli = the longest
// column's slots; slots outside its own layer range emit the dropped-key sentinel.
__global__ void emit_keys(const int32_t *__restrict__ rmap, const int32_t *__restrict__ cmap, int32_t nent,
                          int ar, int ac, int nl, int l0, int nli, int nf, const int32_t *__restrict__ roff,
                          const int32_t *__restrict__ coff, const int32_t *__restrict__ rquot,
                          const int32_t *__restrict__ cquot, const int32_t *__restrict__ layers, int region,
                          int32_t nrows, int32_t ncols, uint64_t *__restrict__ keys) {
    const int nr = nf * ar, nc = nf * ac;
    const int64_t per = (int64_t)nr * nc;
    const int64_t L = nl > 0 ? nli : 1;
    const int64_t total = (int64_t)nent * L * per;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * blockDim.x) {
        int64_t e = t / (L * per);
        int64_t rem = t - e * L * per;
        int l = l0 + (int)(rem / per);
        int ij = (int)(rem % per);
        int i = ij / nc, j = ij - i * nc;
        const int ki = i / ar, ii = i - ki * ar, kj = j / ac, jj = j - kj * ac;
        int r = rmap[e * ar + ii];
        int c = cmap[e * ac + jj];
        bool live = true;
        if (nl > 0) {
            int nle = nl;
            if (layers) {       // this column's cell layers and the part of them the region visits (sparsity.pyx:331-346)
                nle = layers[2 * e + 1] - 1 - layers[2 * e];
                const int lo = region == FD_ON_TOP ? nle - 1 : 0;
                const int hi = region == FD_ON_BOTTOM ? 1 : (region == FD_ON_INTERIOR_FACETS ? nle - 1 : nle);
                live = l >= lo && l < hi;
            }
            const int qr = rquot ? rquot[ii] : 0, qc = cquot ? cquot[jj] : 0;
            if (live && r >= 0) r += roff[ii] * ((l + ki + qr) % nle - qr % nle);
            if (live && c >= 0) c += coff[jj] * ((l + kj + qc) % nle - qc % nle);
        }
        uint64_t key = ~0ull;                       // sentinel: dropped (negative / out of range / no such layer)
        if (live && r >= 0 && r < nrows && c >= 0 && c < ncols) key = ((uint64_t)(uint32_t)r << 32) | (uint32_t)c;
        keys[t] = key;
    }
}

__global__ void emit_diag(int32_t first, int32_t n, uint64_t *__restrict__ keys) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x)
        keys[t] = ((uint64_t)(uint32_t)(first + t) << 32) | (uint32_t)(first + t);
}

__global__ void keys_to_csr(const uint64_t *__restrict__ keys, int64_t nnz, int32_t nrows,
                            fd_nnz_t *__restrict__ rowptr, int32_t *__restrict__ colidx) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t <= nnz; t += (int64_t)gridDim.x * blockDim.x) {
        int32_t r = t < nnz ? (int32_t)(keys[t] >> 32) : nrows;
        int32_t rp = t > 0 ? (int32_t)(keys[t - 1] >> 32) : -1;
        for (int32_t rr = rp + 1; rr <= r; ++rr) rowptr[rr] = (fd_nnz_t)t;
        if (t < nnz) colidx[t] = (int32_t)(keys[t] & 0xffffffffu);
    }
}

__global__ void expand_rowptr(int32_t nnode, const fd_nnz_t *__restrict__ nrp, int rbs, int cbs,
                              fd_nnz_t *__restrict__ rp) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t <= (int64_t)nnode * rbs;
         t += (int64_t)gridDim.x * blockDim.x) {
        if (t == (int64_t)nnode * rbs) { rp[t] = nrp[nnode] * rbs * cbs; continue; }
        int32_t r = (int32_t)(t / rbs), p = (int32_t)(t - (int64_t)r * rbs);
        const fd_nnz_t len = nrp[r + 1] - nrp[r];
        rp[t] = nrp[r] * rbs * cbs + p * len * cbs;
    }
}

__global__ void expand_colidx(int32_t nnode, const fd_nnz_t *__restrict__ nrp, const int32_t *__restrict__ nci,
                              int rbs, int cbs, const fd_nnz_t *__restrict__ rp, int32_t *__restrict__ ci) {
    // one thread per (node row, p): writes its row's len*cbs entries
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < (int64_t)nnode * rbs;
         t += (int64_t)gridDim.x * blockDim.x) {
        int32_t r = (int32_t)(t / rbs);
        fd_nnz_t o = rp[t];
        for (fd_nnz_t q = nrp[r]; q < nrp[r + 1]; ++q)
            for (int c = 0; c < cbs; ++c) ci[o++] = nci[q] * cbs + c;
    }
}

__device__ inline fd_nnz_t csr_find(const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int r, int c) {
    fd_nnz_t lo = rowptr[r], hi = rowptr[r + 1] - 1;
    while (lo <= hi) {
        const fd_nnz_t mid = lo + ((hi - lo) >> 1);
        int v = colidx[mid];
        if (v == c) return mid;
        if (v < c) lo = mid + 1; else hi = mid - 1;
    }
    return -1;
}

__global__ void elem_offsets(const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                             const int32_t *__restrict__ rmap, const int32_t *__restrict__ cmap, int32_t nent,
                             int ar, int ac, int nl, const int32_t *__restrict__ roff, const int32_t *__restrict__ coff,
                             int32_t *__restrict__ out) {
    const int64_t per = (int64_t)ar * ac, L = nl > 0 ? nl : 1, total = (int64_t)nent * L * per;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        int64_t el = t / per;
        int64_t e = el / L;
        int l = (int)(el - e * L);
        int ij = (int)(t - el * per);
        int i = ij / ac, j = ij - i * ac;
        int r = rmap[e * ar + i], c = cmap[e * ac + j];
        if (nl > 0) { if (r >= 0) r += roff[i] * l; if (c >= 0) c += coff[j] * l; }
        out[t] = (r >= 0 && c >= 0) ? (int32_t)csr_find(rowptr, colidx, r, c) : -1;      // (callers check nnz < 2^31)
    }
}

__global__ void set_diag(const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, double *__restrict__ vals,
                         const int32_t *__restrict__ rows, int32_t n, double v, int zero_row) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        int r = rows[t];
        if (r < 0) continue;
        if (zero_row) {
            for (fd_nnz_t q = rowptr[r]; q < rowptr[r + 1]; ++q) vals[q] = (colidx[q] == r) ? v : 0.0;
        } else {
            const fd_nnz_t q = csr_find(rowptr, colidx, r, r);
            if (q >= 0) vals[q] = v;
        }
    }
}

// places of the diagonal entries of the selected rows (-1: row not selected / no diagonal entry), found once; set_at then is one
// coalesced read of the places and one scattered store per row instead of rows -> rowptr -> log2(row length) dependent colidx loads
__global__ void diag_positions(const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, const int32_t *__restrict__ rows,
                               int32_t n, fd_nnz_t *__restrict__ pos) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const int r = rows[t];
        pos[t] = r < 0 ? (fd_nnz_t)-1 : csr_find(rowptr, colidx, r, r);
    }
}
__global__ void set_at(double *__restrict__ vals, const fd_nnz_t *__restrict__ pos, int32_t n, double v) {
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
        const fd_nnz_t q = pos[t];
        if (q >= 0) vals[q] = v;
    }
}

// ---- MPIAIJ split (pyop2/types/mat.py:254-278: d_nnz / o_nnz; MatCreateMPIAIJWithSplitArrays): columns are sorted
// inside a row and the owned columns [0, ncols_owned) come first, so the diagonal block is a prefix of every row
__global__ void split_counts(int32_t nrows, const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx, int32_t ncols_owned,
                             int32_t *__restrict__ dcnt, int32_t *__restrict__ ocnt) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        fd_nnz_t lo = rowptr[r], hi = rowptr[r + 1];
        const fd_nnz_t b = lo, e = hi;
        while (lo < hi) { const fd_nnz_t mid = lo + ((hi - lo) >> 1); if (colidx[mid] < ncols_owned) lo = mid + 1; else hi = mid; }
        dcnt[r] = (int32_t)(lo - b);
        ocnt[r] = (int32_t)(e - lo);
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) { dcnt[nrows] = 0; ocnt[nrows] = 0; }
}

// one wavefront per row: copy the prefix into the diagonal block (local column indices) and the suffix into the
// off-diagonal block (GLOBAL column indices through col_global, as MatCreateMPIAIJWithSplitArrays expects)
// The off-diagonal block is a SeqAIJ matrix of its own (MatCreateSeqAIJWithArrays: sorted column indices per row), and the
// local ghost numbering is not monotone in the global one: WITH_IDX ranks every off-diagonal entry inside its row by global
// column (rows are short: a quadratic count) and stores the rank; WITH_VALS scatters the values through it.
template <bool WITH_IDX, bool WITH_VALS>
__global__ void split_fill(int32_t nrows, const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                           const double *__restrict__ vals, const int32_t *__restrict__ col_global,
                           const int32_t *__restrict__ drp, const int32_t *__restrict__ orp,
                           int32_t *__restrict__ dci, int32_t *__restrict__ oci, double *__restrict__ dv, double *__restrict__ ov,
                           int32_t *__restrict__ orank) {
    const int lane = threadIdx.x & 63;
    for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; r < nrows; r += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        const fd_nnz_t b = rowptr[r];
        const int nd = drp[r + 1] - drp[r], no = orp[r + 1] - orp[r];
        for (int q = lane; q < nd; q += 64) {
            if (WITH_IDX) dci[drp[r] + q] = colidx[b + q];
            if (WITH_VALS) dv[drp[r] + q] = vals[b + q];
        }
        for (int q = lane; q < no; q += 64) {
            if (WITH_IDX) {
                const int32_t c = colidx[b + nd + q], g = col_global ? col_global[c] : c;
                int rank = 0;
                for (int p = 0; p < no; ++p) {
                    const int32_t c2 = colidx[b + nd + p], g2 = col_global ? col_global[c2] : c2;
                    rank += (g2 < g || (g2 == g && p < q)) ? 1 : 0;
                }
                oci[orp[r] + rank] = g;
                orank[orp[r] + q] = rank;
            }
            if (WITH_VALS) ov[orp[r] + (orank ? orank[orp[r] + q] : q)] = vals[b + nd + q];
        }
    }
}

__global__ void get_diag(int32_t nrows, const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                         const double *__restrict__ vals, double *__restrict__ diag) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nrows; r += (int64_t)gridDim.x * blockDim.x) {
        fd_nnz_t lo = rowptr[r], hi = rowptr[r + 1];     // columns are sorted: binary search for the diagonal
        const fd_nnz_t end = hi;
        while (lo < hi) { const fd_nnz_t mid = lo + ((hi - lo) >> 1); if (colidx[mid] < (int32_t)r) lo = mid + 1; else hi = mid; }
        diag[r] = (lo < end && colidx[lo] == (int32_t)r) ? vals[lo] : 0.0;
    }
}

__global__ void spmv(int32_t nrows, const fd_nnz_t *__restrict__ rowptr, const int32_t *__restrict__ colidx,
                     const double *__restrict__ vals, const double *__restrict__ x, double *__restrict__ y) {
    // one wavefront per row
    const int lane = threadIdx.x & 63;
    for (int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) >> 6; r < nrows;
         r += ((int64_t)gridDim.x * blockDim.x) >> 6) {
        double acc = 0.0;
        for (fd_nnz_t q = rowptr[r] + lane; q < rowptr[r + 1]; q += 64) acc += vals[q] * x[colidx[q]];
        for (int d = 32; d > 0; d >>= 1) acc += __shfl_down(acc, d, 64);
        if (lane == 0) y[r] = acc;
    }
}

inline int grid_for(int64_t n, int block = 256) {
    int64_t g = (n + block - 1) / block;
    if (g < 1) g = 1;
    if (g > 256 * 32) g = 256 * 32;
    return (int)g;
}

}  // namespace

extern "C" {

// layers visited and stacked cells of one (pair, region): sparsity.pyx:291-305, 331-346
static void pair_layers(int nl, int region, int periodic, int *l0, int *nli, int *nf) {
    *l0 = 0; *nli = nl > 0 ? nl : 1; *nf = 1;
    if (nl <= 0) return;
    if (region == FD_ON_BOTTOM) *nli = 1;
    else if (region == FD_ON_TOP) { *l0 = nl - 1; *nli = 1; }
    else if (region == FD_ON_INTERIOR_FACETS) { *nf = 2; *nli = periodic ? nl : nl - 1; }
}

int fd_csr_from_maps_ex(int32_t nrows, int32_t ncols, int set_diag, int npairs,
                        const int32_t *const *rmaps, const int32_t *const *cmaps, const int32_t *nent,
                        const int32_t *rarity, const int32_t *carity, const int32_t *nlayers,
                        const int32_t *const *roffs_h, const int32_t *const *coffs_h,
                        const int32_t *region, const int32_t *periodic,
                        const int32_t *const *rquots_h, const int32_t *const *cquots_h,
                        const int32_t *const *layers_dev,
                        fd_nnz_t **rowptr_out, int32_t **colidx_out, int64_t *nnz_out, fd_stream_t s_) {
    hipStream_t s = fd::st(s_);
    int64_t ncand = 0;
    for (int k = 0; k < npairs; ++k) {
        int nl = (nlayers && nlayers[k] > 0) ? nlayers[k] : 0, l0, nli, nf;
        if (nl && region && region[k] != FD_ALL && region[k] != FD_ON_BOTTOM && region[k] != FD_ON_TOP &&
            region[k] != FD_ON_INTERIOR_FACETS) FD_FAIL("fd_csr_from_maps_ex: unknown iteration region");
        pair_layers(nl, region ? region[k] : FD_ALL, periodic ? periodic[k] : 0, &l0, &nli, &nf);
        if (layers_dev && layers_dev[k]) {          // variable layers: nlayers[k] = the longest column, all slots enumerated
            if (periodic && periodic[k]) FD_FAIL("fd_csr_from_maps_ex: periodic extrusion needs constant layers");
            l0 = 0; nli = nl;
        }
        if (nli < 0) nli = 0;
        ncand += (int64_t)nent[k] * nli * nf * rarity[k] * nf * carity[k];
    }
    int32_t ndiag = set_diag ? (nrows < ncols ? nrows : ncols) : 0;
    ncand += ndiag;
    // Candidate (row, column) keys are emitted, sorted and made unique in CHUNKS of at most CHUNK keys, the unique keys of
    // every chunk appended to an accumulator that is itself compacted (sort + unique) whenever it passes 2^30 keys -- or twice
    // what its last compaction left, for patterns beyond 2^30 nonzeros (the 215^3 CG2 cube of BASELINE configs[4] has 6.0e9
    // candidates for 2.3e9 nonzeros; the item counts of the primitives are 64-bit) -- so the peak footprint is a few times the
    // pattern instead of 16 bytes per candidate.
    // The chunk buffers are 2 x 8 B per key and most of this function's time at C2 size was spent allocating and releasing them
    // (954 M candidates in one chunk: 15 GB for 40 ms of kernels): chunks of 2^27 keys by default (2 GB), the accumulator compacted
    // as before when it passes 2^30.
    int64_t CHUNK = 1ll << 27, ACC_LIMIT = 1ll << 30;
    if (const char *e = getenv("FDHIP_CSR_CHUNK")) {          // (tests: force the many-chunk path and the accumulator's own compaction)
        const long long v = atoll(e);
        if (v > 0 && v < ACC_LIMIT) { CHUNK = v; ACC_LIMIT = v; }
    }
    auto sort_unique = [&](uint64_t *in, uint64_t *alt, int64_t n, void **tmp, size_t *tmp_cap, int64_t *nsel_dev, uint64_t **result,
                           int64_t *nout) -> int {
        // sorts `in` (n keys) using `alt` as the second buffer and leaves the unique keys in *result (one of the two)
        if (n == 0) { *result = in; *nout = 0; return 0; }
        hipcub::DoubleBuffer<uint64_t> db(in, alt);
        size_t tb = 0;
        FD_HIP(hipcub::DeviceRadixSort::SortKeys(nullptr, tb, db, (int64_t)n, 0, 64, s));   // (~0 sentinel sorts last: all bits)
        if (tb > *tmp_cap) { if (*tmp) FD_HIP(hipFree(*tmp)); FD_HIP(hipMalloc(tmp, tb)); *tmp_cap = tb; }
        FD_HIP(hipcub::DeviceRadixSort::SortKeys(*tmp, tb, db, (int64_t)n, 0, 64, s));
        uint64_t *sorted = db.Current(), *other = (sorted == in) ? alt : in;
        size_t tb2 = 0;
        FD_HIP(hipcub::DeviceSelect::Unique(nullptr, tb2, sorted, other, nsel_dev, (int64_t)n, s));
        if (tb2 > *tmp_cap) { if (*tmp) FD_HIP(hipFree(*tmp)); FD_HIP(hipMalloc(tmp, tb2)); *tmp_cap = tb2; }
        FD_HIP(hipcub::DeviceSelect::Unique(*tmp, tb2, sorted, other, nsel_dev, (int64_t)n, s));
        FD_HIP(hipMemcpyAsync(nout, nsel_dev, 8, hipMemcpyDeviceToHost, s));
        FD_HIP(hipStreamSynchronize(s));
        *result = other;
        return 0;
    };
    const int64_t ccap = ncand < CHUNK ? ncand + 1 : CHUNK;
    uint64_t *ck = nullptr, *ck2 = nullptr;                   // chunk buffers
    uint64_t *acc = nullptr, *acc2 = nullptr;                 // accumulator (+ its sort partner); grown on demand
    int64_t acc_n = 0, acc_cap = 0, next_compact = ACC_LIMIT;
    bool reserved = false;
    void *tmp = nullptr;
    size_t tmp_cap = 0;
    int64_t *nsel = nullptr;
    FD_HIP(hipMalloc(&ck, (size_t)ccap * 8));
    FD_HIP(hipMalloc(&ck2, (size_t)ccap * 8));
    FD_HIP(hipMalloc(&nsel, 8));
    auto compact = [&]() -> int {                             // accumulator := its unique keys
        uint64_t *res = nullptr;
        int64_t n = 0;
        if (int rc = sort_unique(acc, acc2, acc_n, &tmp, &tmp_cap, nsel, &res, &n)) return rc;
        if (res != acc) std::swap(acc, acc2);
        acc_n = n;
        next_compact = std::max<int64_t>(ACC_LIMIT, 2 * n);
        return 0;
    };
    auto append = [&](const uint64_t *src, int64_t n) -> int {
        if (acc_n + n > acc_cap) {
            if (acc_n > next_compact) { if (int rc = compact()) return rc; }
            if (acc_n + n > acc_cap) {
                int64_t want = acc_cap ? acc_cap * 2 : (n + 1);
                while (want < acc_n + n) want *= 2;
                if (sizeof(fd_nnz_t) == 4 && want > 2147483647ll) want = 2147483647ll;
                if (acc_n + n > want) { fd::set_error("fd_csr_from_maps: more than 2^31-1 distinct candidate entries (nnz exceeds IntType)"); return -1; }
                uint64_t *na = nullptr, *na2 = nullptr;
                FD_HIP(hipMalloc(&na, (size_t)want * 8));
                FD_HIP(hipMalloc(&na2, (size_t)want * 8));
                if (acc_n) FD_HIP(hipMemcpyAsync(na, acc, (size_t)acc_n * 8, hipMemcpyDeviceToDevice, s));
                FD_HIP(hipStreamSynchronize(s));
                if (acc) FD_HIP(hipFree(acc));
                if (acc2) FD_HIP(hipFree(acc2));
                acc = na; acc2 = na2; acc_cap = want;
            }
        }
        if (n) FD_HIP(hipMemcpyAsync(acc + acc_n, src, (size_t)n * 8, hipMemcpyDeviceToDevice, s));
        acc_n += n;
        return 0;
    };
    if (ndiag) {
        for (int64_t d0 = 0; d0 < ndiag; d0 += ccap) {        // (the diagonal keys are distinct: appended as they are)
            const int64_t n = std::min<int64_t>(ccap, ndiag - d0);
            hipLaunchKernelGGL(emit_diag, dim3(grid_for(n)), dim3(256), 0, s, (int32_t)d0, (int32_t)n, ck);
            FD_CHECK_LAUNCH();
            if (int rc = append(ck, n)) return rc;
            FD_HIP(hipStreamSynchronize(s));
        }
    }
    for (int k = 0; k < npairs; ++k) {
        int nl = (nlayers && nlayers[k] > 0) ? nlayers[k] : 0, l0, nli, nf;
        pair_layers(nl, region ? region[k] : FD_ALL, periodic ? periodic[k] : 0, &l0, &nli, &nf);
        const int32_t *lay = (layers_dev && layers_dev[k]) ? layers_dev[k] : nullptr;
        if (lay) { l0 = 0; nli = nl; }
        if (nli < 0) nli = 0;
        const int64_t per_ent = (int64_t)nli * nf * rarity[k] * nf * carity[k];
        if (per_ent == 0 || nent[k] == 0) continue;
        if (per_ent > ccap) FD_FAIL("fd_csr_from_maps_ex: one entity column emits more than 2^30 candidate entries");
        int32_t *roff = nullptr, *coff = nullptr, *rq = nullptr, *cq = nullptr;
        if (nl) {
            if (!roffs_h || !coffs_h || !roffs_h[k] || !coffs_h[k]) FD_FAIL("fd_csr_from_maps_ex: extruded pairs need the map offsets");
            FD_HIP(hipMalloc(&roff, rarity[k] * 4)); FD_HIP(hipMalloc(&coff, carity[k] * 4));
            FD_HIP(hipMemcpyAsync(roff, roffs_h[k], rarity[k] * 4, hipMemcpyHostToDevice, s));
            FD_HIP(hipMemcpyAsync(coff, coffs_h[k], carity[k] * 4, hipMemcpyHostToDevice, s));
            if (rquots_h && rquots_h[k]) {
                FD_HIP(hipMalloc(&rq, rarity[k] * 4));
                FD_HIP(hipMemcpyAsync(rq, rquots_h[k], rarity[k] * 4, hipMemcpyHostToDevice, s));
            }
            if (cquots_h && cquots_h[k]) {
                FD_HIP(hipMalloc(&cq, carity[k] * 4));
                FD_HIP(hipMemcpyAsync(cq, cquots_h[k], carity[k] * 4, hipMemcpyHostToDevice, s));
            }
        }
        const int64_t epc = std::max<int64_t>(1, ccap / per_ent);              // entities per chunk
        for (int64_t e0 = 0; e0 < nent[k]; e0 += epc) {
            const int64_t ne = std::min<int64_t>(epc, (int64_t)nent[k] - e0), cnt = ne * per_ent;
            hipLaunchKernelGGL(emit_keys, dim3(grid_for(cnt)), dim3(256), 0, s, rmaps[k] + e0 * rarity[k], cmaps[k] + e0 * carity[k],
                               (int32_t)ne, rarity[k], carity[k], nl, l0, nli, nf, roff, coff, rq, cq, lay ? lay + 2 * e0 : nullptr,
                               region ? region[k] : FD_ALL, nrows, ncols, ck);
            FD_CHECK_LAUNCH();
            uint64_t *res = nullptr;
            int64_t nu_c = 0;
            if (int rc = sort_unique(ck, ck2, cnt, &tmp, &tmp_cap, nsel, &res, &nu_c)) return rc;
            if (!reserved && cnt > 0) {
                // size the accumulator once from the first chunk's share of distinct keys instead of doubling it chunk after chunk
                reserved = true;
                const double ratio = (double)nu_c / (double)cnt;
                double est = (double)acc_n + (double)nu_c + ratio * 1.05 * (double)(ncand - ndiag - cnt) + 4096.0;
                // never beyond what one compaction cycle can hold: the first chunk's ratio overstates the global one (a random
                // numbering has next to no duplicates inside a chunk), and append() compacts once the accumulator passes ACC_LIMIT
                // -- which it must be allowed to reach before the capacity is
                const double cap_ = (double)ACC_LIMIT + (double)ccap;
                if (est > cap_) est = cap_;
                if (sizeof(fd_nnz_t) == 4 && est > 2147483647.0) est = 2147483647.0;
                const int64_t want = (int64_t)est;
                if (want > acc_cap) {
                    // a reservation that does not fit the device is not an error: append() grows the accumulator step by step
                    uint64_t *na = nullptr, *na2 = nullptr;
                    const bool ok = hipMalloc(&na, (size_t)want * 8) == hipSuccess && hipMalloc(&na2, (size_t)want * 8) == hipSuccess;
                    if (!ok) {
                        (void)hipGetLastError();
                        if (na) (void)hipFree(na);
                        if (na2) (void)hipFree(na2);
                    } else {
                        hipError_t ce = acc_n ? hipMemcpyAsync(na, acc, (size_t)acc_n * 8, hipMemcpyDeviceToDevice, s) : hipSuccess;
                        if (ce == hipSuccess) ce = hipStreamSynchronize(s);
                        if (ce != hipSuccess) { (void)hipFree(na); (void)hipFree(na2); FD_HIP(ce); }
                        if (acc) FD_HIP(hipFree(acc));
                        if (acc2) FD_HIP(hipFree(acc2));
                        acc = na; acc2 = na2; acc_cap = want;
                    }
                }
            }
            if (int rc = append(res, nu_c)) return rc;
            FD_HIP(hipStreamSynchronize(s));                   // (the chunk buffers are reused)
        }
        if (nl) {
            FD_HIP(hipStreamSynchronize(s)); FD_HIP(hipFree(roff)); FD_HIP(hipFree(coff));
            if (rq) FD_HIP(hipFree(rq));
            if (cq) FD_HIP(hipFree(cq));
        }
    }
    if (acc_n > 0) { if (int rc = compact()) return rc; }
    int64_t nu = acc_n;
    uint64_t *other = acc;
    // drop the sentinel (if any candidate was masked it is the last unique key)
    if (nu > 0) {
        uint64_t last;
        FD_HIP(hipMemcpy(&last, other + nu - 1, 8, hipMemcpyDeviceToHost));
        if (last == ~0ull) --nu;
    }
    if (sizeof(fd_nnz_t) == 4 && nu > 2147483647ll) FD_FAIL("fd_csr_from_maps: nnz exceeds int32 (IntType)");
    fd_nnz_t *rowptr = nullptr;
    int32_t *colidx = nullptr;
    FD_HIP(hipMalloc(&rowptr, ((size_t)nrows + 1) * sizeof(fd_nnz_t)));
    FD_HIP(hipMalloc(&colidx, (size_t)(nu > 0 ? nu : 1) * 4));
    hipLaunchKernelGGL(keys_to_csr, dim3(grid_for(nu + 1)), dim3(256), 0, s, other, nu, nrows, rowptr, colidx);
    FD_CHECK_LAUNCH();
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(ck)); FD_HIP(hipFree(ck2)); FD_HIP(hipFree(nsel));
    if (acc) FD_HIP(hipFree(acc));
    if (acc2) FD_HIP(hipFree(acc2));
    if (tmp) FD_HIP(hipFree(tmp));
    *rowptr_out = rowptr; *colidx_out = colidx; *nnz_out = nu;
    return 0;
}

int fd_csr_from_maps(int32_t nrows, int32_t ncols, int set_diag, int npairs,
                     const int32_t *const *rmaps, const int32_t *const *cmaps, const int32_t *nent,
                     const int32_t *rarity, const int32_t *carity, const int32_t *nlayers,
                     const int32_t *const *roffs_h, const int32_t *const *coffs_h,
                     fd_nnz_t **rowptr_out, int32_t **colidx_out, int64_t *nnz_out, fd_stream_t s) {
    return fd_csr_from_maps_ex(nrows, ncols, set_diag, npairs, rmaps, cmaps, nent, rarity, carity, nlayers, roffs_h,
                               coffs_h, nullptr, nullptr, nullptr, nullptr, nullptr, rowptr_out, colidx_out, nnz_out, s);
}

int fd_csr_expand_blocks(int32_t nnode, const fd_nnz_t *nrp, const int32_t *nci, int rbs, int cbs,
                         fd_nnz_t **rp_out, int32_t **ci_out, fd_stream_t s_) {
    hipStream_t s = fd::st(s_);
    fd_nnz_t nnz_node;
    FD_HIP(hipMemcpy(&nnz_node, nrp + nnode, sizeof(fd_nnz_t), hipMemcpyDeviceToHost));
    int64_t nnz = (int64_t)nnz_node * rbs * cbs;
    if (sizeof(fd_nnz_t) == 4 && nnz > 2147483647ll) FD_FAIL("fd_csr_expand_blocks: nnz exceeds int32");
    fd_nnz_t *rp = nullptr;
    int32_t *ci = nullptr;
    FD_HIP(hipMalloc(&rp, ((size_t)nnode * rbs + 1) * sizeof(fd_nnz_t)));
    FD_HIP(hipMalloc(&ci, (size_t)(nnz > 0 ? nnz : 1) * 4));
    hipLaunchKernelGGL(expand_rowptr, dim3(grid_for((int64_t)nnode * rbs + 1)), dim3(256), 0, s, nnode, nrp, rbs, cbs, rp);
    FD_CHECK_LAUNCH();
    hipLaunchKernelGGL(expand_colidx, dim3(grid_for((int64_t)nnode * rbs)), dim3(256), 0, s, nnode, nrp, nci, rbs, cbs, rp, ci);
    FD_CHECK_LAUNCH();
    *rp_out = rp; *ci_out = ci;
    return 0;
}

int fd_csr_elem_offsets(const fd_nnz_t *rowptr, const int32_t *colidx, const int32_t *rmap, const int32_t *cmap,
                        int32_t nent, int ar, int ac, int nlayers, const int32_t *roff_host, const int32_t *coff_host,
                        int32_t *out, fd_stream_t s_) {
    if (nent <= 0) return 0;
    hipStream_t s = fd::st(s_);
    int32_t *roff = nullptr, *coff = nullptr;
    if (nlayers > 0) {
        if (!roff_host || !coff_host) FD_FAIL("fd_csr_elem_offsets: extruded tables need the map offsets");
        FD_HIP(hipMalloc(&roff, ar * 4)); FD_HIP(hipMalloc(&coff, ac * 4));
        FD_HIP(hipMemcpyAsync(roff, roff_host, ar * 4, hipMemcpyHostToDevice, s));
        FD_HIP(hipMemcpyAsync(coff, coff_host, ac * 4, hipMemcpyHostToDevice, s));
    }
    int64_t total = (int64_t)nent * (nlayers > 0 ? nlayers : 1) * ar * ac;
    hipLaunchKernelGGL(elem_offsets, dim3(grid_for(total)), dim3(256), 0, s, rowptr, colidx,
                       rmap, cmap, nent, ar, ac, nlayers, roff, coff, out);
    FD_CHECK_LAUNCH();
    if (nlayers > 0) { FD_HIP(hipStreamSynchronize(s)); FD_HIP(hipFree(roff)); FD_HIP(hipFree(coff)); }
    return 0;
}

int fd_csr_set_diagonal(const fd_nnz_t *rowptr, const int32_t *colidx, double *vals, const int32_t *rows, int32_t n,
                        double v, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(set_diag, dim3(grid_for(n)), dim3(256), 0, fd::st(s), rowptr, colidx, vals, rows, n, v, 0);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_diag_positions(const fd_nnz_t *rowptr, const int32_t *colidx, const int32_t *rows, int32_t n, fd_nnz_t *pos, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(diag_positions, dim3(grid_for(n)), dim3(256), 0, fd::st(s), rowptr, colidx, rows, n, pos);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_set_at(double *vals, const fd_nnz_t *pos, int32_t n, double v, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(set_at, dim3(grid_for(n)), dim3(256), 0, fd::st(s), vals, pos, n, v);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_zero_rows(const fd_nnz_t *rowptr, const int32_t *colidx, double *vals, const int32_t *rows, int32_t n,
                     double v, fd_stream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(set_diag, dim3(grid_for(n)), dim3(256), 0, fd::st(s), rowptr, colidx, vals, rows, n, v, 1);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_split_mpiaij(int32_t nrows_owned, const fd_nnz_t *rowptr, const int32_t *colidx, int32_t ncols_owned,
                        const int32_t *col_global, int32_t **d_rowptr, int32_t **d_colidx, int64_t *d_nnz,
                        int32_t **o_rowptr, int32_t **o_colidx, int32_t **o_rank, int64_t *o_nnz, fd_stream_t s_) {
    if (nrows_owned < 0 || !rowptr || !colidx || !d_rowptr || !d_colidx || !o_rowptr || !o_colidx || !o_rank || !d_nnz || !o_nnz)
        FD_FAIL("fd_csr_split_mpiaij: bad arguments");
    hipStream_t s = fd::st(s_);
    const size_t n1 = (size_t)nrows_owned + 1;
    {
        // the two parts keep PETSc's 32-bit row starts (PetscInt without --with-64-bit-indices, pyop2/datatypes.py:6-10): a local
        // pattern of 2^31 entries or more would overflow the scans below -- refused like the other 32-bit consumers of a pattern
        fd_nnz_t last = 0;
        FD_HIP(hipMemcpyAsync(&last, rowptr + nrows_owned, sizeof(fd_nnz_t), hipMemcpyDeviceToHost, s));
        FD_HIP(hipStreamSynchronize(s));
        if (last > (fd_nnz_t)INT32_MAX) FD_FAIL("fd_csr_split_mpiaij: the owned rows hold 2^31 entries or more; the MPIAIJ parts have 32-bit row starts");
    }
    int32_t *dc = nullptr, *oc = nullptr, *drp = nullptr, *orp = nullptr, *dci = nullptr, *oci = nullptr, *ork = nullptr;
    void *tmp = nullptr;
    FD_HIP(hipMalloc(&dc, n1 * 4)); FD_HIP(hipMalloc(&oc, n1 * 4));
    FD_HIP(hipMalloc(&drp, n1 * 4)); FD_HIP(hipMalloc(&orp, n1 * 4));
    hipLaunchKernelGGL(split_counts, dim3(grid_for(nrows_owned > 0 ? nrows_owned : 1)), dim3(256), 0, s, nrows_owned, rowptr, colidx,
                       ncols_owned, dc, oc);
    FD_CHECK_LAUNCH();
    size_t tb = 0;
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, tb, dc, drp, (int)n1, s));
    FD_HIP(hipMalloc(&tmp, tb ? tb : 8));
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, dc, drp, (int)n1, s));
    FD_HIP(hipcub::DeviceScan::ExclusiveSum(tmp, tb, oc, orp, (int)n1, s));
    int32_t nd = 0, no = 0;
    FD_HIP(hipMemcpyAsync(&nd, drp + nrows_owned, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipMemcpyAsync(&no, orp + nrows_owned, 4, hipMemcpyDeviceToHost, s));
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipMalloc(&dci, (size_t)(nd > 0 ? nd : 1) * 4));
    FD_HIP(hipMalloc(&oci, (size_t)(no > 0 ? no : 1) * 4));
    FD_HIP(hipMalloc(&ork, (size_t)(no > 0 ? no : 1) * 4));
    if (nrows_owned > 0) {
        hipLaunchKernelGGL((split_fill<true, false>), dim3(grid_for((int64_t)nrows_owned * 64)), dim3(256), 0, s, nrows_owned, rowptr, colidx,
                           (const double *)nullptr, col_global, drp, orp, dci, oci, (double *)nullptr, (double *)nullptr, ork);
        FD_CHECK_LAUNCH();
    }
    FD_HIP(hipStreamSynchronize(s));
    FD_HIP(hipFree(dc)); FD_HIP(hipFree(oc)); FD_HIP(hipFree(tmp));
    *d_rowptr = drp; *d_colidx = dci; *d_nnz = nd;
    *o_rowptr = orp; *o_colidx = oci; *o_rank = ork; *o_nnz = no;
    return 0;
}

int fd_csr_split_values(int32_t nrows_owned, const fd_nnz_t *rowptr, const double *vals, const int32_t *d_rowptr,
                        const int32_t *o_rowptr, const int32_t *o_rank, double *d_vals, double *o_vals, fd_stream_t s) {
    if (nrows_owned <= 0) return 0;
    hipLaunchKernelGGL((split_fill<false, true>), dim3(grid_for((int64_t)nrows_owned * 64)), dim3(256), 0, fd::st(s), nrows_owned, rowptr,
                       (const int32_t *)nullptr, vals, (const int32_t *)nullptr, d_rowptr, o_rowptr, (int32_t *)nullptr, (int32_t *)nullptr,
                       d_vals, o_vals, const_cast<int32_t *>(o_rank));
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_get_diagonal(int32_t nrows, const fd_nnz_t *rowptr, const int32_t *colidx, const double *vals, double *diag, fd_stream_t s) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(get_diag, dim3(grid_for(nrows)), dim3(256), 0, fd::st(s), nrows, rowptr, colidx, vals, diag);
    FD_CHECK_LAUNCH();
    return 0;
}

int fd_csr_spmv(int32_t nrows, const fd_nnz_t *rowptr, const int32_t *colidx, const double *vals, const double *x,
                double *y, fd_stream_t s) {
    if (nrows <= 0) return 0;
    hipLaunchKernelGGL(spmv, dim3(grid_for((int64_t)nrows * 64)), dim3(256), 0, fd::st(s), nrows, rowptr, colidx, vals, x, y);
    FD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
