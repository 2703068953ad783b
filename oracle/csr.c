/*
 * oracle/csr.c -- TEST INFRASTRUCTURE ONLY (never imported by the product path).
 *
 * CPU restatement of the two PETSc-side services the reference's generated
 * wrapper relies on for matrices:
 *
 *   1. sparsity construction  = pyop2/sparsity.pyx:105-159 (build_sparsity) and
 *      :162-389 (fill_with_zeros): union over all (rowmap, colmap) pairs of the
 *      outer product of each map row, plus the diagonal of square blocks
 *      (sparsity.pyx:198-203), for non-extruded sets and constant-layer
 *      extruded sets (sparsity.pyx:270-371, iteration region ALL).
 *   2. insertion              = PETSc MatSetValuesBlockedLocal / MatSetValuesLocal
 *      with ADD_VALUES / INSERT_VALUES as emitted by pyop2/codegen/builder.py:573-625:
 *      local->global through an lgmap, negative indices silently dropped
 *      (that is how boundary conditions mask rows/cols: pyop2/parloop.py:279-302,
 *      firedrake/functionspaceimpl.py:913-926), row located by search, value added.
 *
 * PETSc itself is third-party and absent from /root/reference (pyproject pins
 * petsc4py 3.25); the behaviour restated here is the documented ADD_VALUES /
 * negative-index-drop contract that the reference's call sites depend on.
 *
 * The matrix is a scalar CSR ("aij"): row = node*rbs + p.
 */
#include <stdlib.h>
#include <string.h>
#include <stdint.h>

typedef struct {
    int nrows;          /* scalar rows */
    int ncols;          /* scalar cols */
    int rbs, cbs;       /* block sizes of the row / col DataSets */
    int *rowptr;        /* nrows+1 */
    int *colidx;        /* nnz, sorted within a row */
    double *vals;       /* nnz */
    const int *row_lgmap; /* per row NODE (blocked) -> node or -1; NULL = identity */
    const int *col_lgmap; /* per col NODE */
    long dropped;       /* number of scalar entries dropped by negative indices */
    long missing;       /* entries that were not in the sparsity (error) */
} oracle_mat;

static int cmp_int(const void *a, const void *b)
{
    int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

/* ---- sparsity ---------------------------------------------------------- */

/*
 * Build the block (node) sparsity from npairs (rmap, cmap) pairs.
 *   rmaps[k], cmaps[k] : (nent[k], rarity[k]) / (nent[k], carity[k]) int32, row-major
 *   nlayers[k]         : 0 for a non-extruded iteration set, else number of
 *                        cell layers; node = map + offset*layer (builder.py:94-124)
 *   roffs[k], coffs[k] : per-map-entry extruded offsets (NULL when nlayers==0)
 * Output: rowptr (malloc'ed, nrows+1) and colidx (malloc'ed); returns nnz.
 * Node-level pattern; the caller expands by (rbs, cbs).
 */
long oracle_build_node_sparsity(int nrows, int ncols, int set_diag, int npairs,
                                const int **rmaps, const int **cmaps,
                                const int *nent, const int *rarity, const int *carity,
                                const int *nlayers, const int **roffs, const int **coffs,
                                int **rowptr_out, int **colidx_out)
{
    long *cnt = (long *)calloc((size_t)nrows + 1, sizeof(long));
    for (int k = 0; k < npairs; ++k) {
        int L = nlayers[k] > 0 ? nlayers[k] : 1;
        for (int e = 0; e < nent[k]; ++e)
            for (int l = 0; l < L; ++l)
                for (int i = 0; i < rarity[k]; ++i) {
                    int r = rmaps[k][(size_t)e * rarity[k] + i];
                    if (nlayers[k] > 0) r += roffs[k][i] * l;
                    if (r < 0 || r >= nrows) continue;
                    cnt[r + 1] += carity[k];
                }
    }
    if (set_diag)
        for (int r = 0; r < nrows && r < ncols; ++r) cnt[r + 1] += 1;
    for (int r = 0; r < nrows; ++r) cnt[r + 1] += cnt[r];
    long ncand = cnt[nrows];
    int *cand = (int *)malloc((size_t)(ncand > 0 ? ncand : 1) * sizeof(int));
    long *fill = (long *)malloc((size_t)nrows * sizeof(long));
    for (int r = 0; r < nrows; ++r) fill[r] = cnt[r];
    if (set_diag)
        for (int r = 0; r < nrows && r < ncols; ++r) cand[fill[r]++] = r;
    for (int k = 0; k < npairs; ++k) {
        int L = nlayers[k] > 0 ? nlayers[k] : 1;
        for (int e = 0; e < nent[k]; ++e)
            for (int l = 0; l < L; ++l)
                for (int i = 0; i < rarity[k]; ++i) {
                    int r = rmaps[k][(size_t)e * rarity[k] + i];
                    if (nlayers[k] > 0) r += roffs[k][i] * l;
                    if (r < 0 || r >= nrows) continue;
                    for (int j = 0; j < carity[k]; ++j) {
                        int c = cmaps[k][(size_t)e * carity[k] + j];
                        if (nlayers[k] > 0) c += coffs[k][j] * l;
                        cand[fill[r]++] = c;   /* negative cols filtered below */
                    }
                }
    }
    int *rowptr = (int *)malloc(((size_t)nrows + 1) * sizeof(int));
    long nnz = 0;
    rowptr[0] = 0;
    for (int r = 0; r < nrows; ++r) {
        int *row = cand + cnt[r];
        long n = cnt[r + 1] - cnt[r];
        qsort(row, (size_t)n, sizeof(int), cmp_int);
        long w = nnz;
        for (long q = 0; q < n; ++q) {
            if (row[q] < 0 || row[q] >= ncols) continue;
            if (w > nnz && cand[w - 1] == row[q]) continue;
            cand[w++] = row[q];            /* compaction in place (w <= cnt[r]+q) */
        }
        nnz = w;
        rowptr[r + 1] = (int)nnz;
    }
    int *colidx = (int *)malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
    memcpy(colidx, cand, (size_t)nnz * sizeof(int));
    free(cand); free(fill); free(cnt);
    *rowptr_out = rowptr;
    *colidx_out = colidx;
    return nnz;
}

void oracle_free(void *p) { free(p); }

/* Expand a node pattern to the scalar (aij) pattern for block sizes (rbs, cbs). */
void oracle_expand_blocks(int nnode_rows, const int *nrowptr, const int *ncolidx,
                          int rbs, int cbs, int *rowptr, int *colidx)
{
    long w = 0;
    rowptr[0] = 0;
    for (int r = 0; r < nnode_rows; ++r)
        for (int p = 0; p < rbs; ++p) {
            for (int q = nrowptr[r]; q < nrowptr[r + 1]; ++q)
                for (int c = 0; c < cbs; ++c)
                    colidx[w++] = ncolidx[q] * cbs + c;
            rowptr[r * rbs + p + 1] = (int)w;
        }
}

/* ---- insertion --------------------------------------------------------- */

static inline void add_scalar(oracle_mat *A, int row, int col, double v, int insert)
{
    int lo = A->rowptr[row], hi = A->rowptr[row + 1] - 1;
    while (lo <= hi) {
        int mid = (lo + hi) >> 1;
        int c = A->colidx[mid];
        if (c == col) {
            if (insert) A->vals[mid] = v;
            else {
#ifdef _OPENMP
#pragma omp atomic
#endif
                A->vals[mid] += v;
            }
            return;
        }
        if (c < col) lo = mid + 1; else hi = mid - 1;
    }
    A->missing++;
}

/* vals laid out [nr][rbs][nc][cbs] (builder.py:538-548, 601-623). */
int oracle_MatSetValuesBlockedLocal(oracle_mat *A, int nr, const int *rows,
                                    int nc, const int *cols, const double *vals,
                                    int insert)
{
    const int rbs = A->rbs, cbs = A->cbs;
    for (int i = 0; i < nr; ++i) {
        int rn = rows[i];
        if (rn >= 0 && A->row_lgmap) rn = A->row_lgmap[rn];
        for (int p = 0; p < rbs; ++p)
            for (int j = 0; j < nc; ++j) {
                int cn = cols[j];
                if (cn >= 0 && A->col_lgmap) cn = A->col_lgmap[cn];
                for (int q = 0; q < cbs; ++q) {
                    if (rn < 0 || cn < 0) { A->dropped++; continue; }
                    double v = vals[(((size_t)i * rbs + p) * nc + j) * cbs + q];
                    add_scalar(A, rn * rbs + p, cn * cbs + q, v, insert);
                }
            }
    }
    return 0;
}

/* Unrolled variant: rows/cols are scalar dof indices (builder.py:579-581). */
int oracle_MatSetValuesLocal(oracle_mat *A, int nr, const int *rows,
                             int nc, const int *cols, const double *vals, int insert)
{
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) {
            if (rows[i] < 0 || cols[j] < 0) { A->dropped++; continue; }
            add_scalar(A, rows[i], cols[j], vals[(size_t)i * nc + j], insert);
        }
    return 0;
}

/* pyop2/types/mat.py:896-937 set_local_diagonal_entries: A[r,r] = v for rows. */
void oracle_set_diagonal(oracle_mat *A, int n, const int *rows, double v)
{
    for (int k = 0; k < n; ++k)
        if (rows[k] >= 0 && rows[k] < A->nrows) add_scalar(A, rows[k], rows[k], v, 1);
}

/* pyop2/types/mat.py:857-891 zero_rows: zero the rows, put v on the diagonal. */
void oracle_zero_rows(oracle_mat *A, int n, const int *rows, double v)
{
    for (int k = 0; k < n; ++k) {
        int r = rows[k];
        for (int q = A->rowptr[r]; q < A->rowptr[r + 1]; ++q)
            A->vals[q] = (A->colidx[q] == r) ? v : 0.0;
    }
}
