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
 *                        cell layers (constant layers, bottom = 0)
 *   roffs[k], coffs[k] : per-map-entry extruded offsets (NULL when nlayers==0)
 *   region[k]          : iteration region of the pair, oracle numbering ALL=1, ON_BOTTOM=2, ON_TOP=3,
 *                        ON_INTERIOR_FACETS=4 (sparsity.pyx:291-305, 331-346); NULL = ALL everywhere
 *   periodic[k]        : periodic extrusion (sparsity.pyx:273, 343-346); NULL = none
 *   rquot[k], cquot[k] : offset quotients (sparsity.pyx:309-312) or NULL (= 0)
 * Node of entry i (k-th cell of an interior facet: k = i / arity) in layer l, sparsity.pyx:357-368:
 *   map[e][i % arity] + off[i % arity] * ((l + k + quot) % num_layers - quot % num_layers)
 * which for standard extrusion (quot = 0, l + k < num_layers) is map + off*(l + k).
 * Output: rowptr (malloc'ed, nrows+1) and colidx (malloc'ed); returns nnz.
 * Node-level pattern; the caller expands by (rbs, cbs).
 */
typedef struct {
    const int *rmap, *cmap, *roff, *coff, *rquot, *cquot;
    const int *layers;      /* variable layers: (nent, 2) [bottom, top) node levels per entity, else NULL */
    int nent, ar, ac, nl, region, periodic;
} oracle_pair;

static int pair_node(const int *map, const int *off, const int *quot, int arity, int nl, int e, int i, int l)
{
    int k = i / arity, irem = i % arity;
    int v = map[(size_t)e * arity + irem];
    if (nl > 0) {
        int q = quot ? quot[irem] : 0;
        v += off[irem] * ((l + k + q) % nl - q % nl);
    }
    return v;
}

/* layer range (relative to the entity's bottom layer) and number of stacked cells of entity e of a pair
 * (sparsity.pyx:325-346).  Node offsets are taken relative to the entity's own bottom layer, which is what the
 * generated wrapper uses (builder.py:101-103); sparsity.pyx uses the absolute layer modulo num_layers, which
 * enumerates the same set for the ALL region. */
static void pair_layers(const oracle_pair *p, int e, int *l0, int *l1, int *nf, int *nle)
{
    *nf = 1;
    if (p->nl <= 0) { *l0 = 0; *l1 = 1; *nle = 0; return; }
    int nl = p->layers ? p->layers[2 * e + 1] - 1 - p->layers[2 * e] : p->nl;
    *nle = nl;
    *l0 = 0; *l1 = nl;
    if (p->region == 2) *l1 = 1;
    else if (p->region == 3) *l0 = nl - 1;
    else if (p->region == 4) { *nf = 2; if (!p->periodic) *l1 = nl - 1; }
}

long oracle_build_node_sparsity_ex(int nrows, int ncols, int set_diag, int npairs,
                                   const int **rmaps, const int **cmaps,
                                   const int *nent, const int *rarity, const int *carity,
                                   const int *nlayers, const int **roffs, const int **coffs,
                                   const int *region, const int *periodic,
                                   const int **rquots, const int **cquots, const int **layers,
                                   int **rowptr_out, int **colidx_out)
{
    oracle_pair *P = (oracle_pair *)calloc((size_t)(npairs > 0 ? npairs : 1), sizeof(oracle_pair));
    for (int k = 0; k < npairs; ++k) {
        P[k].rmap = rmaps[k]; P[k].cmap = cmaps[k]; P[k].nent = nent[k]; P[k].ar = rarity[k]; P[k].ac = carity[k];
        P[k].nl = nlayers[k] > 0 ? nlayers[k] : 0;
        P[k].roff = P[k].nl ? roffs[k] : NULL; P[k].coff = P[k].nl ? coffs[k] : NULL;
        P[k].region = region ? region[k] : 1; P[k].periodic = periodic ? periodic[k] : 0;
        P[k].rquot = rquots ? rquots[k] : NULL; P[k].cquot = cquots ? cquots[k] : NULL;
        P[k].layers = layers ? layers[k] : NULL;
    }
    long *cnt = (long *)calloc((size_t)nrows + 1, sizeof(long));
    for (int k = 0; k < npairs; ++k) {
        for (int e = 0; e < P[k].nent; ++e) {
            int l0, l1, nf, nle;
            pair_layers(&P[k], e, &l0, &l1, &nf, &nle);
            for (int l = l0; l < l1; ++l)
                for (int i = 0; i < nf * P[k].ar; ++i) {
                    int r = pair_node(P[k].rmap, P[k].roff, P[k].rquot, P[k].ar, nle, e, i, l);
                    if (r < 0 || r >= nrows) continue;
                    cnt[r + 1] += nf * P[k].ac;
                }
        }
    }
    if (set_diag)
        for (int r = 0; r < nrows && r < ncols; ++r) cnt[r + 1] += 1;
    for (int r = 0; r < nrows; ++r) cnt[r + 1] += cnt[r];
    long ncand = cnt[nrows];
    int *cand = (int *)malloc((size_t)(ncand > 0 ? ncand : 1) * sizeof(int));
    long *fill = (long *)malloc((size_t)(nrows > 0 ? nrows : 1) * sizeof(long));
    for (int r = 0; r < nrows; ++r) fill[r] = cnt[r];
    if (set_diag)
        for (int r = 0; r < nrows && r < ncols; ++r) cand[fill[r]++] = r;
    for (int k = 0; k < npairs; ++k) {
        for (int e = 0; e < P[k].nent; ++e) {
            int l0, l1, nf, nle;
            pair_layers(&P[k], e, &l0, &l1, &nf, &nle);
            for (int l = l0; l < l1; ++l)
                for (int i = 0; i < nf * P[k].ar; ++i) {
                    int r = pair_node(P[k].rmap, P[k].roff, P[k].rquot, P[k].ar, nle, e, i, l);
                    if (r < 0 || r >= nrows) continue;
                    for (int j = 0; j < nf * P[k].ac; ++j)   /* negative cols filtered below */
                        cand[fill[r]++] = pair_node(P[k].cmap, P[k].coff, P[k].cquot, P[k].ac, nle, e, j, l);
                }
        }
    }
    free(P);
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

/* iteration region ALL, standard extrusion */
long oracle_build_node_sparsity(int nrows, int ncols, int set_diag, int npairs,
                                const int **rmaps, const int **cmaps,
                                const int *nent, const int *rarity, const int *carity,
                                const int *nlayers, const int **roffs, const int **coffs,
                                int **rowptr_out, int **colidx_out)
{
    return oracle_build_node_sparsity_ex(nrows, ncols, set_diag, npairs, rmaps, cmaps, nent, rarity, carity,
                                         nlayers, roffs, coffs, NULL, NULL, NULL, NULL, NULL, rowptr_out, colidx_out);
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
/* the dropped-entry statistic is kept by the sequential build only: under OpenMP every thread would hammer one counter */
#ifdef _OPENMP
#define ORACLE_COUNT_DROPPED(A) ((void)0)
#else
#define ORACLE_COUNT_DROPPED(A) ((A)->dropped++)
#endif


static inline void add_scalar(oracle_mat *A, int row, int col, double v, int insert)
{
    int lo = A->rowptr[row], hi = A->rowptr[row + 1] - 1;
    while (lo <= hi) {
        int mid = lo + ((hi - lo) >> 1);
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
                    if (rn < 0 || cn < 0) { ORACLE_COUNT_DROPPED(A); continue; }
                    double v = vals[(((size_t)i * rbs + p) * nc + j) * cbs + q];
                    add_scalar(A, rn * rbs + p, cn * cbs + q, v, insert);
                }
            }
    }
    return 0;
}

/* Unrolled variant: rows/cols are scalar dof indices (builder.py:579-581); the lgmaps are then indexed
 * by dof (component-wise boundary conditions mask single dofs of a node). */
int oracle_MatSetValuesLocal(oracle_mat *A, int nr, const int *rows,
                             int nc, const int *cols, const double *vals, int insert)
{
    for (int i = 0; i < nr; ++i) {
        int r = rows[i];
        if (r >= 0 && A->row_lgmap) r = A->row_lgmap[r];
        for (int j = 0; j < nc; ++j) {
            int c = cols[j];
            if (c >= 0 && A->col_lgmap) c = A->col_lgmap[c];
            if (r < 0 || c < 0) { ORACLE_COUNT_DROPPED(A); continue; }
            add_scalar(A, r, c, vals[(size_t)i * nc + j], insert);
        }
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
