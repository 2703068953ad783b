"""Pin the oracle (CPU restatement of the PyOP2 wrapper) against the reference's own
golden vectors -- tests/pyop2/test_matrices.py:463-501,637-733, test_extrusion.py:344-360,
test_indirect_loop.py:134-250, test_subset.py.  CPU only."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import ODat, OGlobal, OMat, READ, WRITE, RW, INC, MIN, MAX
import golden_kernels as gk

G = gk.GOLD


def _mat(cdim=1):
    return oracle.build_sparsity(4, 4, [(gk.ELEM_NODE, gk.ELEM_NODE)], rbs=cdim, cbs=cdim)


def test_sparsity_always_has_diagonal():
    # tests/pyop2/test_matrices.py:597-607 (sparsity.pyx:198-203)
    m = np.array([[2]], dtype=np.int32)
    m2 = np.array([[1]], dtype=np.int32)
    s = oracle.build_sparsity(4, 3, [(m, m2)])
    A = s.todense()
    assert A.shape == (4, 3)
    nnz_rows = np.diff(s.rowptr)
    assert list(nnz_rows) == [1, 1, 2, 0]       # diagonal for r<ncols plus (2,1)


def test_sparsity_nnz():
    # tests/pyop2/test_matrices.py:556-567: nnz == [1,2,1,2] and [0,3,0,3]
    m = np.array([[1, 3]], dtype=np.int32)
    m2 = np.array([[1, 2, 3]], dtype=np.int32)
    s = oracle.build_sparsity(4, 4, [(m, m)])
    assert list(np.diff(s.rowptr)) == [1, 2, 1, 2]
    s2 = oracle.build_sparsity(4, 4, [(m, m2)], set_diag=False)
    assert list(np.diff(s2.rowptr)) == [0, 3, 0, 3]


@pytest.mark.parametrize("src,name", [(gk.MASS_Q6, "mass_q6"), (gk.MASS_AFFINE, "mass_affine")], ids=["q6", "affine"])
def test_assemble_mat(src, name):
    csr = _mat()
    oracle.par_loop(src, name, 0, 2, [OMat(csr, INC, gk.ELEM_NODE, gk.ELEM_NODE),
                                     ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE)])
    assert_allclose(csr.todense(), np.array(G["expected_matrix"]), rtol=G["expected_matrix_rtol"], atol=1e-7)


def test_assemble_rhs_q6():
    b = np.zeros(4)
    oracle.par_loop(gk.RHS_Q6, "rhs_q6", 0, 2, [ODat(b, INC, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE),
                                                ODat(gk.F.copy(), READ, gk.ELEM_NODE)])
    assert_allclose(b, G["expected_rhs"], rtol=G["expected_rhs_rtol_quad6"])


def test_assemble_rhs_affine():
    b = np.zeros(4)
    oracle.par_loop(gk.RHS_AFFINE, "rhs_affine", 0, 2, [ODat(b, INC, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE),
                                                        ODat(gk.F.copy(), READ, gk.ELEM_NODE)])
    assert_allclose(b, G["expected_rhs"], rtol=G["expected_rhs_rtol_quad3"])


def test_solve_consistency():
    # tests/pyop2/test_matrices.py:660-665: solve(M, b) == f
    csr = _mat()
    b = np.zeros(4)
    oracle.par_loop(gk.MASS_Q6, "mass_q6", 0, 2, [OMat(csr, INC, gk.ELEM_NODE, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE)])
    oracle.par_loop(gk.RHS_Q6, "rhs_q6", 0, 2, [ODat(b, INC, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE),
                                                ODat(gk.F.copy(), READ, gk.ELEM_NODE)])
    assert_allclose(np.linalg.solve(csr.todense(), b), gk.F, rtol=G["solve_rtol"])


def test_vector_mat_and_rhs():
    csr = _mat(2)
    oracle.par_loop(gk.MASS_VEC_AFFINE, "mass_vec_affine", 0, 2,
                    [OMat(csr, INC, gk.ELEM_NODE, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE)])
    assert_allclose(csr.todense(), np.array(G["expected_vector_matrix"]), rtol=1e-6, atol=1e-8)
    b = np.zeros((4, 2))
    oracle.par_loop(gk.RHS_VEC_AFFINE, "rhs_vec_affine", 0, 2,
                    [ODat(b, INC, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE), ODat(gk.F_VEC.copy(), READ, gk.ELEM_NODE)])
    assert_allclose(b, np.array(G["expected_vec_rhs"]), rtol=1e-6)


def test_set_matrix_write_vs_inc():
    # tests/pyop2/test_matrices.py:678-695
    csr = _mat()
    g = np.array([1.0])
    inc = "static void inc9(double e[9], double *g) { for (int i = 0; i < 9; ++i) e[i] += g[0]; }"
    st = "static void set9(double e[9], double *g) { for (int i = 0; i < 9; ++i) e[i] = g[0]; }"
    oracle.par_loop(inc, "inc9", 0, 2, [OMat(csr, INC, gk.ELEM_NODE, gk.ELEM_NODE), OGlobal(g, READ)])
    assert csr.values.sum() == 3 * 3 * 2
    oracle.par_loop(st, "set9", 0, 2, [OMat(csr, WRITE, gk.ELEM_NODE, gk.ELEM_NODE), OGlobal(g, READ)])
    assert csr.values.sum() == (3 * 3 - 2) * 2


def test_bc_rows_dropped_by_negative_lgmap():
    # pyop2/parloop.py:279-302 + PETSc negative-index drop; then mat.py:896-937 diagonal
    csr = _mat()
    lg = np.array([-1, 1, 2, 3], dtype=np.int32)
    m = OMat(csr, INC, gk.ELEM_NODE, gk.ELEM_NODE, row_lgmap=lg, col_lgmap=lg)
    oracle.par_loop(gk.MASS_AFFINE, "mass_affine", 0, 2, [m, ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE)])
    A = csr.todense()
    E = np.array(G["expected_matrix"])
    assert np.all(A[0] == 0) and np.all(A[:, 0] == 0)
    assert_allclose(A[1:, 1:], E[1:, 1:], rtol=1e-5, atol=1e-7)
    assert m.stats["dropped"] > 0


def test_indirect_inc_min_max_rw_write():
    # tests/pyop2/test_indirect_loop.py:134-176
    n = 64
    idmap = np.arange(n, dtype=np.int32).reshape(n, 1)
    zmap = np.zeros((n, 1), dtype=np.int32)
    x = np.arange(n, dtype=np.uint32)
    oracle.par_loop("static void wo(unsigned int *x) { *x = 42; }", "wo", 0, n, [ODat(x, WRITE, idmap)])
    assert np.all(x == 42)
    x = np.arange(n, dtype=np.uint32)
    oracle.par_loop("static void rw(unsigned int *x) { *x = *x + 1; }", "rw", 0, n, [ODat(x, RW, idmap)])
    assert x.sum() == n * (n + 1) // 2
    u = np.zeros(1, dtype=np.uint32)
    oracle.par_loop("static void inc(unsigned int *x) { *x = *x + 1; }", "inc", 0, n, [ODat(u, INC, zmap)])
    assert u[0] == n
    a = np.full(n, -10, dtype=np.int32); b = np.full(n, -5, dtype=np.int32)
    oracle.par_loop("static void mx(int *a, int *b) { *a = *a < *b ? *b : *a; }", "mx", 0, n,
                    [ODat(a, MAX, idmap), ODat(b, READ, idmap)])
    assert np.all(a == -5)
    a = np.full(n, 10, dtype=np.int32); b = np.full(n, 5, dtype=np.int32)
    oracle.par_loop("static void mn(int *a, int *b) { *a = *a > *b ? *b : *a; }", "mn", 0, n,
                    [ODat(a, MIN, idmap), ODat(b, READ, idmap)])
    assert np.all(a == 5)


def test_global_inc_and_read():
    # tests/pyop2/test_indirect_loop.py:178-205
    n = 64
    idmap = np.arange(n, dtype=np.int32).reshape(n, 1)
    x = np.arange(n, dtype=np.uint32)
    g = np.zeros(1, dtype=np.uint32)
    k = "static void gi(unsigned int *x, unsigned int *inc) { (*x) = (*x) + 1; (*inc) += (*x); }"
    oracle.par_loop(k, "gi", 0, n, [ODat(x, RW, idmap), OGlobal(g, INC)])
    assert x.sum() == n * (n + 1) // 2 and g[0] == n * (n + 1) // 2


def test_2d_map_edge_sum():
    # tests/pyop2/test_indirect_loop.py:214-236
    n = 33
    node_vals = np.arange(n, dtype=np.uint32)
    edge_vals = np.zeros(n - 1, dtype=np.uint32)
    e2n = np.array([(i, i + 1) for i in range(n - 1)], dtype=np.int32)
    k = "static void sum2(unsigned int *edge, unsigned int *nodes) { *edge = nodes[0] + nodes[1]; }"
    oracle.par_loop(k, "sum2", 0, n - 1, [ODat(edge_vals, WRITE), ODat(node_vals, READ, e2n)])
    assert np.all(edge_vals == np.arange(1, (n - 1) * 2 + 1, 2))


def test_permuted_map():
    # tests/pyop2/test_indirect_loop.py:280-298
    d1 = np.arange(4, dtype=np.int32); d2 = np.zeros(4, dtype=np.int32)
    m1 = np.array([[1, 2, 3, 0]], dtype=np.int32)
    perm = [3, 2, 0, 1]
    k = "static void cp(int *to, const int *from) { for (int i = 0; i < 4; i++) to[i] = from[i]; }"
    oracle.par_loop(k, "cp", 0, 1, [ODat(d2, WRITE, m1, perm=perm), ODat(d1, READ, m1)])
    expect = np.empty_like(d1)
    expect[m1[0][perm]] = d1[m1[0]]
    assert np.all(d2 == expect)


def test_subset_indirect():
    # tests/pyop2/test_subset.py: only the listed entities are visited
    n = 20
    idmap = np.arange(n, dtype=np.int32).reshape(n, 1)
    x = np.zeros(n, dtype=np.float64)
    ss = np.arange(0, n, 2, dtype=np.int32)
    oracle.par_loop("static void one(double *x) { *x += 1.0; }", "one", 0, len(ss), [ODat(x, INC, idmap)], subset=ss)
    assert np.all(x[::2] == 1) and np.all(x[1::2] == 0)


def test_extruded_volume_and_direct_inc():
    # tests/pyop2/test_extrusion.py:344-360 (known volume) and :366-373 (direct INC over layers)
    ex = G["extrusion"]
    nel, layers = ex["nelems"], ex["layers"]
    # strip of right triangles of area 0.5 (2 per unit square), columns extruded by dz = 0.1
    nx = nel // 2
    base_xy = np.array([(i, j) for j in range(2) for i in range(nx + 1)], dtype=np.float64)
    cells = []
    for i in range(nx):
        a, b, c, d = i, i + 1, nx + 1 + i, nx + 2 + i
        cells += [(a, b, c), (b, d, c)]
    cells = np.array(cells, dtype=np.int32)
    nb = len(base_xy)
    coords = np.zeros((nb * layers, 2))                    # node (v, l) -> v*layers + l ; xy only
    for v in range(nb):
        coords[v * layers:(v + 1) * layers] = base_xy[v]
    cmap = np.empty((nel, 6), dtype=np.int32)              # bottom cell: (v,0),(v,1) per vertex
    for e in range(nel):
        for k in range(3):
            cmap[e, 2 * k] = cells[e, k] * layers
            cmap[e, 2 * k + 1] = cells[e, k] * layers + 1
    field = np.ones((nel * (layers - 1), 1))
    fmap = (np.arange(nel, dtype=np.int32) * (layers - 1)).reshape(nel, 1)
    g = np.zeros(1)
    k = """static void vol(double A[1], const double x[12], const double y[1]) {
      double a = x[0]*(x[5]-x[9]) + x[4]*(x[9]-x[1]) + x[8]*(x[1]-x[5]);
      if (a < 0) a = -a;
      A[0] += 0.5*a*0.1*y[0]; }"""
    oracle.par_loop(k, "vol", 0, nel, [OGlobal(g, INC), ODat(coords, READ, cmap, offset=[1] * 6),
                                       ODat(field, READ, fmap, offset=[1])], layers=(0, layers))
    assert int(round(g[0] * 1e6)) == int(round((layers - 1) * 0.1 * (nel // 2) * 1e6))
    assert int(g[0] + 1e-9) == ex["expected_int_volume"]
    d = np.zeros(nel)
    oracle.par_loop("static void k1(double *x) { *x += 1.0; }", "k1", 0, nel, [ODat(d, INC)], layers=(0, 10))
    assert np.allclose(d, 9.0)


def test_zero_rows():
    # tests/pyop2/test_matrices.py:734-757
    import ctypes
    from oracle.wrapper import _csrlib, _CMat
    csr = _mat()
    oracle.par_loop(gk.MASS_Q6, "mass_q6", 0, 2, [OMat(csr, INC, gk.ELEM_NODE, gk.ELEM_NODE), ODat(gk.COORDS.copy(), READ, gk.ELEM_NODE)])
    cm = _CMat(4, 4, 1, 1, csr.rowptr.ctypes.data, csr.colidx.ctypes.data, csr.values.ctypes.data, None, None, 0, 0)
    rows = np.array([0], dtype=np.int32)
    _csrlib().oracle_zero_rows(ctypes.byref(cm), 1, rows.ctypes.data_as(ctypes.c_void_p), ctypes.c_double(12.0))
    E = np.array(G["expected_matrix"]); E[0] = [12.0, 0, 0, 0]
    assert_allclose(csr.todense(), E, rtol=1e-5, atol=1e-7)
