"""GPU parity against the reference's own PyOP2 golden vectors and against the oracle.

Mirrors tests/pyop2/test_matrices.py:637-757, test_indirect_loop.py:134-298,
test_direct_loop.py, test_subset.py, test_extrusion.py:344-450, test_global_reduction.py --
same meshes, same data, same tolerances -- with every parloop executed by the HIP backend
through the C ABI.  Each case also runs the identical C kernel through the oracle.
"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import op2
from firedrake_amd.configuration import configuration
import golden_kernels as gk
from helpers import oracle_run

pytestmark = pytest.mark.gpu
G = gk.GOLD
MODES = ["staged", "direct"]


@pytest.fixture(params=MODES)
def mode(request, monkeypatch):
    monkeypatch.setitem(configuration, "mode", "auto" if request.param == "staged" else "direct")
    return request.param


@pytest.fixture(params=["table", "search"])
def scatter(request, monkeypatch):
    monkeypatch.setitem(configuration, "mat_scatter", request.param)
    return request.param


@pytest.fixture
def mesh():
    nodes, ele = op2.Set(4, "nodes"), op2.Set(2, "elements")
    return nodes, ele, op2.Map(ele, nodes, 3, gk.ELEM_NODE, "elem_node")


def _mat(nodes, m, cdim=1):
    return op2.Mat(op2.Sparsity((nodes ** cdim, nodes ** cdim), [(m, m, None)]), np.float64)


@pytest.mark.parametrize("src,name", [(gk.MASS_Q6, "mass_q6"), (gk.MASS_AFFINE, "mass_affine")], ids=["q6", "affine"])
def test_assemble_mat(mesh, mode, scatter, src, name):
    nodes, ele, m = mesh
    mat = _mat(nodes, m)
    x = op2.Dat(nodes ** 2, gk.COORDS)
    k = op2.Kernel(src, name)
    mat.zero()
    op2.par_loop(k, ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    mat.assemble()
    assert_allclose(mat.values, np.array(G["expected_matrix"]), rtol=G["expected_matrix_rtol"], atol=1e-7)
    ocsr, _ = oracle_run(k, ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    assert_allclose(mat.values, ocsr.todense(), rtol=1e-13, atol=1e-15)


def test_assemble_rhs(mesh, mode):
    nodes, ele, m = mesh
    b, x, f = op2.Dat(nodes), op2.Dat(nodes ** 2, gk.COORDS), op2.Dat(nodes, gk.F)
    k = op2.Kernel(gk.RHS_Q6, "rhs_q6")
    b.zero()
    op2.par_loop(k, ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    assert_allclose(b.data, G["expected_rhs"], rtol=G["expected_rhs_rtol_quad6"])
    ob = oracle_run(k, ele, op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))[0]
    assert_allclose(b.data, ob, rtol=1e-14)


def test_rhs_affine(mesh, mode):
    nodes, ele, m = mesh
    b, x, f = op2.Dat(nodes), op2.Dat(nodes ** 2, gk.COORDS), op2.Dat(nodes, gk.F)
    op2.par_loop(op2.Kernel(gk.RHS_AFFINE, "rhs_affine"), ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    assert_allclose(b.data, G["expected_rhs"], rtol=G["expected_rhs_rtol_quad3"])


def test_solve(mesh, mode):
    nodes, ele, m = mesh
    mat = _mat(nodes, m)
    b, x, f = op2.Dat(nodes), op2.Dat(nodes ** 2, gk.COORDS), op2.Dat(nodes, gk.F)
    op2.par_loop(op2.Kernel(gk.MASS_Q6, "mass_q6"), ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    op2.par_loop(op2.Kernel(gk.RHS_Q6, "rhs_q6"), ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    assert_allclose(np.linalg.solve(mat.values, b.data), gk.F, rtol=G["solve_rtol"])


def test_vector_mat_and_rhs(mesh, mode, scatter):
    nodes, ele, m = mesh
    mat = _mat(nodes, m, 2)
    x = op2.Dat(nodes ** 2, gk.COORDS)
    op2.par_loop(op2.Kernel(gk.MASS_VEC_AFFINE, "mass_vec_affine"), ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    assert_allclose(mat.values, np.array(G["expected_vector_matrix"]), rtol=1e-6, atol=1e-8)
    b, f = op2.Dat(nodes ** 2), op2.Dat(nodes ** 2, gk.F_VEC)
    op2.par_loop(op2.Kernel(gk.RHS_VEC_AFFINE, "rhs_vec_affine"), ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    assert_allclose(b.data, np.array(G["expected_vec_rhs"]), rtol=1e-6)


def test_zero_matrix_and_set_matrix(mesh, mode, scatter):
    nodes, ele, m = mesh
    mat = _mat(nodes, m)
    mat.zero()
    assert_allclose(mat.values, np.zeros((4, 4)), atol=1e-14)
    g = op2.Global(1, 1.0, np.float64, "g")
    inc = op2.Kernel("static void inc9(double e[9], double *g) { for (int i = 0; i < 9; ++i) e[i] += g[0]; }", "inc9")
    st = op2.Kernel("static void set9(double e[9], double *g) { for (int i = 0; i < 9; ++i) e[i] = g[0]; }", "set9")
    op2.par_loop(inc, ele, mat(op2.INC, (m, m)), g(op2.READ))
    mat.assemble()
    assert mat.values.sum() == 3 * 3 * ele.size
    op2.par_loop(st, ele, mat(op2.WRITE, (m, m)), g(op2.READ))
    mat.assemble()
    assert mat.values.sum() == (3 * 3 - 2) * ele.size


def test_zero_rows_and_diagonal(mesh):
    nodes, ele, m = mesh
    mat = _mat(nodes, m)
    x = op2.Dat(nodes ** 2, gk.COORDS)
    op2.par_loop(op2.Kernel(gk.MASS_Q6, "mass_q6"), ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    E = np.array(G["expected_matrix"])
    E[0] = [12.0, 0, 0, 0]
    mat.zero_rows([0], 12.0)
    assert_allclose(mat.values, E, rtol=1e-5, atol=1e-7)
    mat.zero_rows(op2.Subset(nodes, [3]), 4.0)
    E[3] = [0, 0, 0, 4.0]
    assert_allclose(mat.values, E, rtol=1e-5, atol=1e-7)
    for n in (1, 2):
        mt = _mat(nodes, m, n)
        mt.set_local_diagonal_entries(list(range(mt.nblock_rows)))
        mt.assemble()
        assert (mt.values == np.identity(4 * n)).all()


def test_bc_masked_lgmaps(mesh, mode, scatter):
    """BC rows/cols dropped through -1 lgmap entries (pyop2/parloop.py:279-302), then unit diagonal
    (firedrake/assemble.py:1501-1507 -> mat.py:896-937)."""
    nodes, ele, m = mesh
    mat = _mat(nodes, m)
    x = op2.Dat(nodes ** 2, gk.COORDS)
    lg = np.array([-1, 1, 2, 3], dtype=np.int32)
    k = op2.Kernel(gk.MASS_AFFINE, "mass_affine")
    op2.par_loop(k, ele, mat(op2.INC, (m, m), lgmaps=(lg, lg)), x(op2.READ, m))
    mat.set_local_diagonal_entries([0])
    A = mat.values
    ocsr, _ = oracle_run(k, ele, mat(op2.INC, (m, m), lgmaps=(lg, lg)), x(op2.READ, m))
    O = ocsr.todense()
    O[0, 0] = 1.0
    assert_allclose(A, O, rtol=1e-13, atol=1e-15)
    assert A[0, 0] == 1.0 and np.all(A[0, 1:] == 0) and np.all(A[1:, 0] == 0)


def test_sparsity_nnz_and_diagonal():
    s, d, d2 = op2.Set(1), op2.Set(4), op2.Set(4)
    m = op2.Map(s, d, 2, [1, 3])
    m2 = op2.Map(s, d2, 3, [1, 2, 3])
    assert all(op2.Sparsity((d ** 1, d ** 1), [(m, m, None)]).nnz == [1, 2, 1, 2])      # test_matrices.py:556-567
    assert all(op2.Sparsity((d ** 1, d2 ** 1), [(m, m2, None)]).nnz == [0, 3, 0, 3])


# ---- indirect-loop semantics (tests/pyop2/test_indirect_loop.py:134-250) -------------------------
NEL = 4096


@pytest.fixture
def iset():
    it, ind, unit = op2.Set(NEL, "iterset"), op2.Set(NEL, "indset"), op2.Set(1, "unitset")
    rng = np.random.default_rng(0)
    perm = rng.permutation(NEL).astype(np.int32)
    return it, ind, unit, op2.Map(it, ind, 1, perm, "iterset2indset"), op2.Map(it, unit, 1, np.zeros(NEL, np.int32))


def test_onecolor_wo_rw(iset):
    it, ind, unit, i2i, i2u = iset
    x = op2.Dat(ind, np.arange(NEL, dtype=np.uint32), np.uint32)
    op2.par_loop(op2.Kernel("static void kernel_wo(unsigned int* x) { *x = 42; }", "kernel_wo"), it, x(op2.WRITE, i2i))
    assert all(x.data == 42)
    x = op2.Dat(ind, np.arange(NEL, dtype=np.uint32), np.uint32)
    op2.par_loop(op2.Kernel("static void rw(unsigned int* x) { (*x) = (*x) + 1; }", "rw"), it, x(op2.RW, i2i))
    assert x.data.sum() == NEL * (NEL + 1) // 2


def test_indirect_inc(iset, mode):
    it, ind, unit, i2i, i2u = iset
    u = op2.Dat(unit, np.array([0], dtype=np.uint32), np.uint32)
    op2.par_loop(op2.Kernel("static void inc_u(unsigned int* x) { (*x) = (*x) + 1; }", "inc_u"), it, u(op2.INC, i2u))
    assert u.data[0] == NEL


def test_indirect_max_min(iset):
    it, ind, unit, i2i, i2u = iset
    a, b = op2.Dat(ind, dtype=np.int32), op2.Dat(ind, dtype=np.int32)
    a.data[:] = -10
    b.data[:] = -5
    op2.par_loop(op2.Kernel("static void maxify(int *a, int *b) {*a = *a < *b ? *b : *a;}", "maxify"), it,
                 a(op2.MAX, i2i), b(op2.READ, i2i))
    assert np.allclose(a.data_ro, -5)
    a.data[:] = 10
    b.data[:] = 5
    op2.par_loop(op2.Kernel("static void minify(int *a, int *b) {*a = *a > *b ? *b : *a;}", "minify"), it,
                 a(op2.MIN, i2i), b(op2.READ, i2i))
    assert np.allclose(a.data_ro, 5)


def test_global_read_and_inc(iset):
    it, ind, unit, i2i, i2u = iset
    x = op2.Dat(ind, np.arange(NEL, dtype=np.uint32), np.uint32)
    g = op2.Global(1, 2, np.uint32, "g")
    op2.par_loop(op2.Kernel("static void global_read(unsigned int* x, unsigned int* g) { (*x) /= (*g); }", "global_read"),
                 it, x(op2.RW, i2i), g(op2.READ))
    assert x.data.sum() == sum(v // 2 for v in range(NEL))
    x = op2.Dat(ind, np.arange(NEL, dtype=np.uint32), np.uint32)
    g = op2.Global(1, 0, np.uint32, "g")
    op2.par_loop(op2.Kernel("static void global_inc(unsigned int *x, unsigned int *inc) { (*x) = (*x) + 1; (*inc) += (*x); }",
                            "global_inc"), it, x(op2.RW, i2i), g(op2.INC))
    assert x.data.sum() == NEL * (NEL + 1) // 2
    assert g.data[0] == NEL * (NEL + 1) // 2


def test_global_fp64_reductions(iset, mode):
    it, ind, unit, i2i, i2u = iset
    rng = np.random.default_rng(1)
    v = rng.standard_normal(NEL)
    x = op2.Dat(ind, v)
    gs, gmin, gmax = op2.Global(1, 0.0), op2.Global(1, 1e30), op2.Global(1, -1e30)
    op2.par_loop(op2.Kernel("static void red(const double *x, double *s, double *mn, double *mx) { s[0] += x[0]; "
                            "if (x[0] < mn[0]) mn[0] = x[0]; if (x[0] > mx[0]) mx[0] = x[0]; }", "red"),
                 it, x(op2.READ, i2i), gs(op2.INC), gmin(op2.MIN), gmax(op2.MAX))
    assert_allclose(gs.data[0], v.sum(), rtol=1e-12, atol=1e-12)
    assert gmin.data[0] == v.min() and gmax.data[0] == v.max()


def test_2d_dat_and_2d_map(mode):
    n = 513
    it, ind = op2.Set(n), op2.Set(n)
    i2i = op2.Map(it, ind, 1, np.arange(n))
    x2 = op2.Dat(ind ** 2, dtype=np.uint32)
    op2.par_loop(op2.Kernel("static void wo2(unsigned int* x) { x[0] = 42; x[1] = 43; }", "wo2"), it, x2(op2.WRITE, i2i))
    assert all(all(v == [42, 43]) for v in x2.data)
    nodes, edges = op2.Set(n), op2.Set(n - 1)
    nv = op2.Dat(nodes, np.arange(n, dtype=np.uint32), np.uint32)
    ev = op2.Dat(edges, np.zeros(n - 1, dtype=np.uint32), np.uint32)
    e2n = op2.Map(edges, nodes, 2, np.array([(i, i + 1) for i in range(n - 1)]))
    op2.par_loop(op2.Kernel("static void sum2(unsigned int *edge, unsigned int *nodes) { *edge = nodes[0] + nodes[1]; }", "sum2"),
                 edges, ev(op2.WRITE), nv(op2.READ, e2n))
    assert all(np.arange(1, (n - 1) * 2 + 1, 2) == ev.data)


def test_permuted_map():
    fromset, toset = op2.Set(1), op2.Set(4)
    d1, d2 = op2.Dat(op2.DataSet(toset, 1), dtype=np.int32), op2.Dat(op2.DataSet(toset, 1), dtype=np.int32)
    d1.data[:] = np.arange(4, dtype=np.int32)
    k = op2.Kernel("void copy4(int *to, const int * restrict from) { for (int i = 0; i < 4; i++) { to[i] = from[i]; } }", "copy4")
    m1 = op2.Map(fromset, toset, 4, values=[1, 2, 3, 0])
    m2 = op2.PermutedMap(m1, [3, 2, 0, 1])
    op2.par_loop(k, fromset, d2(op2.WRITE, m2), d1(op2.READ, m1))
    expect = np.empty_like(d1.data)
    expect[m1.values[..., m2.permutation]] = d1.data[m1.values]
    assert (d1.data == np.arange(4, dtype=np.int32)).all()
    assert (d2.data == expect).all()


def test_direct_loops():
    n = 1000
    s = op2.Set(n)
    x = op2.Dat(s, np.arange(n, dtype=np.uint32), np.uint32)
    op2.par_loop(op2.Kernel("static void wo_d(unsigned int* x) { *x = 42; }", "wo_d"), s, x(op2.WRITE))
    assert all(x.data == 42)
    y = op2.Dat(s, np.arange(n, dtype=np.float64))
    g = op2.Global(1, 0.0)
    op2.par_loop(op2.Kernel("static void dsum(double *x, double *g) { x[0] *= 2.0; g[0] += x[0]; }", "dsum"), s, y(op2.RW), g(op2.INC))
    assert_allclose(y.data, 2.0 * np.arange(n))
    assert_allclose(g.data[0], n * (n - 1.0))


def test_subset_indirect_and_direct():
    n = 200
    it, ind = op2.Set(n), op2.Set(n)
    m = op2.Map(it, ind, 1, np.arange(n)[::-1].copy())
    ss = op2.Subset(it, np.arange(0, n, 2))
    d = op2.Dat(ind)
    op2.par_loop(op2.Kernel("static void one(double *x) { *x += 1.0; }", "one"), ss, d(op2.INC, m))
    exp = np.zeros(n)
    exp[m.values[::2, 0]] = 1
    assert np.all(d.data == exp)
    e = op2.Dat(it)
    op2.par_loop(op2.Kernel("static void two(double *x) { *x = 2.0; }", "two"), ss, e(op2.WRITE))
    assert np.all(e.data[::2] == 2) and np.all(e.data[1::2] == 0)


# ---- extrusion (tests/pyop2/test_extrusion.py:344-450) ------------------------------------------
def _extruded_strip():
    ex = G["extrusion"]
    nel, layers = ex["nelems"], ex["layers"]
    nx = nel // 2
    base_xy = np.array([(i, j) for j in range(2) for i in range(nx + 1)], dtype=np.float64)
    cells = []
    for i in range(nx):
        a, b, c, d = i, i + 1, nx + 1 + i, nx + 2 + i
        cells += [(a, b, c), (b, d, c)]
    cells = np.array(cells, dtype=np.int32)
    nb = len(base_xy)
    coords = np.repeat(base_xy, layers, axis=0)
    cmap = np.empty((nel, 6), dtype=np.int32)
    for e in range(nel):
        for k in range(3):
            cmap[e, 2 * k] = cells[e, k] * layers
            cmap[e, 2 * k + 1] = cells[e, k] * layers + 1
    base = op2.Set(nel)
    ext = op2.ExtrudedSet(base, layers=layers)
    nodes = op2.Set(nb * layers)
    fset = op2.Set(nel * (layers - 1))
    return (ex, base, ext, op2.Dat(nodes ** 2, coords), op2.Map(ext, nodes, 6, cmap, offset=[1] * 6),
            op2.Dat(fset, np.ones(nel * (layers - 1))),
            op2.Map(ext, fset, 1, np.arange(nel) * (layers - 1), offset=[1]))


def test_extrusion_volume():
    ex, base, ext, coords, cmap, field, fmap = _extruded_strip()
    g = op2.Global(1, data=0.0, name="g")
    k = op2.Kernel("""static void comp_vol(double A[1], const double x[12], const double y[1]) {
      double a = x[0]*(x[5]-x[9]) + x[4]*(x[9]-x[1]) + x[8]*(x[1]-x[5]);
      if (a < 0) a = -a;
      A[0] += 0.5*a*0.1*y[0]; }""", "comp_vol")
    op2.par_loop(k, ext, g(op2.INC), coords(op2.READ, cmap), field(op2.READ, fmap))
    assert int(g.data[0] + 1e-9) == ex["expected_int_volume"]
    og = oracle_run(k, ext, op2.Global(1, 0.0)(op2.INC), coords(op2.READ, cmap), field(op2.READ, fmap))[0]
    assert_allclose(g.data[0], og[0], rtol=1e-13)


def test_extruded_direct_inc_layer_arg_and_write():
    ex, base, ext, coords, cmap, field, fmap = _extruded_strip()
    dat = op2.Dat(base)
    op2.par_loop(op2.Kernel("static void k1(double *x) { *x += 1.0; }", "k1"), op2.ExtrudedSet(base, layers=10), dat(op2.INC))
    assert np.allclose(dat.data, 9.0)
    op2.par_loop(op2.Kernel("static void blah(double* x, int layer_arg){ x[0] = layer_arg; }", "blah"), ext,
                 field(op2.WRITE, fmap), pass_layer_arg=True)
    L = ex["layers"] - 1
    assert np.all(field.data.reshape(-1, L) == np.arange(L))
    op2.par_loop(op2.Kernel("static void wo42(double* x) { x[0] = 42.0; }", "wo42"), ext, field(op2.WRITE, fmap))
    assert np.all(field.data == 42)


def test_extruded_mat_and_interior_facets():
    """Column-structured matrix assembly incl. the ON_INTERIOR_FACETS double pack (builder.py:790-831)."""
    nb, layers = 5, 4
    base = op2.Set(nb)
    ext = op2.ExtrudedSet(base, layers=layers)
    nodes = op2.Set((nb + 1) * layers)
    vm = np.array([[i * layers, i * layers + 1, (i + 1) * layers, (i + 1) * layers + 1] for i in range(nb)], dtype=np.int32)
    m = op2.Map(ext, nodes, 4, vm, offset=[1, 1, 1, 1])
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    g = op2.Global(1, 1.0)
    k = op2.Kernel("static void ones16(double A[16], const double *g) { for (int i = 0; i < 16; ++i) A[i] += g[0] * (i + 1); }", "ones16")
    op2.par_loop(k, ext, mat(op2.INC, (m, m)), g(op2.READ))
    ocsr = oracle_run(k, ext, mat(op2.INC, (m, m)), g(op2.READ))[0]
    assert_allclose(mat.values, ocsr.todense(), rtol=1e-14)
    d = op2.Dat(nodes)
    kf = op2.Kernel("static void ifk(double *y) { for (int i = 0; i < 8; ++i) y[i] += i + 1.0; }", "ifk")
    op2.par_loop(kf, ext, d(op2.INC, m), iteration_region=op2.ON_INTERIOR_FACETS)
    od = oracle_run(kf, ext, op2.Dat(nodes)(op2.INC, m), iteration_region=op2.ON_INTERIOR_FACETS)[0]
    assert_allclose(d.data, od, rtol=1e-14)
    for reg in (op2.ON_BOTTOM, op2.ON_TOP):
        d2 = op2.Dat(nodes)
        kb = op2.Kernel("static void bt(double *y) { for (int i = 0; i < 4; ++i) y[i] += 1.0; }", "bt")
        op2.par_loop(kb, ext, d2(op2.INC, m), iteration_region=reg)
        o2 = oracle_run(kb, ext, op2.Dat(nodes)(op2.INC, m), iteration_region=reg)[0]
        assert_allclose(d2.data, o2)


# ---- composed maps (tests/pyop2/test_indirect_loop.py:320-380) ------------------------------------
@pytest.mark.parametrize("permuted", ["none", "pre"])
def test_composed_map_two_maps(permuted):
    setB, nodesetB = op2.Set(3), op2.Set(6)
    datB = op2.Dat(op2.DataSet(nodesetB, 1), dtype=np.float64)
    mapB = op2.Map(setB, nodesetB, 2, values=[[0, 1], [2, 3], [4, 5]])
    setA, nodesetA = op2.Set(5), op2.Set(8)
    datA = op2.Dat(op2.DataSet(nodesetA, 1), np.array([.0, .1, .2, .3, .4, .5, .6, .7]))
    mapA0 = op2.Map(setA, nodesetA, 2, values=[[0, 1], [2, 3], [4, 5], [6, 7], [0, 1]])
    if permuted == "pre":
        mapA0 = op2.PermutedMap(mapA0, [1, 0])
    mapA = op2.ComposedMap(mapA0, op2.Map(setB, setA, 1, values=[3, 1, 2]))
    k = op2.Kernel("void copy2(double *to, const double * restrict from) { for (int i = 0; i < 2; ++i) { to[i] = from[i]; } }", "copy2")
    op2.par_loop(k, setB, datB(op2.WRITE, mapB), datA(op2.READ, mapA))
    expect = [.6, .7, .2, .3, .4, .5] if permuted == "none" else [.7, .6, .3, .2, .5, .4]
    assert (datB.data == np.array(expect)).all()


@pytest.mark.parametrize("nested", ["none", "first", "last"])
@pytest.mark.parametrize("subset", [False, True])
def test_composed_map_three_maps(nested, subset):
    setC, nodesetC = op2.Set(2), op2.Set(4)
    datC = op2.Dat(op2.DataSet(nodesetC, 1), dtype=np.float64)
    mapC = op2.Map(setC, nodesetC, 2, values=[[0, 1], [2, 3]])
    setB, setA, nodesetA = op2.Set(3), op2.Set(5), op2.Set(8)
    datA = op2.Dat(op2.DataSet(nodesetA, 1), np.array([.0, .1, .2, .3, .4, .5, .6, .7]))
    mapA0 = op2.Map(setA, nodesetA, 2, values=[[0, 1], [2, 3], [4, 5], [6, 7], [0, 1]])
    mapA1 = op2.Map(setB, setA, 1, values=[3, 1, 2])
    mapA2 = op2.Map(setC, setB, 1, values=[2, 0])
    mapA = {"none": lambda: op2.ComposedMap(mapA0, mapA1, mapA2),
            "first": lambda: op2.ComposedMap(op2.ComposedMap(mapA0, mapA1), mapA2),
            "last": lambda: op2.ComposedMap(mapA0, op2.ComposedMap(mapA1, mapA2))}[nested]()
    k = op2.Kernel("void copy2(double *to, const double * restrict from) { for (int i = 0; i < 2; ++i) { to[i] = from[i]; } }", "copy2")
    it = op2.Subset(setC, np.array([1], dtype=np.int32)) if subset else setC
    op2.par_loop(k, it, datC(op2.WRITE, mapC), datA(op2.READ, mapA))
    expect = [.0, .0, .6, .7] if subset else [.4, .5, .6, .7]
    assert (datC.data == np.array(expect)).all()
