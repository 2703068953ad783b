"""The reference's direct / indirect / global-reduction loop tests, mirrored:
tests/pyop2/test_global_reduction.py:45-462, test_direct_loop.py:47-236, test_vector_map.py:88-180 and
test_iteration_space_dats.py:78-225 -- same sets (including the (core, owned, total) triples), data, kernels and
assertions.  Every test runs twice: through the HIP backend (``-m gpu``) and, on machines without a GPU, through the
host-sim of the generated direct wrapper (tests/hostsim.py), which writes its results back into the carriers."""
import numpy
import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import op2
from firedrake_amd.parloop import DatParloopArg, GlobalParloopArg

nelems = 4096
nnodes = 4096
nele = nnodes // 2


def _hostsim_par_loop(kernel, iterset, *args, **kw):
    from hostsim import run_direct
    pl = op2.LegacyParloop(kernel, iterset, *args, **kw)
    outs = run_direct(pl)
    for pa, out in zip(pl.arguments, outs):
        if isinstance(pa, (DatParloopArg, GlobalParloopArg)):
            pa.data._host_rw()[...] = out.reshape(pa.data._host.shape)


@pytest.fixture(params=[pytest.param("gpu", marks=pytest.mark.gpu), "hostsim"])
def par_loop(request):
    return op2.par_loop if request.param == "gpu" else _hostsim_par_loop


SIZES = [(nelems, nelems, nelems), (0, nelems, nelems), (nelems // 2, nelems, nelems)]


# ---- test_global_reduction.py ---------------------------------------------------------------------------------------
@pytest.fixture(params=SIZES)
def set_(request):
    return op2.Set(request.param, "set")


@pytest.fixture
def d1(set_):
    return op2.Dat(op2.DataSet(set_, 1), numpy.arange(nelems) + 1, dtype=numpy.uint32)


@pytest.fixture
def d2(set_):
    return op2.Dat(op2.DataSet(set_, 2), numpy.arange(2 * nelems) + 1, dtype=numpy.uint32)


K1_WRITE_TO_DAT = "static void k(unsigned int *x, unsigned int *g) { *x = *g; }"
K1_INC_TO_GLOBAL = "static void k(unsigned int *g, unsigned int *x) { *g += *x; }"
K1_MIN_TO_GLOBAL = "static void k(unsigned int *g, unsigned int *x) { if (*x < *g) *g = *x; }"
K2_MIN_TO_GLOBAL = """static void k(unsigned int *g, unsigned int *x) {
        if (x[0] < g[0]) g[0] = x[0];
        if (x[1] < g[1]) g[1] = x[1];
        }"""
K1_MAX_TO_GLOBAL = "static void k(unsigned int *g, unsigned int *x) { if (*x > *g) *g = *x; }"
K2_MAX_TO_GLOBAL = """static void k(unsigned int *g, unsigned int *x) {
        if (x[0] > g[0]) g[0] = x[0];
        if (x[1] > g[1]) g[1] = x[1];
        }"""
K2_WRITE_TO_DAT = "static void k(unsigned int *x, unsigned int *g) { *x = g[0] + g[1]; }"
K2_INC_TO_GLOBAL = "static void k(unsigned int *g, unsigned int *x) { g[0] += x[0]; g[1] += x[1]; }"


@pytest.mark.parametrize("ctype,dtype,init,val,access,cmp,expect", [
    ("unsigned int", numpy.uint32, 8, 12, op2.MIN, "<", 8),
    ("int", numpy.int32, 8, -12, op2.MIN, "<", -12),
    ("int", numpy.int32, -42, -12, op2.MAX, ">", -12),
    ("float", numpy.float32, -.8, -12.0, op2.MIN, "<", -12.0),
    ("float", numpy.float32, -42.8, -12.0, op2.MAX, ">", -12.0),
    ("double", numpy.float64, -.8, -12.0, op2.MIN, "<", -12.0),
    ("double", numpy.float64, -42.8, -12.0, op2.MAX, ">", -12.0)])
def test_direct_min_max_by_type(par_loop, set_, ctype, dtype, init, val, access, cmp, expect):
    """test_global_reduction.py:154-246 (test_direct_{min,max}_{uint32,int32,float,double})."""
    d = op2.Dat(op2.DataSet(set_, 1), [val] * nelems, dtype)
    code = "static void k(%s* g, %s* x)\n{\n  if ( *x %s *g ) *g = *x;\n}\n" % (ctype, ctype, cmp)
    g = op2.Global(1, init, dtype, "g")
    par_loop(op2.Kernel(code, "k"), set_, g(access), d(op2.READ))
    assert_allclose(g.data[0], expect)


def test_1d_and_2d_read(par_loop, set_, d1):
    g = op2.Global(1, 1, dtype=numpy.uint32)
    par_loop(op2.Kernel(K1_WRITE_TO_DAT, "k"), set_, d1(op2.WRITE), g(op2.READ))
    assert all(d1.data == g.data)
    g = op2.Global(1, dtype=numpy.uint32)                       # test_1d_read_no_init
    d1.data[:] = 100
    par_loop(op2.Kernel(K1_WRITE_TO_DAT, "k"), set_, d1(op2.WRITE), g(op2.READ))
    assert all(g.data == 0) and all(d1.data == 0)
    g = op2.Global(2, (1, 2), dtype=numpy.uint32)               # test_2d_read
    par_loop(op2.Kernel(K2_WRITE_TO_DAT, "k"), set_, d1(op2.WRITE), g(op2.READ))
    assert all(d1.data == g.data.sum())


def test_1d_inc_min_max(par_loop, set_, d1):
    for g in (op2.Global(1, 0, dtype=numpy.uint32), op2.Global(1, dtype=numpy.uint32)):     # test_1d_inc[_no_data]
        par_loop(op2.Kernel(K1_INC_TO_GLOBAL, "k"), set_, g(op2.INC), d1(op2.READ))
        assert g.data == d1.data.sum()
    val = d1.data.min() + 1                                      # test_1d_min_dat_is_min
    g = op2.Global(1, val, dtype=numpy.uint32)
    par_loop(op2.Kernel(K1_MIN_TO_GLOBAL, "k"), set_, g(op2.MIN), d1(op2.READ))
    assert g.data == d1.data.min()
    val = d1.data.max() - 1                                      # test_1d_max_dat_is_max
    g = op2.Global(1, val, dtype=numpy.uint32)
    par_loop(op2.Kernel(K1_MAX_TO_GLOBAL, "k"), set_, g(op2.MAX), d1(op2.READ))
    assert g.data == d1.data.max()
    val = d1.data.max() + 1                                      # test_1d_max_global_is_max
    g = op2.Global(1, val, dtype=numpy.uint32)
    par_loop(op2.Kernel(K1_MAX_TO_GLOBAL, "k"), set_, g(op2.MAX), d1(op2.READ))
    assert g.data == val
    d1.data[:] += 10                                             # test_1d_min_global_is_min
    val = d1.data.min() - 1
    g = op2.Global(1, val, dtype=numpy.uint32)
    par_loop(op2.Kernel(K1_MIN_TO_GLOBAL, "k"), set_, g(op2.MIN), d1(op2.READ))
    assert g.data == val


def test_2d_inc_min_max(par_loop, set_, d2):
    g = op2.Global(2, (0, 0), dtype=numpy.uint32)                # test_2d_inc
    par_loop(op2.Kernel(K2_INC_TO_GLOBAL, "k"), set_, g(op2.INC), d2(op2.READ))
    assert g.data[0] == d2.data[:, 0].sum() and g.data[1] == d2.data[:, 1].sum()
    v = (d2.data[:, 0].min() + 1, d2.data[:, 1].min() + 1)       # test_2d_min_dat_is_min
    g = op2.Global(2, v, dtype=numpy.uint32)
    par_loop(op2.Kernel(K2_MIN_TO_GLOBAL, "k"), set_, g(op2.MIN), d2(op2.READ))
    assert g.data[0] == d2.data[:, 0].min() and g.data[1] == d2.data[:, 1].min()
    v = (d2.data[:, 0].max() - 1, d2.data[:, 1].max() - 1)       # test_2d_max_dat_is_max
    g = op2.Global(2, v, dtype=numpy.uint32)
    par_loop(op2.Kernel(K2_MAX_TO_GLOBAL, "k"), set_, g(op2.MAX), d2(op2.READ))
    assert g.data[0] == d2.data[:, 0].max() and g.data[1] == d2.data[:, 1].max()
    v = (d2.data[:, 0].max() + 1, d2.data[:, 1].max() + 1)       # test_2d_max_global_is_max
    g = op2.Global(2, v, dtype=numpy.uint32)
    par_loop(op2.Kernel(K2_MAX_TO_GLOBAL, "k"), set_, g(op2.MAX), d2(op2.READ))
    assert tuple(g.data) == v
    d2.data[:, 0] += 10                                          # test_2d_min_global_is_min
    d2.data[:, 1] += 10
    v = (d2.data[:, 0].min() - 1, d2.data[:, 1].min() - 1)
    g = op2.Global(2, v, dtype=numpy.uint32)
    par_loop(op2.Kernel(K2_MIN_TO_GLOBAL, "k"), set_, g(op2.MIN), d2(op2.READ))
    assert tuple(g.data) == v


def test_multi_inc_and_mixed_type_globals(par_loop, set_, d1):
    k = op2.Kernel(K1_INC_TO_GLOBAL, "k")
    g = op2.Global(1, 0, dtype=numpy.uint32)                     # test_1d_multi_inc_same_global[_reset]
    par_loop(k, set_, g(op2.INC), d1(op2.READ))
    assert g.data == d1.data.sum()
    par_loop(k, set_, g(op2.INC), d1(op2.READ))
    assert g.data == d1.data.sum() * 2
    g.data = 10
    par_loop(k, set_, g(op2.INC), d1(op2.READ))
    assert g.data == d1.data.sum() + 10
    g2 = op2.Global(1, 10, dtype=numpy.uint32)                   # test_1d_multi_inc_diff_global
    par_loop(k, set_, g2(op2.INC), d1(op2.READ))
    assert g2.data == d1.data.sum() + 10
    g_uint32 = op2.Global(1, [0], numpy.uint32, "g_uint32")      # test_globals_with_different_types
    g_double = op2.Global(1, [0.0], numpy.float64, "g_double")
    kk = "static void k(unsigned int* i, double* d) { *i += 1; *d += 1.0f; }"
    par_loop(op2.Kernel(kk, "k"), set_, g_uint32(op2.INC), g_double(op2.INC))
    assert_allclose(g_uint32.data[0], g_double.data[0])
    assert g_uint32.data[0] == set_.size
    g = op2.Global(1, 0, dtype=numpy.uint32)                     # test_inc_repeated_loop
    kk = "static void k(unsigned int* g) { *g += 1; }"
    par_loop(op2.Kernel(kk, "k"), set_, g(op2.INC))
    assert_allclose(g.data, set_.size)
    par_loop(op2.Kernel(kk, "k"), set_, g(op2.INC))
    assert_allclose(g.data, 2 * set_.size)
    g.zero()
    par_loop(op2.Kernel(kk, "k"), set_, g(op2.INC))
    assert_allclose(g.data, set_.size)


@pytest.mark.gpu
def test_inc_reused_loop(set_):
    """test_global_reduction.py:447-457: one ParLoop object computed three times."""
    g = op2.Global(1, 0, dtype=numpy.uint32)
    loop = op2.ParLoop(op2.Kernel("void k(unsigned int* g) { *g += 1; }", "k"), set_, g(op2.INC))
    loop.compute()
    assert_allclose(g.data, set_.size)
    loop.compute()
    assert_allclose(g.data, 2 * set_.size)
    g.zero()
    loop.compute()
    assert_allclose(g.data, set_.size)


# ---- test_direct_loop.py ---------------------------------------------------------------------------------------------
@pytest.fixture(params=SIZES + [(0, nelems // 2, nelems)])
def elems(request):
    return op2.Set(request.param, "elems")


def xarray():
    return np.array(range(nelems), dtype=np.uint32)


@pytest.fixture
def x(elems):
    return op2.Dat(op2.DataSet(elems, 1), xarray(), np.uint32, "x")


@pytest.fixture
def y(elems):
    return op2.Dat(op2.DataSet(elems, 2), [xarray(), xarray()], np.uint32, "x")


def test_direct_wo_rw_and_mismatch(par_loop, elems, x):
    kernel_wo = "static void wo(unsigned int* x) { *x = 42; }"
    with pytest.raises(op2.MapValueError):                       # test_mismatch_set_raises_error
        op2.LegacyParloop(op2.Kernel(kernel_wo, "wo"), op2.Set(elems.size), x(op2.WRITE))
    kernel_rw = "static void wo(unsigned int* x) { (*x) = (*x) + 1; }"          # test_rw
    par_loop(op2.Kernel(kernel_rw, "wo"), elems, x(op2.RW))
    _nelems = elems.size
    assert sum(x.data_ro) == _nelems * (_nelems + 1) // 2
    if _nelems == nelems:
        assert sum(x.data_ro_with_halos) == nelems * (nelems + 1) // 2
    par_loop(op2.Kernel(kernel_wo, "wo"), elems, x(op2.WRITE))   # test_wo
    assert all(map(lambda v: v == 42, x.data))


def test_direct_global_inc_max_min_read(par_loop, elems, x):
    g = op2.Global(1, 0, np.uint32, "g")
    kernel_global_inc = """static void global_inc(unsigned int* x, unsigned int* inc) {
          (*x) = (*x) + 1; (*inc) += (*x);
        }"""
    par_loop(op2.Kernel(kernel_global_inc, "global_inc"), elems, x(op2.RW), g(op2.INC))
    _nelems = elems.size
    assert g.data[0] == _nelems * (_nelems + 1) // 2
    g = op2.Global(1, 0, np.uint32, "g")                         # test_global_inc_init_not_zero
    g.data[0] = 10
    par_loop(op2.Kernel("static void k(unsigned int* inc) { (*inc) += 1; }", "k"), elems, g(op2.INC))
    assert g.data[0] == elems.size + 10
    x.data[:] = xarray()[:elems.size]
    g = op2.Global(1, 0, np.uint32, "g")                         # test_global_max_dat_is_max
    par_loop(op2.Kernel("static void k(unsigned int *g, unsigned int *x) {\n if ( *g < *x ) { *g = *x; }\n}", "k"),
             elems, g(op2.MAX), x(op2.READ))
    assert g.data[0] == x.data.max()
    g.data[0] = nelems * 2                                       # test_global_max_g_is_max (argument order swapped)
    par_loop(op2.Kernel("static void k(unsigned int *x, unsigned int *g) {\n if ( *g < *x ) { *g = *x; }\n}", "k"),
             elems, x(op2.READ), g(op2.MAX))
    assert g.data[0] == nelems * 2
    g.data[0] = 1000                                             # test_global_min_dat_is_min
    par_loop(op2.Kernel("static void k(unsigned int *g, unsigned int *x) {\n if ( *g > *x ) { *g = *x; }\n}", "k"),
             elems, g(op2.MIN), x(op2.READ))
    assert g.data[0] == x.data.min()
    g.data[0] = 10                                               # test_global_min_g_is_min
    x.data[:] = 11
    par_loop(op2.Kernel("static void k(unsigned int *x, unsigned int *g) {\n if ( *g > *x ) { *g = *x; }\n}", "k"),
             elems, x(op2.READ), g(op2.MIN))
    assert g.data[0] == 10
    x.data[:] = xarray()[:elems.size]                            # test_global_read
    h = op2.Global(1, 1, np.uint32, "h")
    par_loop(op2.Kernel("static void global_read(unsigned int* x, unsigned int* h) {\n (*x) += (*h);\n}", "global_read"),
             elems, x(op2.RW), h(op2.READ))
    assert sum(x.data_ro) == _nelems * (_nelems + 1) // 2


def test_direct_2d_host_write_zero(par_loop, elems, x, y):
    par_loop(op2.Kernel("static void k2d_wo(unsigned int* x) {\n x[0] = 42; x[1] = 43;\n}", "k2d_wo"), elems, y(op2.WRITE))
    assert all(map(lambda v: all(v == [42, 43]), y.data))
    g = op2.Global(1, 0, np.uint32, "g")                         # test_host_write
    x.data[:] = 1
    g.data[:] = 0
    par_loop(op2.Kernel("static void k(unsigned int *g, unsigned int *x) { *g += *x; }", "k"), elems, g(op2.INC), x(op2.READ))
    _nelems = elems.size
    assert g.data[0] == _nelems
    x.data[:] = 2
    g.data[:] = 0
    par_loop(op2.Kernel("static void k(unsigned int *x, unsigned int *g) { *g += *x; }", "k"), elems, x(op2.READ), g(op2.INC))
    assert g.data[0] == 2 * _nelems
    x.data[:] = 10                                               # test_zero_1d_dat / test_zero_2d_dat
    x.zero()
    assert (x.data == 0).all()
    y.data[:] = 10
    y.zero()
    assert (y.data == 0).all()


@pytest.mark.gpu
def test_kernel_cplusplus():
    """test_direct_loop.py:238-250: a C++ local kernel (cpp=True)."""
    s = op2.Set(nelems)
    yv = op2.Dat(s, dtype=np.float64)
    yv.data[:] = -10.5
    k = op2.Kernel("""
        #include <cmath>
        static void k(double *y)
        {
            *y = std::abs(*y);
        }
        """, "k", cpp=True)
    op2.par_loop(k, s, yv(op2.RW))
    assert (yv.data == 10.5).all()


# ---- test_vector_map.py / test_iteration_space_dats.py --------------------------------------------------------------
@pytest.fixture
def node():
    return op2.Set(nnodes, "node")


@pytest.fixture
def ele():
    return op2.Set(nele, "ele")


@pytest.fixture
def node2ele(node, ele):
    return op2.Map(node, ele, 1, numpy.arange(nnodes) / 2, "node2ele")


@pytest.mark.parametrize("kernel_sum,access", [
    ("static void sum(unsigned int* edge, unsigned int *nodes) {\n *edge = nodes[0] + nodes[1];\n}", op2.WRITE),
    ("static void sum(unsigned int *edge, unsigned int *nodes) {\n  for (int i=0; i<2; ++i)\n    edge[0] += nodes[i];\n}", op2.INC)])
def test_sum_nodes_to_edges(par_loop, kernel_sum, access):
    nedges = nnodes - 1
    nodes = op2.Set(nnodes, "nodes")
    edges = op2.Set(nedges, "edges")
    node_vals = op2.Dat(nodes, numpy.arange(nnodes, dtype=numpy.uint32), numpy.uint32, "node_vals")
    edge_vals = op2.Dat(edges, numpy.zeros(nedges, dtype=numpy.uint32), numpy.uint32, "edge_vals")
    e_map = numpy.array([(i, i + 1) for i in range(nedges)], dtype=numpy.uint32)
    edge2node = op2.Map(edges, nodes, 2, e_map, "edge2node")
    par_loop(op2.Kernel(kernel_sum, "sum"), edges, edge_vals(access), node_vals(op2.READ, edge2node))
    assert all(numpy.arange(1, nedges * 2 + 1, 2) == edge_vals.data)


def test_1d_vector_map_read_write_inc(par_loop, node, ele, node2ele):
    d1 = op2.Dat(node, numpy.zeros(nnodes), dtype=numpy.int32)
    vd1 = op2.Dat(ele, numpy.zeros(nele), dtype=numpy.int32)
    vd1.data[:] = numpy.arange(nele)
    par_loop(op2.Kernel("static void k(int *d, int *vd) {\n *d = vd[0];\n}", "k"), node, d1(op2.WRITE), vd1(op2.READ, node2ele))
    assert all(d1.data[::2] == vd1.data) and all(d1.data[1::2] == vd1.data)
    par_loop(op2.Kernel("static void k(int *vd) {\n vd[0] = 2;\n}", "k"), node, vd1(op2.WRITE, node2ele))
    assert all(vd1.data == 2)
    vd1.data[:] = 3
    d1.data[:] = numpy.arange(nnodes).reshape(d1.data.shape)
    par_loop(op2.Kernel("static void k(int *vd, int *d) {\n vd[0] += *d;\n}", "k"), node, vd1(op2.INC, node2ele), d1(op2.READ))
    expected = numpy.zeros_like(vd1.data)
    expected[:] = 3
    expected += numpy.arange(start=0, stop=nnodes, step=2).reshape(expected.shape)
    expected += numpy.arange(start=1, stop=nnodes, step=2).reshape(expected.shape)
    assert all(vd1.data == expected)


def test_2d_itspace_map_read_write_inc(par_loop, node, ele, node2ele):
    d2 = op2.Dat(node ** 2, numpy.zeros(2 * nnodes), dtype=numpy.int32)
    vd2 = op2.Dat(ele ** 2, numpy.zeros(2 * nele), dtype=numpy.int32)
    vd2.data[:] = numpy.arange(nele * 2).reshape(nele, 2)
    k = "static void k(int *d, int *vd) {\n  for (int i=0; i<1; ++i) {\n    d[0] = vd[i];\n    d[1] = vd[i+1];\n  }\n}"
    par_loop(op2.Kernel(k, "k"), node, d2(op2.WRITE), vd2(op2.READ, node2ele))
    assert all(d2.data[::2, 0] == vd2.data[:, 0]) and all(d2.data[::2, 1] == vd2.data[:, 1])
    assert all(d2.data[1::2, 0] == vd2.data[:, 0]) and all(d2.data[1::2, 1] == vd2.data[:, 1])
    k = "static void k(int *vd) {\n  for (int i=0; i<1; ++i) {\n    vd[i] = 2;\n    vd[i+1] = 3;\n  }\n}"
    par_loop(op2.Kernel(k, "k"), node, vd2(op2.WRITE, node2ele))
    assert all(vd2.data[:, 0] == 2) and all(vd2.data[:, 1] == 3)
    vd2.data[:, 0] = 3
    vd2.data[:, 1] = 4
    d2.data[:] = numpy.arange(2 * nnodes).reshape(d2.data.shape)
    k = "static void k(int *vd, int *d) {\n  for (int i=0; i<1; ++i) {\n    vd[i] += d[0];\n    vd[i+1] += d[1];\n  }\n}"
    par_loop(op2.Kernel(k, "k"), node, vd2(op2.INC, node2ele), d2(op2.READ))
    expected = numpy.zeros_like(vd2.data)
    expected[:, 0] = 3
    expected[:, 1] = 4
    expected[:, 0] += numpy.arange(start=0, stop=2 * nnodes, step=4)
    expected[:, 0] += numpy.arange(start=2, stop=2 * nnodes, step=4)
    expected[:, 1] += numpy.arange(start=1, stop=2 * nnodes, step=4)
    expected[:, 1] += numpy.arange(start=3, stop=2 * nnodes, step=4)
    assert all(vd2.data[:, 0] == expected[:, 0]) and all(vd2.data[:, 1] == expected[:, 1])
