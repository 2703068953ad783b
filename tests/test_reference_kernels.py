"""a9 parity: the reference's OWN kernel text through this backend's wrappers.

Every local kernel below is the C string the reference's PyOP2 tests pass to ``op2.Kernel`` -- lifted unchanged into
tests/golden/reference_kernels.json and verified against /root/reference by tests/golden/make_reference_kernels.py -- and every
test re-enacts the reference test that uses it: same sets, maps, data, access descriptors, expected values and tolerances
(tests/pyop2/test_matrices.py:637-757, 870-945; test_extrusion.py:344-450; test_indirect_loop.py:134-470; test_subset.py:139-254;
test_direct_loop.py:88-211; test_global_reduction.py:158-258; test_vector_map.py:108-180; test_iteration_space_dats.py:98-225).
Each test runs through the HIP wrappers on the GPU (``-m gpu``) and, without a GPU, through the generated wrapper compiled for
the host (tests/refkernels.py).  Whatever ``fd_wrapper.h`` must provide for this text to compile as a ``__device__`` function --
``PetscScalar``, C99 ``restrict``, non-const READ arguments, block-scope tables, ``unsigned`` -- is the a9 contract.
"""
import json
import os

import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import op2
from refkernels import GpuBackend, HostBackend, ref_kernel

G = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "pyop2_matrices.json")))
M = "test_matrices.py::"
nelems = 4096


@pytest.fixture(params=[pytest.param("gpu", marks=pytest.mark.gpu), pytest.param("gpu-direct", marks=pytest.mark.gpu), "host"])
def be(request, monkeypatch):
    """gpu: the wrapper shape the backend picks (staged / owner-computes-rows / direct); gpu-direct: the direct wrapper forced"""
    if request.param == "gpu-direct":
        from firedrake_amd.configuration import configuration
        monkeypatch.setitem(configuration, "mode", "direct")
    return GpuBackend() if request.param.startswith("gpu") else HostBackend()


# ---------------------------------------------------------------------------------------- test_matrices.py
@pytest.fixture
def tri():
    nodes, elements = op2.Set(4, "nodes"), op2.Set(2, "elements")
    elem_node = op2.Map(elements, nodes, 3, np.asarray(G["elem_node_map"], dtype=np.uint32), "elem_node")
    coords = op2.Dat(nodes ** 2, np.asarray(G["coords"], dtype=np.float64), np.float64, "coords")
    f = op2.Dat(nodes ** 1, np.asarray(G["f"], dtype=np.float64), np.float64, "f")
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(elem_node, elem_node, None)], name="sparsity"), np.float64, "mat")
    return nodes, elements, elem_node, coords, f, mat


def _mass_mat(be, tri, key=M + "mass::mass"):
    nodes, elements, elem_node, coords, f, mat = tri
    be.zero(mat)
    be.par_loop(ref_kernel(key), elements, mat(op2.INC, (elem_node, elem_node)), coords(op2.READ, elem_node))
    return mat


@pytest.mark.parametrize("key", [M + "mass::mass", M + "mass_ffc::mass_ffc"])
def test_assemble_mat(be, tri, key):
    """test_matrices.py:637-646 (mass) and :699-707 (mass_ffc): eps = 1e-5 on the 4x4 golden matrix"""
    mat = _mass_mat(be, tri, key)
    assert_allclose(be.values(mat), np.asarray(G["expected_matrix"]), 1.e-5)


def test_assemble_rhs(be, tri):
    """test_matrices.py:648-658: eps = 1e-12"""
    nodes, elements, elem_node, coords, f, mat = tri
    b = op2.Dat(nodes ** 1, np.zeros(4), np.float64, "b")
    b.zero()
    be.par_loop(ref_kernel(M + "rhs::rhs"), elements, b(op2.INC, elem_node), coords(op2.READ, elem_node), f(op2.READ, elem_node))
    assert_allclose(b.data, np.asarray(G["expected_rhs"]), 1.e-12)


def test_solve(be, tri):
    """test_matrices.py:660-665: the solution of M x = b is f, eps = 1e-8"""
    nodes, elements, elem_node, coords, f, mat = tri
    mat = _mass_mat(be, tri)
    b = op2.Dat(nodes ** 1, np.zeros(4), np.float64, "b")
    be.par_loop(ref_kernel(M + "rhs::rhs"), elements, b(op2.INC, elem_node), coords(op2.READ, elem_node), f(op2.READ, elem_node))
    assert_allclose(np.linalg.solve(be.values(mat), b.data), f.data, 1.e-8)


def test_set_matrix(be, tri):
    """test_matrices.py:674-691: INC ones, then WRITE them back"""
    nodes, elements, elem_node, coords, f, mat = tri
    g = op2.Global(1, 1.0, np.float64, "g")
    be.zero(mat)
    be.par_loop(ref_kernel(M + "kernel_inc::inc"), elements, mat(op2.INC, (elem_node, elem_node)), g(op2.READ))
    assert be.values(mat).sum() == 3 * 3 * elements.size
    be.par_loop(ref_kernel(M + "kernel_set::set"), elements, mat(op2.WRITE, (elem_node, elem_node)), g(op2.READ))
    assert be.values(mat).sum() == (3 * 3 - 2) * elements.size


def test_zero_rhs_and_rhs_ffc(be, tri):
    """test_matrices.py:693-697 (zero_dat), :709-719 (rhs_ffc, eps 1e-6), :721-736 (rhs_ffc_itspace after zero_dat)"""
    nodes, elements, elem_node, coords, f, mat = tri
    b = op2.Dat(nodes ** 1, np.full(4, 7.0), np.float64, "b")
    be.par_loop(ref_kernel(M + "zero_dat::zero_dat"), nodes, b(op2.WRITE))
    assert all(b.data == np.zeros_like(b.data))
    be.par_loop(ref_kernel(M + "rhs_ffc::rhs_ffc"), elements, b(op2.INC, elem_node), coords(op2.READ, elem_node), f(op2.READ, elem_node))
    assert_allclose(b.data, np.asarray(G["expected_rhs"]), 1.e-6)
    be.par_loop(ref_kernel(M + "zero_dat::zero_dat"), nodes, b(op2.WRITE))
    be.par_loop(ref_kernel(M + "rhs_ffc_itspace::rhs_ffc_itspace"), elements, b(op2.INC, elem_node), coords(op2.READ, elem_node),
                f(op2.READ, elem_node))
    assert_allclose(b.data, np.asarray(G["expected_rhs"]), 1.e-6)
    bv = op2.Dat(nodes ** 2, np.full((4, 2), 3.0), np.float64, "bv")
    be.par_loop(ref_kernel(M + "zero_vec_dat::zero_vec_dat"), nodes, bv(op2.WRITE))
    assert (bv.data == 0).all()


def test_zero_rows(be, tri):
    """test_matrices.py:738-766: zeroed rows carry the given diagonal"""
    nodes, elements, elem_node, coords, f, mat = tri
    mat = _mass_mat(be, tri)
    E = np.asarray(G["expected_matrix"]).copy()
    E[0] = [12.0, 0.0, 0.0, 0.0]
    be.zero_rows(mat, [0], 12.0)
    assert_allclose(be.values(mat), E, 1.e-5)
    E[3] = [0.0, 0.0, 0.0, 4.0]
    be.zero_rows(mat, op2.Subset(nodes, [3]), 4.0)
    assert_allclose(be.values(mat), E, 1.e-5)


def test_minimal_zero_mat(be):
    """test_matrices.py:617-635"""
    n = 128
    s = op2.Set(n)
    m = op2.Map(s, s, 1, np.array(list(range(n)), np.uint32))
    mat = op2.Mat(op2.Sparsity((s ** 1, s ** 1), [(m, m, None)]), np.float64)
    be.par_loop(ref_kernel(M + "TestMatrices.test_minimal_zero_mat::zero_mat"), s, mat(op2.WRITE, (m, m)))
    assert_allclose(be.values(mat), np.zeros((n, n)), 1.e-12)


@pytest.fixture
def mixed():
    mset = op2.MixedSet((op2.Set(3), op2.Set(4)))
    elem, node = mset
    mmap = op2.MixedMap((op2.Map(elem, elem, 1, [0, 1, 2]), op2.Map(elem, node, 2, [0, 1, 1, 2, 2, 3])))
    rdata = lambda s: np.arange(1, s + 1, dtype=np.float64)          # noqa: E731
    mdat = op2.MixedDat(op2.Dat(s, rdata(s.size)) for s in mset)
    mvdat = op2.MixedDat(op2.Dat(s ** 2, list(zip(rdata(s.size), rdata(s.size)))) for s in mset)
    return mset, mmap, mdat, mvdat, rdata


def test_assemble_mixed_mat_and_rhs(be, mixed):
    """test_matrices.py:870-945 (TestMixedMatrices): the 3x3 kernels split over the 2x2 blocks of a mixed space"""
    mset, mmap, mdat, mvdat, rdata = mixed
    sp = op2.Sparsity((mset ** 1, mset ** 1), {(i, j): [(rm, cm, None)] for i, rm in enumerate(mmap) for j, cm in enumerate(mmap)})
    mat = op2.Mat(sp)
    be.par_loop(ref_kernel(M + "TestMixedMatrices.mat::addone_mat"), mmap.iterset, mat(op2.INC, (mmap, mmap)), mdat(op2.READ, mmap))
    od = np.array([[1.0, 2.0, 0.0, 0.0], [0.0, 4.0, 6.0, 0.0], [0.0, 0.0, 9.0, 12.0]])
    ll = np.diag([1.0, 8.0, 18.0, 16.0]) + np.diag([2.0, 6.0, 12.0], -1) + np.diag([2.0, 6.0, 12.0], 1)
    eps = 1.e-12
    assert_allclose(be.values(mat[0, 0]), np.diag([1.0, 4.0, 9.0]), eps)
    assert_allclose(be.values(mat[0, 1]), od, eps)
    assert_allclose(be.values(mat[1, 0]), od.T, eps)
    assert_allclose(be.values(mat[1, 1]), ll, eps)
    dat = op2.MixedDat(mset)
    be.par_loop(ref_kernel(M + "TestMixedMatrices.dat::addone_rhs"), mmap.iterset, dat(op2.INC, mmap), mdat(op2.READ, mmap))
    assert_allclose(dat[0].data_ro, rdata(3), eps)
    assert_allclose(dat[1].data_ro, [1.0, 4.0, 6.0, 4.0], eps)
    vdat = op2.MixedDat(mset ** 2)
    be.par_loop(ref_kernel(M + "TestMixedMatrices.test_assemble_mixed_rhs_vector::addone_rhs_vec"), mmap.iterset,
                vdat(op2.INC, mmap), mvdat(op2.READ, mmap))
    assert_allclose(vdat[0].data_ro, np.kron(list(zip(rdata(3))), np.ones(2)), eps)
    assert_allclose(vdat[1].data_ro, np.kron(list(zip([1.0, 4.0, 6.0, 4.0])), np.ones(2)), eps)


# ---------------------------------------------------------------------------------------- test_indirect_loop.py
I = "test_indirect_loop.py::TestIndirectLoop."


@pytest.fixture(params=[(nelems, nelems, nelems), (nelems // 2, nelems, nelems)])
def ind(request):
    iterset, indset, unitset = op2.Set(request.param, "iterset"), op2.Set(nelems, "indset"), op2.Set(1, "unitset")
    mapd = list(range(nelems))[::-1]
    x = op2.Dat(indset, list(range(nelems)), np.uint32, "x")
    x2 = op2.Dat(indset ** 2, np.array([list(range(nelems)), list(range(nelems))], dtype=np.uint32), np.uint32, "x2")
    i2i = op2.Map(iterset, indset, 1, np.array(mapd, dtype=np.uint32), "iterset2indset")
    i2u = op2.Map(iterset, unitset, 1, np.zeros(nelems, dtype=np.uint32), "iterset2unitset")
    return iterset, indset, unitset, x, x2, i2i, i2u


def test_indirect_wo_rw_inc_max_min(be, ind):
    """test_indirect_loop.py:134-178 (on a set whose owned part is its whole or its first half)"""
    iterset, indset, unitset, x, x2, i2i, i2u = ind
    n = iterset.size
    touched = np.asarray(i2i.values_with_halo[:n, 0])
    be.par_loop(ref_kernel(I + "test_onecolor_wo::kernel_wo"), iterset, x(op2.WRITE, i2i))
    assert all(x.data[touched] == 42)
    x = op2.Dat(indset, list(range(nelems)), np.uint32, "x")
    be.par_loop(ref_kernel(I + "test_onecolor_rw::rw"), iterset, x(op2.RW, i2i))
    assert sum(x.data[touched]) == sum(int(v) + 1 for v in touched)
    u = op2.Dat(unitset, np.array([0], dtype=np.uint32), np.uint32, "u")
    be.par_loop(ref_kernel(I + "test_indirect_inc::inc"), iterset, u(op2.INC, i2u))
    assert u.data[0] == n
    a, b = op2.Dat(indset, dtype=np.int32), op2.Dat(indset, dtype=np.int32)
    a.data[:] = -10
    b.data[:] = -5
    be.par_loop(ref_kernel(I + "test_indirect_max::maxify"), iterset, a(op2.MAX, i2i), b(op2.READ, i2i))
    assert np.allclose(a.data_ro[touched], -5)
    a.data[:] = 10
    b.data[:] = 5
    be.par_loop(ref_kernel(I + "test_indirect_min::minify"), iterset, a(op2.MIN, i2i), b(op2.READ, i2i))
    assert np.allclose(a.data_ro[touched], 5)


def test_indirect_globals_and_2d(be, ind):
    """test_indirect_loop.py:180-236"""
    iterset, indset, unitset, x, x2, i2i, i2u = ind
    n = iterset.size
    touched = np.asarray(i2i.values_with_halo[:n, 0])
    g = op2.Global(1, 2, np.uint32, "g")
    be.par_loop(ref_kernel(I + "test_global_read::global_read"), iterset, x(op2.RW, i2i), g(op2.READ))
    assert sum(x.data[touched]) == sum(int(v) // 2 for v in touched)
    x = op2.Dat(indset, list(range(nelems)), np.uint32, "x")
    g = op2.Global(1, 0, np.uint32, "g")
    be.par_loop(ref_kernel(I + "test_global_inc::global_inc"), iterset, x(op2.RW, i2i), g(op2.INC))
    assert sum(x.data[touched]) == sum(int(v) + 1 for v in touched)
    assert g.data[0] == sum(int(v) + 1 for v in touched)
    be.par_loop(ref_kernel(I + "test_2d_dat::wo"), iterset, x2(op2.WRITE, i2i))
    assert all(all(v == [42, 43]) for v in x2.data[touched])
    nedges = nelems - 1
    nodes, edges = op2.Set(nelems, "nodes"), op2.Set(nedges, "edges")
    node_vals = op2.Dat(nodes, np.arange(nelems, dtype=np.uint32), np.uint32, "node_vals")
    edge_vals = op2.Dat(edges, np.zeros(nedges, dtype=np.uint32), np.uint32, "edge_vals")
    edge2node = op2.Map(edges, nodes, 2, np.array([(i, i + 1) for i in range(nedges)], dtype=np.uint32), "edge2node")
    be.par_loop(ref_kernel(I + "test_2d_map::sum"), edges, edge_vals(op2.WRITE), node_vals(op2.READ, edge2node))
    assert all(np.arange(1, nedges * 2 + 1, 2) == edge_vals.data)


@pytest.mark.parametrize("key", ["test_mixed_non_mixed_dat::inc", "test_mixed_non_mixed_dat_itspace::inc"])
def test_mixed_indirect(be, key):
    """test_indirect_loop.py:256-280"""
    iterset, indset, unitset = op2.Set(nelems, "iterset"), op2.Set(nelems, "indset"), op2.Set(1, "unitset")
    mdat = op2.MixedDat(op2.MixedSet((indset, unitset)))
    mmap = op2.MixedMap((op2.Map(iterset, indset, 1, np.array(list(range(nelems))[::-1], dtype=np.uint32)),
                         op2.Map(iterset, unitset, 1, np.zeros(nelems, dtype=np.uint32))))
    d = op2.Dat(iterset, np.ones(iterset.size), dtype=np.float64)
    be.par_loop(ref_kernel("test_indirect_loop.py::TestMixedIndirectLoop." + key), iterset, mdat(op2.INC, mmap), d(op2.READ))
    assert all(mdat[0].data == 1.0) and mdat[1].data == 4096.0


def test_permuted_maps(be):
    """test_indirect_loop.py:283-318"""
    fromset, toset = op2.Set(1), op2.Set(4)
    for key, m1v, p2, p3 in (("test_permuted_map::copy", [1, 2, 3, 0], [3, 2, 0, 1], None),
                             ("test_permuted_map_both::copy", [0, 2, 1, 3], [3, 2, 1, 0], [0, 2, 3, 1])):
        d1, d2 = op2.Dat(op2.DataSet(toset, 1), dtype=np.int32), op2.Dat(op2.DataSet(toset, 1), dtype=np.int32)
        d1.data[:] = np.arange(4, dtype=np.int32)
        m1 = op2.Map(fromset, toset, 4, values=m1v)
        m2 = op2.PermutedMap(m1, p2)
        m3 = m1 if p3 is None else op2.PermutedMap(m1, p3)
        be.par_loop(ref_kernel("test_indirect_loop.py::" + key), fromset, d2(op2.WRITE, m2), d1(op2.READ, m3))
        expect = np.empty_like(d1.data)
        src = m1.values if p3 is None else m1.values[..., m3.permutation]
        expect[m1.values[..., m2.permutation]] = d1.data[src]
        assert (d1.data == np.arange(4, dtype=np.int32)).all()
        assert (d2.data == expect).all()


@pytest.mark.parametrize("permuted", ["none", "pre"])
def test_composed_map_two_maps(be, permuted):
    """test_indirect_loop.py:321-345"""
    setB, nodesetB = op2.Set(3), op2.Set(6)
    datB = op2.Dat(op2.DataSet(nodesetB, 1), dtype=np.float64)
    mapB = op2.Map(setB, nodesetB, 2, values=[[0, 1], [2, 3], [4, 5]])
    setA, nodesetA = op2.Set(5), op2.Set(8)
    datA = op2.Dat(op2.DataSet(nodesetA, 1), dtype=np.float64)
    datA.data[:] = np.array([.0, .1, .2, .3, .4, .5, .6, .7], dtype=np.float64)
    mapA0 = op2.Map(setA, nodesetA, 2, values=[[0, 1], [2, 3], [4, 5], [6, 7], [0, 1]])
    if permuted == "pre":
        mapA0 = op2.PermutedMap(mapA0, [1, 0])
    mapA = op2.ComposedMap(mapA0, op2.Map(setB, setA, 1, values=[3, 1, 2]))
    be.par_loop(ref_kernel("test_indirect_loop.py::test_composed_map_two_maps::copy"), setB, datB(op2.WRITE, mapB), datA(op2.READ, mapA))
    exp = [.6, .7, .2, .3, .4, .5] if permuted == "none" else [.7, .6, .3, .2, .5, .4]
    assert (datB.data == np.array(exp, dtype=np.float64)).all()


@pytest.mark.parametrize("nested", ["none", "first", "last"])
@pytest.mark.parametrize("subset", [False, True])
def test_composed_map_three_maps(be, nested, subset):
    """test_indirect_loop.py:348-381"""
    setC, nodesetC = op2.Set(2), op2.Set(4)
    datC = op2.Dat(op2.DataSet(nodesetC, 1), dtype=np.float64)
    mapC = op2.Map(setC, nodesetC, 2, values=[[0, 1], [2, 3]])
    setB, setA, nodesetA = op2.Set(3), op2.Set(5), op2.Set(8)
    datA = op2.Dat(op2.DataSet(nodesetA, 1), dtype=np.float64)
    datA.data[:] = np.array([.0, .1, .2, .3, .4, .5, .6, .7], dtype=np.float64)
    mapA0 = op2.Map(setA, nodesetA, 2, values=[[0, 1], [2, 3], [4, 5], [6, 7], [0, 1]])
    mapA1, mapA2 = op2.Map(setB, setA, 1, values=[3, 1, 2]), op2.Map(setC, setB, 1, values=[2, 0])
    mapA = {"none": lambda: op2.ComposedMap(mapA0, mapA1, mapA2), "first": lambda: op2.ComposedMap(op2.ComposedMap(mapA0, mapA1), mapA2),
            "last": lambda: op2.ComposedMap(mapA0, op2.ComposedMap(mapA1, mapA2))}[nested]()
    it = op2.Subset(setC, np.array([1], dtype=np.int32)) if subset else setC
    be.par_loop(ref_kernel("test_indirect_loop.py::test_composed_map_three_maps::copy"), it, datC(op2.WRITE, mapC), datA(op2.READ, mapA))
    assert (datC.data == np.array([.0, .0, .6, .7] if subset else [.4, .5, .6, .7], dtype=np.float64)).all()


# ---------------------------------------------------------------------------------------- test_subset.py
S = "test_subset.py::TestSubSet."


def test_subset_loops(be):
    """test_subset.py:139-208"""
    iterset = op2.Set(nelems, "iterset")
    ss = op2.Subset(iterset, np.array([i for i in range(nelems) if not i % 2], dtype=np.int32))
    indset = op2.Set(2, "indset")
    m = op2.Map(iterset, indset, 1, [(1 if i % 2 else 0) for i in range(nelems)])
    d = op2.Dat(indset ** 1, data=None, dtype=np.uint32)
    be.par_loop(ref_kernel(S + "test_indirect_loop::inc"), ss, d(op2.INC, m))
    assert d.data[0] == nelems // 2
    values = [2976579765] * nelems
    values[::2] = [i // 2 for i in range(nelems)][::2]
    dat1 = op2.Dat(iterset ** 1, data=values, dtype=np.uint32)
    dat2 = op2.Dat(indset ** 1, data=None, dtype=np.uint32)
    be.par_loop(ref_kernel(S + "test_indirect_loop_with_direct_dat::inc"), ss, dat2(op2.INC, m), dat1(op2.READ))
    assert dat2.data[0] == sum(values[::2])
    even = op2.Subset(iterset, np.array([i for i in range(nelems) if not i % 2], dtype=np.int32))
    odd = op2.Subset(iterset, np.array([i for i in range(nelems) if i % 2], dtype=np.int32))
    indset = op2.Set(nelems, "indset")
    m = op2.Map(iterset, indset, 1, [i for i in range(nelems)])
    dat1, dat2 = op2.Dat(iterset ** 1, data=None, dtype=np.uint32), op2.Dat(indset ** 1, data=None, dtype=np.uint32)
    k = ref_kernel(S + "test_complementary_subsets::inc")
    be.par_loop(k, even, dat1(op2.RW), dat2(op2.INC, m))
    be.par_loop(k, odd, dat1(op2.RW), dat2(op2.INC, m))
    assert np.sum(dat1.data) == nelems and np.sum(dat2.data) == nelems


def test_subset_matrix(be):
    """test_subset.py:210-254: a matrix assembled over a set equals the one assembled over its permuted subsets"""
    iterset, idset, indset = op2.Set(2), op2.Set(2), op2.Set(4)
    dat = op2.Dat(idset ** 1, data=[0, 1], dtype=np.float64)
    m = op2.Map(iterset, indset, 4, [0, 1, 2, 3, 0, 1, 2, 3])
    idmap = op2.Map(iterset, idset, 1, [0, 1])
    sp = op2.Sparsity((indset ** 1, indset ** 1), {(0, 0): [(m, m, None)]})
    k = ref_kernel(S + "test_matrix::unique_id")
    vals = []
    for it in (iterset, op2.Subset(iterset, [0, 1]), op2.Subset(iterset, [1, 0])):
        mat = op2.Mat(sp, np.float64)
        be.zero(mat)
        be.par_loop(k, it, mat(op2.INC, (m, m)), dat(op2.READ, idmap))
        vals.append(be.values(mat))
    assert (vals[1] == vals[0]).all() and (vals[2] == vals[0]).all() and vals[0].sum() > 0


# ---------------------------------------------------------------------------------------- test_extrusion.py
X = "test_extrusion.py::TestExtrusion."


@pytest.fixture
def strip():
    """the extruded strip of test_extrusion.py:76-200: 32 triangles in a row, 10 cell layers, coordinates at the vertices
    (2 values per node, 11 nodes per column), one field value per cell"""
    ex = G["extrusion"]
    nel, layers = ex["nelems"], ex["layers"]
    nnodes = nel + 2
    wedges = layers - 1
    elems2nodes = np.array([[i, i + 1, i + 2] for i in range(nel)], dtype=np.int32)
    # coordinate dofs: node n, layer l -> n*layers + l; the map holds (bottom, top) pairs per vertex with offset 1
    cmap = np.array([[n * layers + d for n in row for d in (0, 1)] for row in elems2nodes], dtype=np.int32)
    fmap = np.array([[i * wedges] for i in range(nel)], dtype=np.int32)
    base = op2.Set(nel, "base")
    elements = op2.ExtrudedSet(base, layers)
    node_set, cell_set = op2.Set(nnodes * layers, "nodes1"), op2.Set(nel * wedges, "elems1")
    coords_map = op2.Map(elements, node_set, 6, cmap, "elem_dofs", np.ones(6, dtype=np.int32))
    field_map = op2.Map(elements, cell_set, 1, fmap, "elem_elem", np.ones(1, dtype=np.int32))
    # the reference's coordinates (test_extrusion.py:168-185): vertex i of the strip at (i // 2, i % 2) -- every triangle has
    # area 1/2 --, repeated over the layers
    xy = np.array([(float(i // 2), float(i % 2)) for i in range(nnodes)])
    coords = np.repeat(xy, layers, axis=0)
    dat_coords = op2.Dat(node_set ** 2, coords, np.float64, "coords")
    dat_field = op2.Dat(cell_set ** 1, np.ones(nel * wedges), np.float64, "field")
    return nel, layers, wedges, nnodes, base, elements, node_set, cell_set, coords_map, field_map, dat_coords, dat_field


def test_extrusion_volume(be, strip):
    """test_extrusion.py:344-361: the volume of the extruded strip through the reference's comp_vol"""
    nel, layers, wedges, nnodes, base, elements, node_set, cell_set, coords_map, field_map, dat_coords, dat_field = strip
    g = op2.Global(1, data=0.0, name="g")
    be.par_loop(ref_kernel(X + "test_extrusion::comp_vol"), elements, g(op2.INC), dat_coords(op2.READ, coords_map),
                dat_field(op2.READ, field_map))
    assert int(g.data[0]) == int((layers - 1) * 0.1 * (nel // 2))
    assert abs(g.data[0] - (layers - 1) * 0.1 * (nel // 2)) < 1e-12 * nel


def test_extruded_direct_inc_layer_arg_and_writes(be, strip):
    """test_extrusion.py:367-450"""
    nel, layers, wedges, nnodes, base, elements, node_set, cell_set, coords_map, field_map, dat_coords, dat_field = strip
    iterset = op2.Set(nelems, "iterset")
    dat = op2.Dat(iterset ** 1)
    dat.data[:] = 0
    be.par_loop(ref_kernel(X + "test_direct_loop_inc::k"), op2.ExtrudedSet(iterset, layers=10), dat(op2.INC))
    assert np.allclose(dat.data[:], 9.0)
    dat_f = op2.Dat(cell_set ** 1, np.zeros(nel * wedges), np.float64, "f")
    be.par_loop(ref_kernel(X + "test_extruded_layer_arg::blah"), elements, dat_f(op2.WRITE, field_map), pass_layer_arg=True)
    assert all((dat_f.data[wedges * n:wedges * (n + 1)] == np.arange(0, wedges)).all() for n in range(nel))
    be.par_loop(ref_kernel(X + "test_write_data_field::wo"), elements, dat_f(op2.WRITE, field_map))
    assert all(v == 42 for v in dat_f.data)
    dat_c = op2.Dat(node_set ** 2, np.zeros((nnodes * layers, 2)), np.float64, "c")
    be.par_loop(ref_kernel(X + "test_write_data_coords::wo_c"), elements, dat_c(op2.WRITE, coords_map))
    assert all(v[0] == 42 and v[1] == 42 for v in dat_c.data)
    be.par_loop(ref_kernel(X + "test_read_coord_neighbours_write_to_field::wtf"), elements, dat_f(op2.WRITE, field_map),
                dat_coords(op2.READ, coords_map))
    assert all(dat_f.data >= 0)
    # every cell's value = the sum of its six vertices' coordinates
    cm = np.asarray(coords_map.values_with_halo)
    exp = np.array([[dat_coords.data_ro[cm[e] + l].sum() for l in range(wedges)] for e in range(nel)]).ravel()
    assert_allclose(dat_f.data, exp, rtol=1e-14)
    dat_c = op2.Dat(node_set ** 2, np.zeros((nnodes * layers, 2)), np.float64, "c")
    be.par_loop(ref_kernel(X + "test_indirect_coords_inc::inc"), elements, dat_c(op2.RW, coords_map), dat_coords(op2.READ, coords_map))
    assert sum(sum(dat_c.data)) == nnodes * layers * 2


# ------------------------------------------------- test_direct_loop.py / test_global_reduction.py / test_vector_map.py /
# ------------------------------------------------- test_iteration_space_dats.py: the kernels whose text differs from the
# ------------------------------------------------- mirrored tests' own strings (tests/test_pyop2_loops.py) by layout only
D = "test_direct_loop.py::TestDirectLoop."


@pytest.fixture(params=[(nelems, nelems, nelems), (nelems // 2, nelems, nelems)])
def elems(request):
    return op2.Set(request.param, "elems")


def test_direct_loop_kernels(be, elems):
    """test_direct_loop.py:133-211"""
    xarray = lambda: np.array(range(nelems), dtype=np.uint32)           # noqa: E731
    x = op2.Dat(elems ** 1, xarray(), np.uint32, "x")
    g = op2.Global(1, 0, np.uint32, "g")
    be.par_loop(ref_kernel(D + "test_global_max_dat_is_max::k"), elems, g(op2.MAX), x(op2.READ))
    assert g.data[0] == x.data.max()
    g.data[0] = nelems * 2
    be.par_loop(ref_kernel(D + "test_global_max_g_is_max::k"), elems, x(op2.READ), g(op2.MAX))
    assert g.data[0] == nelems * 2
    g.data[0] = 1000
    be.par_loop(ref_kernel(D + "test_global_min_dat_is_min::k"), elems, g(op2.MIN), x(op2.READ))
    assert g.data[0] == x.data.min()
    g.data[0] = 10
    x.data[:] = 11
    be.par_loop(ref_kernel(D + "test_global_min_g_is_min::k"), elems, x(op2.READ), g(op2.MIN))
    assert g.data[0] == 10
    x = op2.Dat(elems ** 1, xarray(), np.uint32, "x")
    h = op2.Global(1, 1, np.uint32, "h")
    be.par_loop(ref_kernel(D + "test_global_read::global_read"), elems, x(op2.RW), h(op2.READ))
    assert sum(x.data_ro) == elems.size * (elems.size + 1) // 2
    y = op2.Dat(elems ** 2, [xarray(), xarray()], np.uint32, "y")
    be.par_loop(ref_kernel(D + "test_2d_dat::k2d_wo"), elems, y(op2.WRITE))
    assert all(all(v == [42, 43]) for v in y.data)


@pytest.mark.parametrize("which,dtype,init,expect", [
    ("min_uint32", np.uint32, 8, 8), ("min_int32", np.int32, 8, -12), ("max_int32", np.int32, -42, -12),
    ("min_float", np.float32, -.8, -12.0), ("max_float", np.float32, -42.8, -12.0),
    ("min_double", np.float64, -.8, -12.0), ("max_double", np.float64, -42.8, -12.0)])
def test_direct_global_min_max_by_type(be, which, dtype, init, expect):
    """test_global_reduction.py:138-258: a Global reduced against a Dat of 12s (unsigned) / -12s"""
    s = op2.Set(nelems, "set")
    g = op2.Global(1, init, dtype, "g")
    x = op2.Dat(s ** 1, [12 if dtype == np.uint32 else -12] * nelems, dtype)
    be.par_loop(ref_kernel(f"test_global_reduction.py::TestGlobalReductions.test_direct_{which}::k"), s,
                g(op2.MIN if which.startswith("min") else op2.MAX), x(op2.READ))
    assert_allclose(g.data[0], dtype(expect))


@pytest.mark.parametrize("f,T", [("test_vector_map.py", "TestVectorMap"), ("test_iteration_space_dats.py", "TestIterationSpaceDats")])
def test_vector_map_and_itspace_kernels(be, f, T):
    """test_vector_map.py:108-180 and test_iteration_space_dats.py:78-225"""
    nnodes, nele = 4096, 2048
    P = f"{f}::{T}."
    suffix = "vector_map" if "vector" in f else "itspace_map"
    node, ele = op2.Set(nnodes, "node"), op2.Set(nele, "ele")
    node2ele = op2.Map(node, ele, 1, np.arange(nnodes) // 2, "node2ele")
    nedges = nnodes - 1
    nodes, edges = op2.Set(nnodes, "nodes"), op2.Set(nedges, "edges")
    node_vals = op2.Dat(nodes, np.arange(nnodes, dtype=np.uint32), np.uint32, "node_vals")
    edge_vals = op2.Dat(edges, np.zeros(nedges, dtype=np.uint32), np.uint32, "edge_vals")
    edge2node = op2.Map(edges, nodes, 2, np.array([(i, i + 1) for i in range(nedges)], dtype=np.uint32), "edge2node")
    be.par_loop(ref_kernel(P + "test_sum_nodes_to_edges::sum"), edges, edge_vals(op2.WRITE if "vector" in f else op2.INC),
                node_vals(op2.READ, edge2node))
    assert all(np.arange(1, nedges * 2 + 1, 2) == edge_vals.data)
    d1, vd1 = op2.Dat(node, np.zeros(nnodes), dtype=np.int32), op2.Dat(ele, np.zeros(nele), dtype=np.int32)
    vd1.data[:] = np.arange(nele)
    be.par_loop(ref_kernel(P + f"test_read_1d_{suffix}::k"), node, d1(op2.WRITE), vd1(op2.READ, node2ele))
    assert all(d1.data[::2] == vd1.data) and all(d1.data[1::2] == vd1.data)
    be.par_loop(ref_kernel(P + f"test_write_1d_{suffix}::k"), node, vd1(op2.WRITE, node2ele))
    assert all(vd1.data == 2)
    vd1.data[:] = 3
    d1.data[:] = np.arange(nnodes).reshape(d1.data.shape)
    be.par_loop(ref_kernel(P + f"test_inc_1d_{suffix}::k"), node, vd1(op2.INC, node2ele), d1(op2.READ))
    expected = np.full_like(vd1.data, 3) + np.arange(0, nnodes, 2).reshape(vd1.data.shape) + np.arange(1, nnodes, 2).reshape(vd1.data.shape)
    assert all(vd1.data == expected)
    if "vector" in f:
        return
    d2, vd2 = op2.Dat(node ** 2, np.zeros(2 * nnodes), dtype=np.int32), op2.Dat(ele ** 2, np.zeros(2 * nele), dtype=np.int32)
    vd2.data[:] = np.arange(nele * 2).reshape(nele, 2)
    be.par_loop(ref_kernel(P + "test_read_2d_itspace_map::k"), node, d2(op2.WRITE), vd2(op2.READ, node2ele))
    assert all(d2.data[::2, 0] == vd2.data[:, 0]) and all(d2.data[::2, 1] == vd2.data[:, 1])
    assert all(d2.data[1::2, 0] == vd2.data[:, 0]) and all(d2.data[1::2, 1] == vd2.data[:, 1])
    be.par_loop(ref_kernel(P + "test_write_2d_itspace_map::k"), node, vd2(op2.WRITE, node2ele))
    assert all(vd2.data[:, 0] == 2) and all(vd2.data[:, 1] == 3)
    vd2.data[:, 0] = 3
    vd2.data[:, 1] = 4
    d2.data[:] = np.arange(2 * nnodes).reshape(d2.data.shape)
    be.par_loop(ref_kernel(P + "test_inc_2d_itspace_map::k"), node, vd2(op2.INC, node2ele), d2(op2.READ))
    e0 = 3 + np.arange(0, 2 * nnodes, 4) + np.arange(2, 2 * nnodes, 4)
    e1 = 4 + np.arange(1, 2 * nnodes, 4) + np.arange(3, 2 * nnodes, 4)
    assert all(vd2.data[:, 0] == e0) and all(vd2.data[:, 1] == e1)


def test_lifted_text_matches_the_reference_sources():
    """the committed kernel text is the reference's, character for character (only where /root/reference exists)"""
    import subprocess
    import sys
    if not os.path.isdir("/root/reference/tests/pyop2"):
        pytest.skip("the reference tree is not present on this machine")
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, "golden", "make_reference_kernels.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
