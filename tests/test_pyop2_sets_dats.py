"""More of the reference's PyOP2 tests, mirrored: tests/pyop2/test_subset.py:57-311 (loops over Subsets, subsets of
subsets, the subset matrix test, set algebra), test_dats.py:75-283 (copy constructors, copy, versions, axpy/maxpy),
test_indirect_loop.py:112-131 (argument validation), :256-277 (increment into a MixedDat) and
test_extrusion.py:401-450 (WRITE / RW through extruded maps).  Loops run through the HIP backend (``-m gpu``) and, where
the direct wrapper applies, through its host-sim."""
import numpy as np
import pytest

from firedrake_amd import op2
from firedrake_amd.parloop import DatParloopArg, GlobalParloopArg

nelems = 32


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


@pytest.fixture(params=[(nelems, nelems, nelems), (0, nelems, nelems), (nelems // 2, nelems, nelems)])
def iterset(request):
    return op2.Set(request.param, "iterset")


INC1 = "static void inc(unsigned int* v) { *v += 1; }"


# ---- test_subset.py --------------------------------------------------------------------------------------------------
def test_direct_loops_on_subsets(par_loop, iterset):
    k = op2.Kernel(INC1, "inc")
    indices = np.array([i for i in range(nelems) if not i % 2], dtype=np.int32)
    d = op2.Dat(iterset ** 1, data=None, dtype=np.uint32)
    par_loop(k, op2.Subset(iterset, indices), d(op2.RW))
    assert (np.where(d.data)[0] == indices).all()
    d = op2.Dat(iterset ** 1, data=None, dtype=np.uint32)                  # empty subset
    par_loop(k, op2.Subset(iterset, []), d(op2.RW))
    assert not d.data.any()
    even, odd = np.arange(0, nelems, 2, dtype=np.int32), np.arange(1, nelems, 2, dtype=np.int32)
    for se, so in ((op2.Subset(iterset, even), op2.Subset(iterset, odd)), (iterset(even), iterset(odd))):
        d = op2.Dat(iterset ** 1, data=None, dtype=np.uint32)              # complementary subsets
        par_loop(k, se, d(op2.RW))
        par_loop(k, so, d(op2.RW))
        assert (d.data == 1).all()
    for sss in (op2.Subset(op2.Subset(iterset, even), np.arange(0, nelems // 2, 2, dtype=np.int32)),
                iterset(even)(np.arange(0, nelems // 2, 2, dtype=np.int32))):
        d, d2 = (op2.Dat(iterset ** 1, data=None, dtype=np.uint32) for _ in range(2))       # a subset of a subset
        par_loop(k, sss, d(op2.RW))
        par_loop(k, iterset(np.arange(0, nelems, 4, dtype=np.int32)), d2(op2.RW))
        assert (d.data == d2.data).all()


def test_indirect_loops_on_subsets(par_loop, iterset):
    indices = np.array([i for i in range(nelems) if not i % 2], dtype=np.int32)
    ss = op2.Subset(iterset, indices)
    indset = op2.Set(2, "indset")
    m = op2.Map(iterset, indset, 1, [(1 if i % 2 else 0) for i in range(nelems)])
    k = op2.Kernel("static void inc(unsigned int* v) { *v += 1;}", "inc")
    d = op2.Dat(indset ** 1, data=None, dtype=np.uint32)
    par_loop(k, ss, d(op2.INC, m))
    assert d.data[0] == nelems // 2
    d = op2.Dat(indset ** 1, data=None, dtype=np.uint32)                   # empty subset
    par_loop(k, op2.Subset(iterset, []), d(op2.INC, m))
    assert (d.data == 0).all()
    values = [2976579765] * nelems                                         # indirect loop with a direct Dat
    values[::2] = [i // 2 for i in range(nelems)][::2]
    dat1 = op2.Dat(iterset ** 1, data=values, dtype=np.uint32)
    dat2 = op2.Dat(indset ** 1, data=None, dtype=np.uint32)
    par_loop(op2.Kernel("static void inc(unsigned* d, unsigned int* s) { *d += *s;}", "inc"), ss, dat2(op2.INC, m), dat1(op2.READ))
    assert dat2.data[0] == sum(values[::2])
    even, odd = np.arange(0, nelems, 2, dtype=np.int32), np.arange(1, nelems, 2, dtype=np.int32)
    big = op2.Set(nelems, "indset")                                        # complementary subsets, two arguments
    idm = op2.Map(iterset, big, 1, list(range(nelems)))
    dat1 = op2.Dat(iterset ** 1, data=None, dtype=np.uint32)
    dat2 = op2.Dat(big ** 1, data=None, dtype=np.uint32)
    k2 = op2.Kernel("static void inc(unsigned int* v1, unsigned int* v2) {\n  *v1 += 1;\n  *v2 += 1;\n}", "inc")
    par_loop(k2, op2.Subset(iterset, even), dat1(op2.RW), dat2(op2.INC, idm))
    par_loop(k2, op2.Subset(iterset, odd), dat1(op2.RW), dat2(op2.INC, idm))
    assert np.sum(dat1.data) == nelems and np.sum(dat2.data) == nelems


@pytest.mark.gpu
def test_subset_matrix():
    """test_subset.py:210-256: assembling over Subsets that cover the set gives the same matrix."""
    iterset, idset, indset = op2.Set(2), op2.Set(2), op2.Set(4)
    ss01, ss10 = op2.Subset(iterset, [0, 1]), op2.Subset(iterset, [1, 0])
    dat = op2.Dat(idset ** 1, data=[0, 1], dtype=op2.ScalarType)
    m = op2.Map(iterset, indset, 4, [0, 1, 2, 3, 0, 1, 2, 3])
    idmap = op2.Map(iterset, idset, 1, [0, 1])
    sparsity = op2.Sparsity((indset ** 1, indset ** 1), {(0, 0): [(m, m, None)]})
    mat, mat01, mat10 = (op2.Mat(sparsity, op2.ScalarType) for _ in range(3))
    k = op2.Kernel("""
static void unique_id(PetscScalar mat[16], PetscScalar *dat) {
  for (int i=0; i<4; ++i)
    for (int j=0; j<4; ++j)
      mat[i*4+j] += (*dat)*16+i*4+j;
}
        """, "unique_id")
    for a, it in ((mat, iterset), (mat01, ss01), (mat10, ss10)):
        a.zero()
        op2.par_loop(k, it, a(op2.INC, (m, m)), dat(op2.READ, idmap))
        a.assemble()
    assert (mat01.values == mat.values).all() and (mat10.values == mat.values).all()
    assert mat.values.any()


def test_set_algebra():
    """test_subset.py:262-311."""
    a = op2.Set(10)
    assert a.union(a) is a and a.intersection(a) is a
    assert a.difference(a)._indices.size == 0 and a.symmetric_difference(a)._indices.size == 0
    b = op2.Subset(a, np.array([2, 3, 5, 7], dtype=np.int32))
    assert a.union(b) is a and a.intersection(b) is b
    assert (a.difference(b)._indices == [0, 1, 4, 6, 8, 9]).all()
    assert (a.symmetric_difference(b)._indices == a.difference(b)._indices).all()
    assert b.union(a) is a and b.intersection(a) is b and b.difference(a)._indices.size == 0
    assert (b.symmetric_difference(a)._indices == [0, 1, 4, 6, 8, 9]).all()
    c = op2.Subset(a, np.array([2, 4, 6, 8], dtype=np.int32))
    assert (b.union(c)._indices == [2, 3, 4, 5, 6, 7, 8]).all() and (b.intersection(c)._indices == [2]).all()
    assert (b.difference(c)._indices == [3, 5, 7]).all() and (b.symmetric_difference(c)._indices == [3, 4, 5, 6, 7, 8]).all()
    with pytest.raises(ValueError):
        a.union(op2.Set(10))
    with pytest.raises(TypeError):
        b.union(op2.Subset(op2.Set(10), [1]))


# ---- test_dats.py ----------------------------------------------------------------------------------------------------
@pytest.fixture
def s():
    return op2.Set(5)


@pytest.fixture
def d1(s):
    return op2.Dat(s, list(range(5)), dtype=op2.ScalarType)


def test_dat_copies(s, d1):
    d2 = op2.Dat(d1)                                                       # copy constructor copies values
    assert d1.dataset.set == d2.dataset.set and (d1.data_ro == d2.data_ro).all()
    d1.data[:] = -1
    assert (d1.data_ro != d2.data_ro).all()
    d1.data[:] = np.arange(5)
    mdat = op2.MixedDat([d1, d1])
    mdat2 = op2.MixedDat(mdat)
    assert mdat.dataset.set == mdat2.dataset.set
    assert all(all(d.data_ro == d_.data_ro) for d, d_ in zip(mdat, mdat2))
    for dat in mdat.data:
        dat[:] = -1
    assert all(all(d.data_ro != d_.data_ro) for d, d_ in zip(mdat, mdat2))
    d1.data[:] = np.arange(5)
    d2 = op2.Dat(s)                                                        # Dat.copy
    d1.copy(d2)
    assert (d1.data_ro == d2.data_ro).all()
    d1.data[:] = -1
    assert (d1.data_ro != d2.data_ro).all()
    mdat2 = op2.MixedDat([s, s])                                           # MixedDat.copy
    mdat.copy(mdat2)
    assert all(all(d.data_ro == d_.data_ro) for d, d_ in zip(mdat, mdat2))
    with pytest.raises(NotImplementedError):
        mdat.copy(op2.MixedDat([s, s]), subset=op2.Subset(s, []))
    for dim in (1, 2):
        assert op2.Dat(op2.Set(10) ** dim).nbytes == 10 * np.dtype(op2.ScalarType).itemsize * dim


def test_dat_version_on_host(s, d1):
    """test_dats.py:144-168, 260-265 (the accessor, zero and copy rules)."""
    d2 = op2.Dat(s)
    assert d1.dat_version == 0 and d2.dat_version == 0
    d1.data
    assert d1.dat_version == 1 and d2.dat_version == 0
    d2.data[:] += 1
    assert d1.dat_version == 1 and d2.dat_version == 1
    d1.zero()
    assert d1.dat_version == 2 and d2.dat_version == 1
    d2.copy(d1)
    assert d1.dat_version == 3 and d2.dat_version == 1
    d3 = op2.Dat(s, list(range(5)), dtype=op2.ScalarType)
    d3.data_ro_with_halos
    assert d3.dat_version == 0
    d3.data_with_halos
    assert d3.dat_version == 1
    d4 = op2.Dat(s)                                                        # mixed versions are the sum of the parts
    md, md2 = op2.MixedDat([d3, d3]), op2.MixedDat([d3, d4])
    assert md.dat_version == 2 and md2.dat_version == 1
    md2.data
    assert d3.dat_version == 2 and d4.dat_version == 1 and md.dat_version == 4 and md2.dat_version == 3
    md.zero()
    assert d3.dat_version == 4 and md.dat_version == 8 and md2.dat_version == 5


@pytest.mark.gpu
def test_dat_version_copy_subset_axpy_on_device(s, d1):
    d3 = op2.Dat(s ** 1, data=None, dtype=np.uint32)
    op2.par_loop(op2.Kernel("static void write(unsigned int* v) {\n  *v = 1;\n}", "write"), s, d3(op2.WRITE))
    assert d3.dat_version == 1                                             # test_dats.py:196-206
    d4 = op2.Dat(s ** 1, data=None, dtype=np.uint32)
    d3d4 = op2.MixedDat([d3, d4])
    m = op2.Map(s, op2.Set(5), 1, values=[0, 1, 2, 3, 4])
    d3.zero()
    v3 = d3.dat_version
    k = op2.Kernel("static void write(unsigned int* v) {\n  v[0] = 1;\n  v[1] = 2;\n}", "write")
    op2.par_loop(k, s, d3d4(op2.WRITE, op2.MixedMap([m, m])))             # test_dats.py:244-258
    assert d3.dat_version == v3 + 1 and d4.dat_version == 1
    assert (d3.data_ro == 1).all() and (d4.data_ro == 2).all()
    d2 = op2.Dat(s)                                                        # test_copy_subset
    ss = op2.Subset(s, list(range(1, 5, 2)))
    d1.copy(d2, subset=ss)
    assert (d1.data_ro[ss.indices] == d2.data_ro[ss.indices]).all() and (d2.data_ro[::2] == 0).all()
    a, b, c = op2.Dat(d1.dataset), op2.Dat(d1.dataset), op2.Dat(d1.dataset)  # test_axpy / test_maxpy
    b.data[:] = 2
    c.data[:] = 3
    a.axpy(3, b)
    assert (a.data_ro == 3 * 2).all()
    a.data[:] = 0
    a.maxpy((2, 3), (b, c))
    assert (a.data_ro == 2 * 2 + 3 * 3).all()


# ---- test_indirect_loop.py -------------------------------------------------------------------------------------------
def test_indirect_argument_validation(iterset):
    indset = op2.Set(nelems, "indset")
    x = op2.Dat(indset, list(range(nelems)), np.uint32, "x")
    with pytest.raises(op2.MapValueError):                                 # map on another iteration set
        op2.LegacyParloop(op2.Kernel("", "dummy"), iterset, x(op2.WRITE, op2.Map(op2.Set(nelems), indset, 1)))
    with pytest.raises(op2.MapValueError):                                 # map into another set than the Dat's
        x(op2.WRITE, op2.Map(iterset, op2.Set(nelems), 1))
    with pytest.raises(op2.MapValueError):                                 # map without values
        op2.LegacyParloop(op2.Kernel("static void wo(unsigned int* x) { *x = 42; }\n", "wo"), iterset,
                          x(op2.WRITE, op2.Map(iterset, indset, 1)))


@pytest.mark.parametrize("body", ["d[0] += x[0]; d[1] += x[0];", "for (int i=0; i<2; ++i)\n            d[i] += x[0];"])
def test_mixed_non_mixed_dat(par_loop, iterset, body):
    """test_indirect_loop.py:256-277: increment into a MixedDat (identity map x all-to-one map) from a plain Dat."""
    indset, unitset = op2.Set(nelems, "indset"), op2.Set(1, "unitset")
    mdat = op2.MixedDat(op2.MixedSet((indset, unitset)))
    mmap = op2.MixedMap((op2.Map(iterset, indset, 1, np.arange(nelems)[::-1].copy(), "iterset2indset"),
                         op2.Map(iterset, unitset, 1, np.zeros(nelems, dtype=np.uint32), "iterset2unitset")))
    d = op2.Dat(iterset, np.ones(iterset.total_size), dtype=op2.ScalarType)
    if par_loop is op2.par_loop:
        par_loop(op2.Kernel("static void inc(double *d, double *x) {\n  %s\n}" % body, "inc"), iterset, mdat(op2.INC, mmap), d(op2.READ))
    else:
        from hostsim import run_direct
        pl = op2.LegacyParloop(op2.Kernel("static void inc(double *d, double *x) {\n  %s\n}" % body, "inc"), iterset,
                               mdat(op2.INC, mmap), d(op2.READ))
        outs = run_direct(pl)                                              # flattened: part 0, part 1, d
        mdat[0]._host_rw()[...] = outs[0]
        mdat[1]._host_rw()[...] = outs[1]
    assert all(mdat[0].data == 1.0) and mdat[1].data == float(nelems)


# ---- test_extrusion.py:401-450 ---------------------------------------------------------------------------------------
def _strip(layers=11, nel=8):
    """A strip of triangles extruded to wedges: coordinates on (vertex, level) nodes, one field value per cell."""
    nx = nel // 2
    cells = []
    for i in range(nx):
        a, b, c, d = i, i + 1, nx + 1 + i, nx + 2 + i
        cells += [(a, b, c), (b, d, c)]
    cells = np.array(cells, dtype=np.int32)
    nb = 2 * (nx + 1)
    cmap = np.empty((nel, 6), dtype=np.int32)
    for e in range(nel):
        for k in range(3):
            cmap[e, 2 * k] = cells[e, k] * layers
            cmap[e, 2 * k + 1] = cells[e, k] * layers + 1
    base = op2.Set(nel)
    ext = op2.ExtrudedSet(base, layers=layers)
    nodes, fset = op2.Set(nb * layers), op2.Set(nel * (layers - 1))
    rng = np.random.default_rng(3)
    return (ext, nodes, fset, op2.Dat(nodes ** 2, rng.uniform(0.0, 1.0, (nb * layers, 2))),
            op2.Map(ext, nodes, 6, cmap, offset=[1] * 6), op2.Map(ext, fset, 1, np.arange(nel) * (layers - 1), offset=[1]))


def test_extruded_write_and_rw(par_loop):
    ext, nodes, fset, coords, cmap, fmap = _strip()
    layers = 11
    dat_c = op2.Dat(nodes ** 2, np.zeros((nodes.size, 2)))
    kernel_wo_c = "static void wo_c(double x[12]) {\n for (int i = 0; i < 12; ++i) x[i] = 42.0;\n}"      # test_write_data_coords
    par_loop(op2.Kernel(kernel_wo_c, "wo_c"), ext, dat_c(op2.WRITE, cmap))
    assert all(map(lambda v: v[0] == 42 and v[1] == 42, dat_c.data))
    dat_f = op2.Dat(fset, np.zeros(fset.size))
    kernel_wtf = """static void wtf(double* y, double x[12]) {
           double sum = 0.0;
           for (int i=0; i<6; i++){
                sum += x[i*2] + x[i*2+1];
           }
           y[0] = sum;
        }"""                                                                                                # test_read_coord_neighbours_write_to_field
    par_loop(op2.Kernel(kernel_wtf, "wtf"), ext, dat_f(op2.WRITE, fmap), coords(op2.READ, cmap))
    x = coords.data_ro
    ref = np.array([[x[cmap.values[e] + l].sum() for l in range(layers - 1)] for e in range(ext.size)]).ravel()
    assert np.allclose(dat_f.data, ref, rtol=1e-14) and all(dat_f.data >= 0)
    dat_c = op2.Dat(nodes ** 2, np.zeros((nodes.size, 2)))
    kernel_inc = """static void inc(double y[12], double x[12]) {
           for (int i=0; i<6; i++){
             if (y[i*2+0] == 0){
                y[i*2+0] += 1;
                y[i*2+1] += 1;
             }
           }
        }"""                                                                                                # test_indirect_coords_inc (RW)
    # concurrent cells may both see a shared node at 0 and both write 1: the outcome is the sequential one
    par_loop(op2.Kernel(kernel_inc, "inc"), ext, dat_c(op2.RW, cmap), coords(op2.READ, cmap))
    assert dat_c.data.sum() == nodes.size * 2


# ---- DatView (pyop2/types/dat.py:714-805, tests/pyop2/test_dats.py:286-340) -------------------------------------------
@pytest.mark.gpu
def test_dat_view_on_device():
    rng = np.random.default_rng(31)
    s = op2.Set(2000)
    it = op2.Set(3000)
    m = op2.Map(it, s, 2, rng.integers(0, 2000, size=(3000, 2)))
    v = op2.Dat(s ** 3, rng.standard_normal((2000, 3)))
    out = op2.Dat(s ** 2, rng.standard_normal((2000, 2)))
    V, O = v.data_ro.copy(), out.data_ro.copy()
    k = op2.Kernel("static void kvw(double *o, const double *a) { for (int i = 0; i < 2; ++i) o[i] += 2.0*a[i] + i; }", "kvw")
    op2.par_loop(k, it, op2.DatView(out, 1)(op2.INC, m), op2.DatView(v, 2)(op2.READ, m))
    exp = O.copy()
    np.add.at(exp[:, 1], m.values[:, 0], 2.0 * V[m.values[:, 0], 2])
    np.add.at(exp[:, 1], m.values[:, 1], 2.0 * V[m.values[:, 1], 2] + 1)
    assert np.abs(out.data_ro - exp).max() < 1e-12 and np.array_equal(out.data_ro[:, 0], O[:, 0])
    comp = op2.DatView(out, 0)
    comp.zero()                                                  # one component zeroed on the device, the other untouched
    assert (out.data_ro[:, 0] == 0).all() and np.abs(out.data_ro[:, 1] - exp[:, 1]).max() < 1e-12
    kd = op2.Kernel("static void kd(double *x, const double *y) { x[0] = 3.0*y[0]; }", "kd")
    op2.par_loop(kd, s, comp(op2.WRITE), op2.DatView(v, 1)(op2.READ))
    assert np.allclose(out.data_ro[:, 0], 3.0 * V[:, 1])
    assert comp.dat_version == out.dat_version
