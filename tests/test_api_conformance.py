"""CPU: the user-facing API of the PyOP2-shaped carriers behaves like the reference's -- constructor validation and
exception types, container protocol, equality, ``str``/``repr`` formats.  Condensed from the behaviours
tests/pyop2/test_api.py:190-1620 pins (class by class, cited below); no device is needed for any of it."""
import numpy as np
import pytest
from numpy.testing import assert_equal

from firedrake_amd import exceptions, op2
from firedrake_amd.op2 import DataSet, MixedDataSet, MixedSet, Set  # noqa: F401  (needed by eval(repr(...)))


@pytest.fixture
def set_():
    return op2.Set(5, 'foo')


@pytest.fixture
def iterset():
    return op2.Set(2, 'iterset')


@pytest.fixture
def toset():
    return op2.Set(3, 'toset')


@pytest.fixture
def sets(set_, iterset, toset):
    return set_, iterset, toset


@pytest.fixture
def mset(sets):
    return op2.MixedSet(sets)


@pytest.fixture(params=['sets', 'mset', 'gen'])
def msets(sets, mset, request):
    return {'sets': sets, 'mset': mset, 'gen': iter(sets)}[request.param]


@pytest.fixture(params=[1, 2, (2, 3)])
def dset(request, set_):
    return op2.DataSet(set_, request.param, 'dfoo')


@pytest.fixture
def diterset(iterset):
    return op2.DataSet(iterset, 1, 'diterset')


@pytest.fixture
def dtoset(toset):
    return op2.DataSet(toset, 1, 'dtoset')


@pytest.fixture
def dsets(dset, diterset, dtoset):
    return dset, diterset, dtoset


@pytest.fixture
def mdset(dsets):
    return op2.MixedDataSet(dsets)


# ---- Set (test_api.py:209-263) ---------------------------------------------------------------------------------------
def test_set_api(set_, dset):
    with pytest.raises(exceptions.SizeTypeError):
        op2.Set('illegalsize')
    with pytest.raises(exceptions.NameTypeError):
        op2.Set(1, 2)
    assert [s for s in set_] == [set_] and len(set_) == 1
    assert isinstance(eval(repr(set_)), op2.Set)
    assert str(set_) == "OP2 Set: %s with size %s" % (set_.name, set_.size)
    assert set_ == set_ and not set_ != set_
    assert dset in set_
    assert dset not in op2.Set(5, 'bar')
    d = set_ ** 1
    assert isinstance(d, op2.DataSet) and d.cdim == 1 and (set_ ** 3).cdim == 3
    assert isinstance(set_, op2.Set) and not isinstance(set_, op2.Dat)


# ---- ExtrudedSet (test_api.py:266-301) -------------------------------------------------------------------------------
def test_extruded_set_api(set_, iterset, toset):
    with pytest.raises(exceptions.SizeTypeError):
        op2.ExtrudedSet(set_, 1)
    with pytest.raises(TypeError):
        op2.ExtrudedSet(1, 3)
    e = op2.ExtrudedSet(set_, 5)
    assert set_ in e and iterset not in e
    # maps on the base set are legal on the extruded set, maps on another set are not
    m_ok = op2.Map(iterset, toset, 2, [1] * 2 * iterset.size, 'm_iterset_toset')
    e2 = op2.ExtrudedSet(iterset, 5)
    dat = op2.Dat(toset ** 1, np.arange(toset.size, dtype=np.int32))
    k = op2.Kernel('static void k() { }', 'k')
    op2.ParLoop(k, e2, dat(op2.READ, m_ok))
    with pytest.raises(exceptions.MapValueError):
        op2.ParLoop(k, e, dat(op2.READ, m_ok))


# ---- Subset (test_api.py:304-355) ------------------------------------------------------------------------------------
def test_subset_api(set_):
    with pytest.raises(TypeError):
        op2.Subset("fail", [0, 1])
    with pytest.raises(exceptions.SubsetIndexOutOfBounds):
        op2.Subset(set_, list(range(set_.total_size + 1)))
    with pytest.raises(exceptions.SubsetIndexOutOfBounds):
        op2.Subset(set_, [-1])
    assert len(op2.Subset(set_, []).indices) == 0
    assert_equal(set_(0, 1).indices, op2.Subset(set_, [0, 1]).indices)
    assert_equal(set_(0).indices, op2.Subset(set_, [0]).indices)
    assert_equal(set_(np.arange(5)).indices, op2.Subset(set_, np.arange(5)).indices)
    ss = op2.Subset(set_, [0, 0, 1, 1])
    assert np.sum(ss.indices == 0) == 1 and np.sum(ss.indices == 1) == 1
    assert_equal(op2.Subset(set_, [0, 4, 1, 2, 3]).indices, list(range(5)))


# ---- MixedSet (test_api.py:358-455) ----------------------------------------------------------------------------------
def test_mixed_set_api(sets, mset, set_, iterset, toset):
    with pytest.raises(TypeError):
        op2.MixedSet(('foo', 'bar'))
    assert all(mset[i] == s for i, s in enumerate(sets))
    assert op2.MixedSet(sets).split == sets
    assert mset.core_size == sum(s.core_size for s in mset)
    assert mset.size == sum(s.size for s in mset)
    assert mset.total_size == sum(s.total_size for s in mset)
    assert mset.sizes == (mset.core_size, mset.size, mset.total_size)
    assert mset.name == tuple(s.name for s in mset)
    assert mset.halo is None
    assert mset.layers == mset[0].layers
    with pytest.raises(AssertionError):
        op2.MixedSet([op2.ExtrudedSet(s, layers=i + 4) for i, s in enumerate(sets)])
    assert tuple(s for s in mset) == sets and len(op2.MixedSet(sets)) == len(sets)
    ref = op2.MixedDataSet([s ** 1 for s in mset])
    assert mset ** 1 == ref and mset ** ((1,) * len(mset)) == ref and mset ** (1 for _ in mset) == ref
    assert op2.MixedSet(sets) == op2.MixedSet(sets) and not op2.MixedSet(sets) != op2.MixedSet(sets)
    assert op2.MixedSet((set_, iterset, toset)) != op2.MixedSet((set_, toset, iterset))
    assert op2.MixedSet(sets) != sets[0] and not op2.MixedSet(sets) == sets[0]
    assert isinstance(eval(repr(mset)), op2.MixedSet)
    assert str(mset) == "OP2 MixedSet composed of Sets: %s" % (mset._sets,)


# ---- DataSet (test_api.py:458-535) -----------------------------------------------------------------------------------
def test_dataset_api(iterset, dset):
    with pytest.raises(TypeError):
        op2.DataSet(iterset, 'illegaldim')
    with pytest.raises(TypeError):
        op2.DataSet(iterset, (1, 'illegaldim'))
    with pytest.raises(exceptions.NameTypeError):
        op2.DataSet(iterset, 1, 2)
    assert op2.DataSet(iterset).dim == (1,) and op2.DataSet(iterset, 1).dim == (1,)
    assert op2.DataSet(iterset, [2, 3]).dim == (2, 3)
    assert [s for s in dset] == [dset] and len(dset) == 1
    assert isinstance(eval(repr(dset)), op2.DataSet)
    assert str(dset) == "OP2 DataSet: %s on set %s, with dim %s, %s" % (dset.name, dset.set, dset.dim, False)
    copy = op2.DataSet(dset.set, dset.dim)
    assert copy == dset and not copy != dset
    other = op2.DataSet(op2.Set(dset.set.size), dset.dim)
    assert other != dset and not other == dset
    other = op2.DataSet(dset.set, tuple(d + 1 for d in dset.dim))
    assert other != dset and not other == dset
    assert op2.Dat(dset) in dset
    assert op2.Dat(dset) not in op2.DataSet(op2.Set(5, 'bar'))


# ---- MixedDataSet (test_api.py:538-665) ------------------------------------------------------------------------------
@pytest.mark.parametrize('arg', ['illegalarg', (set, 'illegalarg'), iter((set, 'illegalarg'))])
def test_mixed_dset_illegal_arg(arg):
    with pytest.raises(TypeError):
        op2.MixedDataSet(arg)


def test_mixed_dataset_api(dsets, msets, mset, sets, set_, dset, diterset, dtoset, mdset):
    for dims in ('illegaldim', (1, 2, 'illegaldim')):
        with pytest.raises((TypeError, ValueError)):
            op2.MixedDataSet(dsets, dims)
    with pytest.raises(TypeError):
        op2.MixedDataSet(dsets, 1)
    assert op2.MixedDataSet((set_, dset)).split == (set_ ** 1, dset)
    assert op2.MixedDataSet(iter((set_, dset))).split == (set_ ** 1, dset)
    assert all(mdset[i] == ds for i, ds in enumerate(mdset))
    assert op2.MixedDataSet(dsets).split == dsets
    assert mdset.dim == tuple(s.dim for s in mdset) and mdset.cdim == sum(s.cdim for s in mdset)
    assert mdset.name == tuple(s.name for s in mdset)
    assert op2.MixedDataSet(mset).set == mset
    assert tuple(s for s in mdset) == dsets and len(op2.MixedDataSet(dsets)) == len(dsets)
    assert op2.MixedDataSet(dsets) == op2.MixedDataSet(dsets) and not op2.MixedDataSet(dsets) != op2.MixedDataSet(dsets)
    assert op2.MixedDataSet((dset, diterset, dtoset)) != op2.MixedDataSet((dset, dtoset, diterset))
    assert op2.MixedDataSet((diterset, dtoset)) != diterset and not op2.MixedDataSet((diterset, dtoset)) == diterset
    assert isinstance(eval(repr(mdset)), op2.MixedDataSet)
    assert str(mdset) == "OP2 MixedDataSet composed of DataSets: %s" % (mdset._dsets,)


def test_mixed_dataset_dims(msets, mset, sets):
    if not isinstance(msets, tuple) and not isinstance(msets, op2.MixedSet):
        msets = tuple(msets)          # a generator can be consumed once
    assert op2.MixedDataSet(msets) == mset ** 1
    assert op2.MixedDataSet(msets).dim == ((1,),) * len(mset)
    assert op2.MixedDataSet(msets, 2).dim == ((2,),) * len(mset)
    assert op2.MixedDataSet(msets, (2 for _ in mset)).dim == ((2,),) * len(mset)
    dims = ((2,), (2, 2), (1,))
    assert op2.MixedDataSet(msets, dims).dim == dims
    with pytest.raises(ValueError):
        op2.MixedDataSet(msets, list(range(1, len(sets))))


# ---- Dat (test_api.py:668-834) ---------------------------------------------------------------------------------------
@pytest.fixture
def dat(dtoset):
    return op2.Dat(dtoset, np.arange(dtoset.cdim * dtoset.size, dtype=np.int32))


@pytest.fixture
def dats(dtoset, dset):
    return op2.Dat(dtoset), op2.Dat(dset)


@pytest.fixture
def mdat(dats):
    return op2.MixedDat(dats)


def test_dat_construction(dset, set_):
    with pytest.raises(exceptions.DataSetTypeError):
        op2.Dat('illegalset', 1)
    with pytest.raises(exceptions.NameTypeError):
        op2.Dat(dset, name=2)
    d = op2.Dat(dset)
    assert d.data.size == dset.size * dset.cdim and d.data.dtype == op2.ScalarType and isinstance(d.dtype, np.dtype)
    assert not d._is_allocated                                   # no device storage until a kernel needs it
    assert op2.Dat(dset, dtype=np.int32).data.dtype == np.int32
    n = dset.size * dset.cdim
    assert op2.Dat(dset, [1.0] * n).dtype == np.float64
    assert op2.Dat(dset, [1] * n).dtype == np.asarray(1).dtype
    assert op2.Dat(dset, [1] * n, np.double).dtype == np.double
    assert op2.Dat(dset, [1.5] * n, np.int32).dtype == np.int32
    with pytest.raises(exceptions.DataTypeError):
        op2.Dat(dset, dtype='illegal_type')
    with pytest.raises(exceptions.DataValueError):
        op2.Dat(dset, [1] * (n + 1))
    assert op2.Dat(dset, [1.0] * n).data.shape == (dset.size,) + (() if dset.cdim == 1 else dset.dim)
    d = op2.Dat(dset, [1] * n, 'double', 'bar')
    assert d.dataset.set == dset.set and d.dtype == np.float64 and d.name == 'bar' and d.data.sum() == n
    d1 = op2.Dat(set_)
    assert d1.cdim == 1 and isinstance(d1.dataset, op2.DataSet) and d1.dataset.cdim == 1
    z = op2.Dat(set_ ** 0)                                       # zero-dim DataSets are allowed (test_api.py:828-834)
    assert z.shape == (set_.total_size, 0) and z._data.size == 0


def test_dat_protocol(dat, dset):
    from firedrake_amd.op2 import Dat  # noqa: F401
    from numpy import dtype  # noqa: F401
    assert dat[0] is dat and dat.split == (dat,) and [d for d in dat] == [dat] and len(dat) == 1
    with pytest.raises(exceptions.IndexValueError):
        dat[1]
    assert dat(op2.READ).map_ is None
    set1, set2 = op2.Set(3), op2.Set(2)
    with pytest.raises(exceptions.MapValueError):
        op2.Dat(dset)(op2.READ, op2.Map(set1, set2, 1, [0, 0, 0]))
    assert isinstance(eval(repr(dat)), op2.Dat)
    d = op2.Dat(dset, dtype='double', name='bar')
    assert str(d) == "OP2 Dat: %s on (%s) with datatype %s" % (d.name, d.dataset, d.data.dtype.name)
    x = dat.data_ro
    with pytest.raises((RuntimeError, ValueError)):
        x[0] = 1
    x = dat.data
    x[0] = -100
    assert (dat.data_ro[0] == -100).all()


# ---- MixedDat (test_api.py:837-947) ----------------------------------------------------------------------------------
def test_mixed_dat_api(set_, dats, mdset, mset, mdat):
    from firedrake_amd.op2 import Dat, MixedDat  # noqa: F401
    from numpy import dtype  # noqa: F401
    with pytest.raises(exceptions.DataSetTypeError):
        op2.MixedDat('illegalarg')
    with pytest.raises(exceptions.DataValueError):
        op2.MixedDat((op2.Dat(set_, dtype=np.int32), op2.Dat(set_)))
    assert op2.MixedDat(dats).split == dats
    assert op2.MixedDat(mdset).dataset == mdset
    assert op2.MixedDat(mset).dataset == op2.MixedDataSet(mset)
    assert all(mdat[i] == d for i, d in enumerate(mdat)) and mdat[:-1] == tuple(mdat)[:-1]
    assert op2.MixedDat(mdset).dim == mdset.dim and op2.MixedDat(mdset).cdim == mdset.cdim
    assert mdat.halo_valid                                       # before any writable access (data marks halos dirty)
    for attr in ("data_ro", "data_ro_with_halos", "data", "data_with_halos"):
        assert all((d1 == getattr(d2, attr)).all() for d1, d2 in zip(getattr(mdat, attr), mdat))
    mdat.halo_valid = True
    mdat[0].halo_valid = False
    assert not mdat.halo_valid
    mdat.halo_valid = True
    mdat.halo_valid = False
    assert not any(d.halo_valid for d in mdat)
    assert tuple(s for s in mdat) == dats and len(op2.MixedDat(dats)) == len(dats)
    assert op2.MixedDat(dats) == op2.MixedDat(dats) and not op2.MixedDat(dats) != op2.MixedDat(dats)
    assert op2.MixedDat(dats) != op2.MixedDat(reversed(dats))
    assert op2.MixedDat(dats) != dats[0] and not op2.MixedDat(dats) == dats[0]
    assert isinstance(eval(repr(mdat)), op2.MixedDat)
    assert str(mdat) == "OP2 MixedDat composed of Dats: %s" % (mdat.split,)


# ---- shared map fixtures (test_api.py:118-187) -----------------------------------------------------------------------
@pytest.fixture
def m_iterset_toset(iterset, toset):
    return op2.Map(iterset, toset, 2, [1] * 2 * iterset.size, 'm_iterset_toset')


@pytest.fixture
def m_iterset_set(iterset, set_):
    return op2.Map(iterset, set_, 2, [1] * 2 * iterset.size, 'm_iterset_set')


@pytest.fixture
def m_set_toset(set_, toset):
    return op2.Map(set_, toset, 1, [1] * set_.size, 'm_set_toset')


@pytest.fixture
def m_set_set(set_):
    return op2.Map(set_, set_, 1, [1] * set_.size, 'm_set_set')


@pytest.fixture
def maps(m_iterset_toset, m_iterset_set):
    return m_iterset_toset, m_iterset_set


@pytest.fixture
def mmap(maps):
    return op2.MixedMap(maps)


@pytest.fixture
def mds(dtoset, set_):
    return op2.MixedDataSet((dtoset, set_))


@pytest.fixture(params=[('mds', 'mds', 'mmap', 'mmap'), ('mds', 'dtoset', 'mmap', 'm_iterset_toset'),
                        ('dtoset', 'mds', 'm_iterset_toset', 'mmap')])
def ms(request):
    rds, cds, rmm, cmm = [request.getfixturevalue(p) for p in request.param]
    return op2.Sparsity((rds, cds), {(i, j): [(rm, cm, None)] for i, rm in enumerate(rmm) for j, cm in enumerate(cmm)})


@pytest.fixture
def sparsity(m_iterset_toset, dtoset):
    return op2.Sparsity((dtoset, dtoset), [(m_iterset_toset, m_iterset_toset, None)])


@pytest.fixture
def mat(sparsity):
    return op2.Mat(sparsity)


# ---- Sparsity (test_api.py:950-1135) ---------------------------------------------------------------------------------
def test_sparsity_validation(toset, iterset, m_iterset_toset):
    mi = op2.Map(op2.Set(3, 'iterset2'), toset, 1, [1] * 3, 'mi')
    dataset2 = op2.Set(1, 'dataset2')
    md = op2.Map(iterset, dataset2, 1, [0] * iterset.size, 'md')
    di, dd = op2.DataSet(toset, 1, 'di'), op2.DataSet(dataset2, 1, 'dd')
    for bad in ((('illegalrmap', di), [(mi, mi, None)]), ((di, 'illegalrmap'), [(mi, mi, None)]),
                ((di, di), [('illegalrmap', mi, None)]), ((di, di), [(mi, 'illegalcmap', None)])):
        with pytest.raises(TypeError):
            op2.Sparsity(*bad)
    with pytest.raises(TypeError):
        op2.Sparsity((di, di), [(mi, mi, None)], 0)
    s = op2.Sparsity((di, dd), [(m_iterset_toset, md, None)], name="foo")
    assert s.rcmaps[(0, 0)][0] == (m_iterset_toset, md) and s.dims[0][0] == (1, 1) and s.name == "foo" and s.dsets == (di, dd)
    s = op2.Sparsity((di, di), [(mi, mi, None), (mi, mi, None)], name="foo")             # duplicates are filtered
    assert s.rcmaps[(0, 0)] == [(mi, mi)] and s.dims[0][0] == (1, 1)
    pairs = ((m_iterset_toset, m_iterset_toset), (mi, mi))                                 # different iteration sets
    s = op2.Sparsity((di, di), [(*pairs[0], None), (*pairs[1], None)], name="foo")
    assert frozenset(s.rcmaps[(0, 0)]) == frozenset(pairs)
    s1 = op2.Sparsity((di, di), [(m_iterset_toset, m_iterset_toset, None), (mi, mi, None)])
    s2 = op2.Sparsity((di, di), [(mi, mi, None), (m_iterset_toset, m_iterset_toset, None)])
    assert s1.rcmaps[(0, 0)] == s2.rcmaps[(0, 0)]                                         # deterministic order
    with pytest.raises(RuntimeError):                                                     # different itersets in a pair
        op2.Sparsity((dd, di), [(md, mi, None)])
    with pytest.raises(RuntimeError):                                                     # row maps on another data set
        op2.Sparsity((di, di), [(mi, mi, None), (md, mi, None)])
    with pytest.raises(RuntimeError):
        op2.Sparsity((di, di), [(mi, mi, None), (mi, md, None)])
    s = op2.Sparsity((di, di), [(mi, mi, None)])
    assert s.shape == (1, 1) and [b for b in s] == [s] and s[0, 0] == s


def test_sparsity_blocks_of_mixed_spaces(ms):
    cols = ms.shape[1]
    assert ms.shape == (len(ms.dsets[0]), len(ms.dsets[1]))
    for i, block in enumerate(ms):
        assert block == ms[i // cols, i % cols]
    for i, rds in enumerate(ms.dsets[0]):
        for j, cds in enumerate(ms.dsets[1]):
            block = ms[i, j]
            assert block == ms[i][j]
            assert block.dsets == (rds, cds) and block.rcmaps[(0, 0)] == ms.rcmaps[(i, j)]
    assert op2.Mat(ms).dtype == op2.ScalarType


def test_sparsity_mixed_validation(m_iterset_toset, m_iterset_set, m_set_toset, m_set_set, mds):
    def build(rmm, cmm):
        return op2.Sparsity((mds, mds), {(i, j): [(rm, cm, None)] for i, rm in enumerate(rmm) for j, cm in enumerate(cmm)})
    with pytest.raises(RuntimeError):          # both maps of a pair must share the iteration set
        build(op2.MixedMap((m_iterset_toset, m_iterset_set)), op2.MixedMap((m_set_toset, m_set_set)))
    with pytest.raises(RuntimeError):          # row / column maps must map to the block's data sets
        build(op2.MixedMap((m_iterset_toset, m_iterset_set)), op2.MixedMap((m_set_toset, m_set_toset)))
    with pytest.raises(RuntimeError):
        build(op2.MixedMap((m_set_toset, m_set_toset)), op2.MixedMap((m_iterset_toset, m_iterset_set)))


def test_sparsity_and_mat_text(sparsity, mat, m_iterset_toset):
    r = "Sparsity(%r, %r, name=%r, nested=%r, block_sparse=%r, diagonal_block=%r)" % (
        sparsity.dsets, sparsity._maps_and_regions, sparsity.name, sparsity._nested, sparsity._block_sparse, sparsity._diagonal_block)
    assert repr(sparsity) == r
    s = "OP2 Sparsity: dsets %s, maps_and_regions %s, name %s, nested %s, block_sparse %s, diagonal_block %s" % (
        sparsity.dsets, sparsity._maps_and_regions, sparsity.name, sparsity._nested, sparsity._block_sparse, sparsity._diagonal_block)
    assert str(sparsity) == s
    assert repr(mat) == "Mat(%r, %r, %r)" % (mat.sparsity, mat.dtype, mat.name)
    assert str(mat) == "OP2 Mat: %s, sparsity (%s), datatype %s" % (mat.name, mat.sparsity, mat.dtype.name)


# ---- Mat (test_api.py:1138-1197) -------------------------------------------------------------------------------------
def test_mat_api(sparsity, mat, m_iterset_toset):
    with pytest.raises(TypeError):
        op2.Mat('illegalsparsity')
    with pytest.raises(exceptions.NameTypeError):
        op2.Mat(sparsity, name=2)
    assert mat.dtype == op2.ScalarType
    m = op2.Mat(sparsity, op2.ScalarType, 'bar')
    assert m.sparsity == sparsity and m.dtype == op2.ScalarType and m.name == 'bar'
    wrongmap = op2.Map(op2.Set(2), op2.Set(3), 2, [0, 0, 0, 0])
    with pytest.raises(exceptions.MapValueError):
        mat(op2.INC, (wrongmap, wrongmap))
    for mode in (op2.READ, op2.RW, op2.MIN, op2.MAX):
        with pytest.raises(exceptions.ModeValueError):
            mat(mode, (m_iterset_toset, m_iterset_toset))
    assert [b for b in mat] == [mat]


# ---- Global (test_api.py:1200-1303) ----------------------------------------------------------------------------------
def test_global_api():
    with pytest.raises(TypeError):
        op2.Global('illegaldim')
    with pytest.raises(TypeError):
        op2.Global((1, 'illegaldim'))
    with pytest.raises(exceptions.NameTypeError):
        op2.Global(1, 1, name=2)
    assert op2.Global(1, 1).dim == (1,) and op2.Global([2, 3], [1] * 6).dim == (2, 3)
    assert op2.Global(1, 1.0).dtype == np.asarray(1.0).dtype and op2.Global(1, 1).dtype == np.asarray(1).dtype
    assert op2.Global(1, 1, dtype=np.float64).dtype == np.float64
    assert op2.Global(1, 1.5, dtype=np.int64).dtype == np.int64
    with pytest.raises(exceptions.DataValueError):
        op2.Global(1, 'illegal_type', 'double')
    for dim in (1, (2, 2)):
        with pytest.raises(exceptions.DataValueError):
            op2.Global(dim, [1] * (int(np.prod(dim)) + 1))
    g = op2.Global((2, 2), [1.0] * 4)
    assert g.dim == (2, 2) and g.data.shape == (2, 2)
    g = op2.Global((2, 2), [1] * 4, 'double', 'bar')
    assert g.dim == (2, 2) and g.dtype == np.float64 and g.name == 'bar' and g.data.sum() == 4
    g = op2.Global(1, 1)
    g.data = 2
    assert g.data.sum() == 2
    with pytest.raises(exceptions.DataValueError):
        g.data = [1, 2]
    assert [x for x in g] == [g] and len(g) == 1
    g = op2.Global(1, 1, 'double')
    assert str(g) == "OP2 Global Argument: %s with dim %s and value %s" % (g.name, g.dim, g.data)
    for mode in (op2.RW, op2.WRITE):
        with pytest.raises(exceptions.ModeValueError):
            g(mode)


# ---- Map / MixedMap (test_api.py:1306-1521) --------------------------------------------------------------------------
def test_map_api(set_, iterset, toset, m_iterset_toset):
    with pytest.raises(exceptions.SetTypeError):
        op2.Map('illegalset', set_, 1, [])
    with pytest.raises(exceptions.SetTypeError):
        op2.Map(set_, 'illegalset', 1, [])
    with pytest.raises(exceptions.ArityTypeError):
        op2.Map(set_, set_, 'illegalarity', [])
    with pytest.raises(exceptions.ArityTypeError):
        op2.Map(set_, set_, (2, 2), [])
    with pytest.raises(exceptions.NameTypeError):
        op2.Map(set_, set_, 1, [], name=2)
    with pytest.raises(exceptions.DataValueError):
        op2.Map(set_, set_, 1, 'abcdefg')
    with pytest.raises(exceptions.DataValueError):
        op2.Map(iterset, toset, 1, [1] * (iterset.size + 1))
    m = op2.Map(iterset, toset, 1, [1.5] * iterset.size)                  # floats are converted to IntType
    assert m.values.dtype == op2.IntType and m.values.sum() == iterset.size
    m = op2.Map(iterset, toset, 2, [1] * 2 * iterset.size, 'bar')
    assert m.arity == 2 and m.values.shape == (iterset.size, 2)
    assert (m.iterset == iterset and m.toset == toset and m.arities == (2,) and m.arange == (0, 2)
            and m.values.sum() == 2 * iterset.size and m.name == 'bar' and m.split == (m,))
    mm = m_iterset_toset
    # Maps compare by identity: a copy with the same sets, arity and values is a different Map
    for mcopy in (op2.Map(mm.iterset, mm.toset, mm.arity, mm.values),
                  op2.Map(op2.Set(mm.iterset.size), mm.toset, mm.arity, mm.values),
                  op2.Map(mm.iterset, op2.Set(mm.toset.size), mm.arity, mm.values),
                  op2.Map(mm.iterset, mm.toset, mm.arity * 2, list(mm.values) * 2)):
        assert mm != mcopy and not mm == mcopy and mcopy == mcopy
    assert [x for x in mm] == [mm] and len(mm) == 1
    assert repr(mm) == "Map(%r, %r, %r, None, %r, %r, %r)" % (mm.iterset, mm.toset, mm.arity, mm.name, mm._offset, mm._offset_quotient)
    assert str(mm) == "OP2 Map: %s from (%s) to (%s) with arity %s" % (mm.name, mm.iterset, mm.toset, mm.arity)


def test_mixed_map_api(maps, mmap):
    with pytest.raises(TypeError):
        op2.MixedMap('illegalarg')
    assert mmap.split == maps and mmap.split[:-1] == tuple(mmap)[:-1]
    assert all(mmap.iterset == m.iterset for m in mmap)
    assert mmap.toset == op2.MixedSet(m.toset for m in mmap)
    assert mmap.arity == sum(m.arity for m in mmap) and mmap.arities == tuple(m.arity for m in mmap)
    assert mmap.arange == (0,) + tuple(np.cumsum(mmap.arities))
    assert all((v == m.values).all() for v, m in zip(mmap.values, mmap))
    assert all((v == m.values_with_halo).all() for v, m in zip(mmap.values_with_halo, mmap))
    assert mmap.name == tuple(m.name for m in mmap) and mmap.offset == tuple(m.offset for m in mmap)
    assert tuple(m for m in op2.MixedMap(maps)) == maps and len(op2.MixedMap(maps)) == len(maps)
    assert op2.MixedMap(maps) == op2.MixedMap(maps) and not op2.MixedMap(maps) != op2.MixedMap(maps)
    assert op2.MixedMap((maps[0], maps[1])) != op2.MixedMap((maps[1], maps[0]))
    assert op2.MixedMap(maps) != maps[0] and not op2.MixedMap(maps) == maps[0]
    assert repr(mmap) == "MixedMap(%r)" % (mmap.split,)
    assert str(mmap) == "OP2 MixedMap composed of Maps: %s" % (mmap.split,)


# ---- Kernel / ParLoop (test_api.py:1524-1617) ------------------------------------------------------------------------
def test_kernel_api():
    with pytest.raises(exceptions.NameTypeError):
        op2.Kernel("", name=2)
    assert op2.CStringLocalKernel("", "foo", accesses=(), dtypes=()).name == "foo"
    k = op2.Kernel("static int foo() { return 0; }", 'foo')
    assert str(k) == "OP2 Kernel: %s" % k.name


def test_parloop_validation(set_, dat, m_iterset_toset, sparsity):
    with pytest.raises(exceptions.KernelTypeError):
        op2.par_loop('illegal_kernel', set_, dat(op2.READ, m_iterset_toset))
    with pytest.raises(exceptions.SetTypeError):
        op2.par_loop(op2.Kernel("", "k"), 'illegal_set', dat(op2.READ, m_iterset_toset))
    set1, set2 = op2.Set(2), op2.Set(3)
    d = op2.Dat(op2.DataSet(set1, 1))
    with pytest.raises(exceptions.MapValueError):                          # the map's iterset is not the loop's
        op2.ParLoop(op2.Kernel("void k() { }", "k"), set1, d(op2.READ, op2.Map(set2, set1, 1, [0, 0, 0])))
    m = op2.Mat(sparsity)
    rmap, cmap = sparsity.rcmaps[(0, 0)][0]
    with pytest.raises(exceptions.MapValueError):
        op2.ParLoop(op2.Kernel("static void k() { }", "k"), set1, m(op2.INC, (rmap, cmap)))


def test_frozen_dats_cannot_use_different_access_mode():
    """test_api.py:1604-1616 (the loops are only constructed here: construction is where the check lives)."""
    s1, s2 = op2.Set(2), op2.Set(3)
    m = op2.Map(s1, s2, 3, [0] * 6)
    d = op2.Dat(s2 ** 1, [0] * 3, dtype=int)
    k = op2.Kernel("static void k(int64_t *x) {}", "k")
    with d.frozen_halo(op2.INC):
        op2.ParLoop(k, s1, d(op2.INC, m))
        with pytest.raises(RuntimeError):
            op2.ParLoop(k, s1, d(op2.WRITE, m))


@pytest.mark.gpu
def test_empty_map_and_iterset():
    """test_api.py:1594-1602: a loop over an empty set with an unpopulated map is legal and does nothing."""
    s1, s2 = op2.Set(0), op2.Set(10)
    m = op2.Map(s1, s2, 3)
    d = op2.Dat(s2 ** 1, [0] * 10, dtype=int)
    op2.par_loop(op2.Kernel("static void k(int64_t *x) {}", "k"), s1, d(op2.READ, m))
    assert (d.data_ro == 0).all()
