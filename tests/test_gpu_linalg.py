"""GPU: the reference's Dat arithmetic on the device.  Mirrors tests/pyop2/test_linalg.py:80-340 (same data, same
assertions) -- every operator is a direct parloop with a generated kernel (pyop2/types/dat.py:354-620), here run by
the HIP backend -- plus Dat.zero(subset)/copy(subset) (dat.py:297-336) and larger random comparisons with numpy."""
import numpy as np
import pytest

from firedrake_amd import op2

pytestmark = pytest.mark.gpu
nelems = 8


@pytest.fixture
def dset():
    return op2.DataSet(op2.Set(nelems), 1)


@pytest.fixture
def x(dset):
    return op2.Dat(dset, None, np.float64, "x")


@pytest.fixture
def y(dset):
    return op2.Dat(dset, np.arange(1, nelems + 1), np.float64, "y")


@pytest.fixture
def yi(dset):
    return op2.Dat(dset, np.arange(1, nelems + 1), np.int64, "y")


@pytest.fixture
def x2():
    return op2.Dat(op2.Set(nelems, "s1") ** (1, 2), np.zeros(2 * nelems), np.float64, "x")


@pytest.fixture
def y2():
    return op2.Dat(op2.Set(nelems, "s2") ** (2, 1), np.zeros(2 * nelems), np.float64, "y")


def test_binary_ops(x, y):
    x.data[:] = 2 * y.data
    assert all((x + y).data == 3 * y.data)
    assert all((x - y).data == y.data)
    assert all((x * y).data == 2 * y.data * y.data)
    assert all((x / y).data == 2.0)


@pytest.mark.parametrize("op", ["__add__", "__sub__", "__mul__", "__truediv__", "__iadd__", "__isub__", "__imul__", "__itruediv__"])
def test_shape_mismatch(x2, y2, op):
    with pytest.raises(ValueError):
        getattr(x2, op)(y2)


def test_scalar_ops(x, y):
    assert all((y + 1.0).data == y.data + 1.0)
    assert all((1.0 + y).data == y.data + 1.0)
    assert all((y - 1.0).data == y.data - 1.0)
    assert all((1.0 - y).data == 1.0 - y.data)
    assert all((y * 2.0).data == 2 * y.data)
    assert all((2.0 * y).data == 2 * y.data)
    assert all((y / 2.0).data == y.data / 2)


def test_pos_neg_copy(y):
    z = +y
    assert all(z.data == y.data) and z is not y
    z = -y
    assert all(z.data == -y.data) and z is not y


def test_result_types(y, yi):
    for r in (y + yi, y - yi, y * yi, y / yi):
        assert r.data.dtype == np.float64
    for r in (yi + y, yi - y, yi * y, yi / y):
        assert r.data.dtype == np.int64
    assert all((yi * y).data == np.arange(1, nelems + 1) ** 2)


def test_linalg_and_parloop(x, y):
    k = op2.Kernel('static void k(double *x) { *x = 1.0; }', 'k')
    op2.par_loop(k, x.dataset.set, x(op2.WRITE))
    z = x + y
    assert all(z.data == y.data + 1)


def test_inplace_ops(x, y, yi):
    x.data[:] = 2 * y.data
    x += y
    assert all(x.data == 3 * y.data)
    x -= y
    assert all(x.data == 2 * y.data)
    x *= y
    assert all(x.data == 2 * y.data * y.data)
    x /= y
    assert all(x.data == 2 * y.data)
    x += 1.0
    x -= 1.0
    x *= 2.0
    x /= 4.0
    assert all(x.data == y.data)
    x += x
    assert all(x.data == 2 * y.data)
    y += yi
    assert y.data.dtype == np.float64 and all(y.data == 2 * np.arange(1, nelems + 1))
    yi += y
    assert yi.data.dtype == np.int64 and all(yi.data == 3 * np.arange(1, nelems + 1))


def test_norm_inner():
    s = op2.Set(2)
    n = op2.Dat(s, [3, 4], np.float64, "n")
    o = op2.Dat(s, [4, 5], np.float64)
    assert abs(n.norm - 5) < 1e-12
    assert abs(n.inner(o) - 32) < 1e-12 and abs(o.inner(n) - 32) < 1e-12
    s1 = op2.Set(1)
    md = op2.MixedDat([op2.Dat(s1, [3], np.float64), op2.Dat(s1, [4], np.float64)])
    md1 = op2.MixedDat([op2.Dat(s1, [4], np.float64), op2.Dat(s1, [5], np.float64)])
    assert abs(md.norm - 5) < 1e-12
    assert abs(md.inner(md1) - 32) < 1e-12


def test_large_vector_dats_against_numpy():
    rng = np.random.default_rng(2)
    s = op2.Set((90000, 100000, 100003))          # core/owned/ghost: arithmetic touches the owned rows only
    a = op2.Dat(s ** 3, rng.standard_normal((100003, 3)))
    b = op2.Dat(s ** 3, rng.standard_normal((100003, 3)) + 3.0)
    A, B = a.data_ro_with_halos.copy(), b.data_ro_with_halos.copy()
    n = s.size
    close = lambda u, v: np.abs(u - v).max() <= 4e-16 * max(1.0, np.abs(v).max())     # fused multiply-adds allowed
    c = a * b - a / b
    assert close(c.data_ro, A[:n] * B[:n] - A[:n] / B[:n])
    a.axpy(0.75, b)
    assert close(a.data_ro, 0.75 * B[:n] + A[:n])
    assert np.array_equal(a.data_ro_with_halos[n:], A[n:])
    a.maxpy([2.0, -1.0], [b, c])
    assert close(a.data_ro, -1.0 * c.data_ro + (2.0 * B[:n] + (0.75 * B[:n] + A[:n])))
    ip = b.inner(c)
    exact = float(np.sum(B[:n] * c.data_ro))
    assert abs(ip - exact) <= 1e-11 * abs(exact)
    assert abs(b.norm - np.linalg.norm(B[:n])) <= 1e-12 * np.linalg.norm(B[:n])


def test_zero_and_copy_subsets():
    rng = np.random.default_rng(5)
    s = op2.Set(1000)
    d = op2.Dat(s ** 2, rng.standard_normal((1000, 2)))
    ref = d.data_ro.copy()
    ss = op2.Subset(s, np.arange(3, 1000, 7))
    d.zero(subset=ss)                                  # bc.zero (firedrake/bcs.py:192-221)
    ref[ss.indices] = 0
    assert np.array_equal(d.data_ro, ref)
    e = op2.Dat(s ** 2)
    d.copy(e, subset=op2.Subset(s, np.arange(0, 1000, 3)))
    exp = np.zeros_like(ref)
    exp[::3] = ref[::3]
    assert np.array_equal(e.data_ro, exp)
    d.copy(e)
    assert np.array_equal(e.data_ro, ref)
    with pytest.raises(op2.MapValueError):
        d.zero(subset=op2.Subset(op2.Set(1000), [1]))
