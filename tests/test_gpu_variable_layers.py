"""GPU: loops over extruded sets with VARIABLE layers (set.py:326-337; builder.py:754-831) through the wrappers the backend picks
since round 5 -- staged (Dat loops) and owner-computes-rows (Mat loops) over the ragged space of existing cells -- and through the
direct wrapper, against the oracle."""
import numpy as np
import pytest

from firedrake_amd import op2
from firedrake_amd.configuration import configuration
from helpers import oracle_run

pytestmark = pytest.mark.gpu


def _columns(rng, nb, L):
    bot = rng.integers(0, 3, nb)
    top = bot + 2 + rng.integers(0, L - 4, nb)
    return bot, top


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP])
@pytest.mark.parametrize("subset", [False, True])
def test_dat_loop_over_variable_layers(mode, region, subset, monkeypatch):
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(4)
    nbase, L, nv = 3000, 9, 900
    base = op2.Set(nbase)
    bot, top = _columns(rng, nbase, L)
    ext = op2.ExtrudedSet(base, layers=np.stack([bot, top], axis=1))
    nodes = op2.Set(nv * L)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * L + bot[:, None], tri * L + bot[:, None] + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nv * L, 2)))
    w = op2.Dat(base, rng.standard_normal(nbase))
    out = op2.Dat(nodes)
    k = op2.Kernel("static void ksv(double *o, const double *x, const double *w, int layer) "
                   "{ for (int i = 0; i < 6; ++i) o[i] += (1 + layer) * w[0] * (x[2*i] + 0.5*x[2*i+1]); }", "ksv")
    it = op2.Subset(ext, rng.choice(nbase, 2000, replace=False)) if subset else ext
    args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, it, *args, iteration_region=region, pass_layer_arg=True)
    for _ in range(2):
        out.zero()
        pl()
    assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
    got = np.array(out.data_ro)
    out.zero()                                    # (the oracle starts from the carrier's current values)
    ref = oracle_run(k, it, *args, iteration_region=region, pass_layer_arg=True)[0]
    assert np.abs(ref).max() > 0 and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("region", [None, op2.ON_TOP])
@pytest.mark.parametrize("sliced", [False, True])
def test_matrix_loop_over_variable_layers(mode, region, sliced, monkeypatch):
    monkeypatch.setitem(configuration, "mode", mode)
    monkeypatch.setitem(configuration, "ocr_sliced_min_arity", 4 if sliced else 8)
    rng = np.random.default_rng(9)
    nb, L = 600, 10
    base = op2.Set(nb)
    bot, top = _columns(rng, nb, L)
    ext = op2.ExtrudedSet(base, layers=np.stack([bot, top], axis=1))
    nodes = op2.Set((nb + 2) * L)
    vm = np.array([[i * L + bot[i], i * L + bot[i] + 1, (i + 1) * L + bot[i], (i + 1) * L + bot[i] + 1, (i + 2) * L + bot[i], (i + 2) * L + bot[i] + 1]
                   for i in range(nb)], dtype=np.int32)
    m = op2.Map(ext, nodes, 6, vm, offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.uniform(0, 1, (nodes.total_size, 2)), np.float64)
    w = op2.Dat(ext, rng.uniform(1, 2, nb), np.float64)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.Kernel("""
static void prismv(double *A, const double *x, const double *w, int layer)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      A[i*6 + j] += w[0] * (x[2*i] + 2.0 * x[2*j + 1]) + 0.25 * layer + (i == j ? 1.0 : 0.0);
}""", "prismv")
    kw = dict(iteration_region=region, pass_layer_arg=True)
    pl = op2.LegacyParloop(k, ext, mat(op2.INC, (m, m)), x(op2.READ, m), w(op2.READ), **kw)
    for _ in range(2):
        mat.zero()
        pl()
    got_mode = pl._prepare()["cw"].src.mode
    assert got_mode.startswith(("ocrs" if sliced else "ocr") if mode == "auto" else "direct")
    ref = oracle_run(k, ext, mat(op2.INC, (m, m)), x(op2.READ, m), w(op2.READ), **kw)[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    assert np.abs(v - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("region", [None, op2.ON_TOP])
def test_periodic_columns_through_the_fast_shapes(mode, region, monkeypatch):
    """Periodic extrusion (builder.py:101-123): the wrap folded into the derived map of the (column, layer) cells -- staged Dat loop and
    owner-computes-rows Mat loop -- and through the direct wrapper, against the oracle."""
    from mixed_cases import periodic_column_mesh
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(31)
    base, ext, nodes, cm = periodic_column_mesh(rng, nbase=2500, ncl=6, nv=900)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.standard_normal(base.size))
    out = op2.Dat(nodes)
    k = op2.Kernel("static void kpw(double *o, const double *x, const double *w, int layer) { for (int i = 0; i < 6; ++i) "
                   "o[i] += (1 + layer) * w[0] * ((i+1)*x[2*i] + 0.5*x[2*i+1]); }", "kpw")
    kw = dict(iteration_region=region, pass_layer_arg=True)
    args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, ext, *args, **kw)
    for _ in range(2):
        out.zero()
        pl()
    assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
    got = np.array(out.data_ro)
    out.zero()
    ref = oracle_run(k, ext, *args, **kw)[0]
    assert np.abs(ref).max() > 0 and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [region] if region is not None else None)]))
    km = op2.Kernel("static void kpm2(double *A, const double *x, const double *w) { for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) "
                    "A[i*6+j] += w[0]*x[2*i]*x[2*j+1] + (i == j); }", "kpm2")
    margs = (mat(op2.INC, (cm, cm)), x(op2.READ, cm), w(op2.READ))
    plm = op2.LegacyParloop(km, ext, *margs, iteration_region=region)
    for _ in range(2):
        mat.zero()
        plm()
    assert plm._prepare()["cw"].src.mode.startswith("ocr" if mode == "auto" else "direct")
    mref = oracle_run(km, ext, *margs, iteration_region=region)[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, mref.rowptr) and np.array_equal(ci, mref.colidx)
    assert np.abs(v - mref.values).max() <= 1e-12 * np.abs(mref.values).max()


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("periodic", [False, True])
def test_interior_facets_of_extruded_columns(mode, periodic, monkeypatch):
    """ON_INTERIOR_FACETS (dS_h): the kernel sees the cell below and the cell above a horizontal facet (builder.py:94-124, 806-809);
    the staged wrapper takes both cells' nodes from one row of the derived map, also across the seam of a periodic column."""
    from mixed_cases import periodic_column_mesh
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(51)
    ncl, nb, nv = 7, 2500, 900
    if periodic:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=nb, ncl=ncl, nv=nv)
    else:
        base = op2.Set(nb)
        ext = op2.ExtrudedSet(base, layers=ncl + 1)
        nodes = op2.Set(nv * (ncl + 1))
        tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nb)])
        cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (ncl + 1), tri * (ncl + 1) + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.standard_normal(base.size))
    out = op2.Dat(nodes)
    k = op2.Kernel("static void kfac(double *o, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                   "o[i] += (1 + layer) * w[0] * ((i+1)*x[2*i] + 0.5*x[2*((i+5)%12)+1]); }", "kfac")
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS, pass_layer_arg=True)
    for it in (ext, op2.Subset(ext, rng.choice(nb, 1800, replace=False))):
        args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
        out.zero()
        ref = oracle_run(k, it, *args, **kw)[0]
        pl = op2.LegacyParloop(k, it, *args, **kw)
        for _ in range(2):
            out.zero()
            pl()
        assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
        assert np.abs(ref).max() > 0 and np.abs(out.data_ro - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("periodic", [False, True])
def test_interior_facet_matrix_loop(mode, periodic, monkeypatch):
    """The matrix of an interior-facet integral on an extruded set (dS_h: 2a x 2a element tensors coupling the cells below and above
    a horizontal facet): row-sliced owner-computes-rows on the derived facet maps (twice the arity), and the direct wrapper."""
    from mixed_cases import periodic_column_mesh
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(61)
    ncl, nb, nv = 6, 2000, 700
    if periodic:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=nb, ncl=ncl, nv=nv)
    else:
        base = op2.Set(nb)
        ext = op2.ExtrudedSet(base, layers=ncl + 1)
        nodes = op2.Set(nv * (ncl + 1))
        tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nb)])
        cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (ncl + 1), tri * (ncl + 1) + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.uniform(1, 2, base.size))
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [op2.ON_INTERIOR_FACETS])]))
    k = op2.Kernel("static void kfm(double *A, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                   "for (int j = 0; j < 12; ++j) A[i*12 + j] += w[0] * (x[2*i] * x[2*j+1] + 0.125 * layer) + (i == j ? 1.0 : 0.0) + (i < 6 && j >= 6 ? 0.5 : 0.0); }", "kfm")
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS, pass_layer_arg=True)
    args = (mat(op2.INC, (cm, cm)), x(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, ext, *args, **kw)
    for _ in range(2):
        mat.zero()
        pl()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs" if mode == "auto" else "direct")
    ref = oracle_run(k, ext, *args, **kw)[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    assert np.abs(v - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("subset", [False, True])
def test_interior_facets_of_variable_layer_columns(mode, subset, monkeypatch):
    """Round 6: ON_INTERIOR_FACETS on columns of DIFFERENT heights (builder.py:754-776 bounds per entity, :806-809 one trip per pair of
    stacked cells) through the staged wrapper (Dat loop) and the row-sliced owner-computes-rows wrapper (Mat loop) -- the ragged
    virtual space holds top_e - bottom_e - 2 facets per column, a derived row both cells' nodes -- and through the direct wrapper."""
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(77)
    nb, L, nv = 2500, 9, 800
    base = op2.Set(nb)
    bot, top = _columns(rng, nb, L)
    top[:40] = bot[:40] + 2                                   # one-cell columns: no interior facet
    ext = op2.ExtrudedSet(base, layers=np.stack([bot, top], axis=1))
    nodes = op2.Set(nv * L)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nb)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * L + bot[:, None], tri * L + bot[:, None] + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nv * L, 2)))
    w = op2.Dat(base, rng.uniform(1, 2, nb))
    out = op2.Dat(nodes)
    it = op2.Subset(ext, rng.choice(nb, 1700, replace=False)) if subset else ext
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS, pass_layer_arg=True)
    k = op2.Kernel("static void kfv(double *o, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                   "o[i] += (1 + layer) * w[0] * ((i+1)*x[2*i] + 0.5*x[2*((i+5)%12)+1]); }", "kfv")
    args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
    out.zero()
    ref = oracle_run(k, it, *args, **kw)[0]
    pl = op2.LegacyParloop(k, it, *args, **kw)
    for _ in range(2):
        out.zero()
        pl()
    assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
    assert np.abs(ref).max() > 0 and np.abs(out.data_ro - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    if subset:
        return
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [op2.ON_INTERIOR_FACETS])]))
    km = op2.Kernel("static void kfvm(double *A, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                    "for (int j = 0; j < 12; ++j) A[i*12 + j] += w[0] * (x[2*i] * x[2*j+1] + 0.125 * layer) + (i == j ? 1.0 : 0.0) + (i < 6 && j >= 6 ? 0.5 : 0.0); }", "kfvm")
    margs = (mat(op2.INC, (cm, cm)), x(op2.READ, cm), w(op2.READ))
    plm = op2.LegacyParloop(km, ext, *margs, **kw)
    for _ in range(2):
        mat.zero()
        plm()
    assert plm._prepare()["cw"].src.mode.startswith("ocrs" if mode == "auto" else "direct")
    mref = oracle_run(km, ext, *margs, **kw)[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, mref.rowptr) and np.array_equal(ci, mref.colidx)
    assert np.abs(v - mref.values).max() <= 1e-12 * np.abs(mref.values).max()


def test_loops_without_any_cell_are_no_ops():
    """One cell layer has no interior facet: the staged wrapper's virtual space is empty and nothing is launched (the reference's
    wrapper loop runs zero trips, builder.py:806-809)."""
    rng = np.random.default_rng(5)
    base = op2.Set(50)
    ext = op2.ExtrudedSet(base, layers=2)
    nodes = op2.Set(40 * 2)
    tri = np.array([rng.choice(40, 3, replace=False) for _ in range(50)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * 2, tri * 2 + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    out = op2.Dat(nodes)
    k = op2.Kernel("static void kf0(double *o, const double *x) { for (int i = 0; i < 12; ++i) o[i] += x[2*i]; }", "kf0")
    pl = op2.LegacyParloop(k, ext, out(op2.INC, cm), x(op2.READ, cm), iteration_region=op2.ON_INTERIOR_FACETS)
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("staged") and not np.asarray(out.data_ro).any()
