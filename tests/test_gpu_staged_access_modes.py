"""GPU: WRITE / MIN / MAX through a map in loops whose READ arguments are staged (builder.py:405-429: the DatPack of every access
mode): the backend-picked staged wrapper -- READ rows through LDS, the lane addressing the WRITE / MIN / MAX argument in global
memory -- and the direct wrapper, against the oracle."""
import numpy as np
import pytest

from firedrake_amd import op2
from firedrake_amd.configuration import configuration
from helpers import oracle_run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("subset", [False, True])
def test_write_min_max_through_maps(mode, subset, monkeypatch):
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(41)
    nc, nn = 60000, 21000
    cells, nodes = op2.Set(nc), op2.Set(nn)
    cm = op2.Map(cells, nodes, 3, np.array([rng.choice(nn, 3, replace=False) for _ in range(nc)], dtype=np.int32))
    x = op2.Dat(nodes ** 2, rng.standard_normal((nn, 2)))
    cv = op2.Dat(cells, rng.standard_normal(nc))
    it = op2.Subset(cells, rng.choice(nc, 45000, replace=False)) if subset else cells
    out = op2.Dat(nodes, np.full(nn, -7.0))
    kw = op2.Kernel("static void interp(double *o, const double *x) { for (int i = 0; i < 3; ++i) o[i] = 2.0*x[2*i] - x[2*i+1]*x[2*i+1]; }", "interp")
    ref = oracle_run(kw, it, out(op2.WRITE, cm), x(op2.READ, cm))[0]
    pl = op2.LegacyParloop(kw, it, out(op2.WRITE, cm), x(op2.READ, cm))
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
    assert np.allclose(out.data_ro, ref, rtol=1e-14, atol=0) and (ref != -7.0).sum() > 1000
    lo, hi = op2.Dat(nodes, np.full(nn, 1e30)), op2.Dat(nodes, np.full(nn, -1e30))
    km = op2.Kernel("static void bounds(double *lo, double *hi, const double *c, const double *x) { for (int i = 0; i < 3; ++i) { "
                    "const double v = c[0] + 0.1*x[2*i]; if (v < lo[i]) lo[i] = v; if (v > hi[i]) hi[i] = v; } }", "bounds")
    args = (lo(op2.MIN, cm), hi(op2.MAX, cm), cv(op2.READ), x(op2.READ, cm))
    refs = oracle_run(km, it, *args)
    plm = op2.LegacyParloop(km, it, *args)
    plm()
    assert plm._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
    assert np.allclose(lo.data_ro, refs[0], rtol=1e-14, atol=0) and np.allclose(hi.data_ro, refs[1], rtol=1e-14, atol=0)
    assert (refs[0] < 1e29).sum() > 1000


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("periodic", [False, True])
def test_write_and_max_through_maps_on_extruded_columns(mode, periodic, monkeypatch):
    """WRITE / MAX through the (column, layer) addressing of an extruded set (builder.py:94-124), beside staged READ arguments: the
    lane applies map + offset * layer (with the periodic wrap) to the base entity's map row."""
    from mixed_cases import periodic_column_mesh
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(71)
    ncl, nb, nv = 6, 3000, 1000
    if periodic:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=nb, ncl=ncl, nv=nv)
    else:
        base = op2.Set(nb)
        ext = op2.ExtrudedSet(base, layers=ncl + 1)
        nodes = op2.Set(nv * (ncl + 1))
        tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nb)])
        cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (ncl + 1), tri * (ncl + 1) + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    cv = op2.Dat(base, rng.standard_normal(base.size))
    k = op2.Kernel("static void interp_max(double *o, double *hi, const double *x, const double *c) { for (int i = 0; i < 6; ++i) { "
                   "o[i] = 2.0*x[2*i] - x[2*i+1]*x[2*i+1]; const double v = c[0] + x[2*i]; if (v > hi[i]) hi[i] = v; } }", "interp_max")
    for region in (None, op2.ON_TOP):
        out = op2.Dat(nodes, np.full(nodes.size, -7.0))
        hi = op2.Dat(nodes, np.full(nodes.size, -1e30))
        args = (out(op2.WRITE, cm), hi(op2.MAX, cm), x(op2.READ, cm), cv(op2.READ))
        refs = oracle_run(k, ext, *args, iteration_region=region)
        pl = op2.LegacyParloop(k, ext, *args, iteration_region=region)
        pl()
        assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
        assert np.allclose(out.data_ro, refs[0], rtol=1e-14, atol=0) and np.allclose(hi.data_ro, refs[1], rtol=1e-14, atol=0)
        assert (refs[0] != -7.0).sum() > 1000


@pytest.mark.parametrize("mode", ["auto", "direct"])
@pytest.mark.parametrize("shape", ["variable", "facets", "variable-facets"])
def test_write_and_max_through_maps_on_variable_layers_and_interior_facets(mode, shape, monkeypatch):
    """Round 6: the last two extruded shapes on which a WRITE / MAX argument through a map demoted the loop to the direct wrapper --
    columns of variable height and interior facets -- now keep the staged wrapper: the READ arguments come from LDS, the lane
    addresses the others itself from the column's own bottom (builder.py:754-776) and over both stacked cells (builder.py:94-124)."""
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(83)
    nb, L, nv = 3000, 8, 900
    base = op2.Set(nb)
    if shape.startswith("variable"):
        bot = rng.integers(0, 3, nb)
        top = bot + 2 + rng.integers(0, L - 3, nb)
        layers = np.stack([bot, top], axis=1)
    else:
        bot = np.zeros(nb, dtype=np.int64)
        layers = L
    ext = op2.ExtrudedSet(base, layers=layers)
    nodes = op2.Set(nv * (L + 2))
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nb)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (L + 2) + bot[:, None], tri * (L + 2) + bot[:, None] + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    cv = op2.Dat(base, rng.standard_normal(nb))
    facets = shape.endswith("facets")
    n = 12 if facets else 6
    k = op2.Kernel(f"static void interp_max_v(double *o, double *hi, const double *x, const double *c) {{ for (int i = 0; i < {n}; ++i) {{ "
                   "o[i] = 2.0*x[2*i] - x[2*i+1]*x[2*i+1]; const double v = c[0] + x[2*i]; if (v > hi[i]) hi[i] = v; } }", "interp_max_v")
    region = op2.ON_INTERIOR_FACETS if facets else None
    for it in (ext, op2.Subset(ext, rng.choice(nb, 2000, replace=False))):
        out = op2.Dat(nodes, np.full(nodes.size, -7.0))
        hi = op2.Dat(nodes, np.full(nodes.size, -1e30))
        args = (out(op2.WRITE, cm), hi(op2.MAX, cm), x(op2.READ, cm), cv(op2.READ))
        refs = oracle_run(k, it, *args, iteration_region=region)
        pl = op2.LegacyParloop(k, it, *args, iteration_region=region)
        pl()
        assert pl._prepare()["cw"].src.mode.startswith("staged" if mode == "auto" else "direct")
        assert np.allclose(out.data_ro, refs[0], rtol=1e-14, atol=0) and np.allclose(hi.data_ro, refs[1], rtol=1e-14, atol=0)
        assert (refs[0] != -7.0).sum() > 1000
