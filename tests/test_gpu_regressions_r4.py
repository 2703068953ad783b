"""GPU regressions for the round-3 advisor findings."""
import numpy as np
import pytest

from firedrake_amd import forms, mesh as fmesh, op2
from firedrake_amd.configuration import configuration
from helpers import oracle_run

pytestmark = pytest.mark.gpu


def test_locality_retry_exhausted_keeps_order_and_rows_together(monkeypatch):
    """Parloop._staged_geometry: when every tile size of the derived entity order exceeds the LDS budget, the uniform fallback
    runs on the map rows gathered in the LAST order tried -- the order table handed to the kernel must be that same order
    (direct arguments are indexed through it)."""
    monkeypatch.setitem(configuration, "lds_limit", 2500)
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(10, degrees=(1,), perturb=0.1, numbering="random")
    V = m.space(1)
    cm = V.cell_node_map
    rng = np.random.default_rng(0)
    w = op2.Dat(m.cell_set, rng.uniform(0.5, 2.0, m.cell_set.total_size))       # a DIRECT argument: indexed by the entity id
    u = op2.Dat(V.node_set, rng.standard_normal(V.node_set.total_size))
    k = op2.Kernel("""
static void wsum(double *r, const double *x, const double *u, const double *w)
{
  for (int i = 0; i < 4; ++i) r[i] += w[0] * (u[i] + x[3*i] - 2.0*x[3*i+2]);
}""", "wsum")
    r = op2.Dat(V.node_set)
    args = lambda out: (out(op2.INC, cm), m.coordinates(op2.READ, m.coord_space.cell_node_map), u(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, m.cell_set, *args(r))
    pl.compute()
    geo = next(iter(g for key, g in pl._prepared["parts"].items() if isinstance(g, dict) and "epb" in g))
    assert geo["order"] is not None                       # the derived order was kept (its blocks touch fewer nodes)
    ref = oracle_run(k, m.cell_set, *args(op2.Dat(V.node_set)))[0]
    assert np.abs(r.data_ro - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
