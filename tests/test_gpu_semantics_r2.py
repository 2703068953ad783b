"""GPU parity for the wrapper semantics added in round 2: zeroed MIN/MAX packs (builder.py:276-279, 368-371), Mats reached
through PermutedMaps (builder.py:144-176), and the one-buffer guarantee of ``dat.data`` views (pyop2/types/dat.py:145-204)."""
import numpy as np
import pytest

from firedrake_amd import op2
from helpers import oracle_run, structured_tri_mesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("zeroed", [False, True])
def test_zeroed_output_arguments_for_min_max(zeroed):
    rng = np.random.default_rng(8)
    it, ind = op2.Set(3000), op2.Set(1100)
    mp = op2.Map(it, ind, 2, rng.integers(0, 1100, size=(3000, 2)))
    b = op2.Dat(it, rng.standard_normal(3000))
    code = ("static void mz(double *a, double *g, const double *b) { for (int i = 0; i < 2; ++i) a[i] = a[i] < *b ? *b : a[i]; "
            "g[0] = g[0] < *b ? *b : g[0]; }")
    a, g = op2.Dat(ind, np.full(1100, -5.0)), op2.Global(1, -7.0)
    k = op2.Kernel(code, "mz", requires_zeroed_output_arguments=zeroed)
    ref = oracle_run(k, it, a(op2.MAX, mp), g(op2.MAX), b(op2.READ))
    op2.par_loop(k, it, a(op2.MAX, mp), g(op2.MAX), b(op2.READ))
    assert np.array_equal(a.data_ro, ref[0]) and np.array_equal(g.data_ro, ref[1])      # max is order independent
    assert g.data_ro[0] == max(-7.0, max(0.0, b.data_ro.max()) if zeroed else b.data_ro.max())


@pytest.mark.parametrize("scatter", ["table", "search"])
def test_mat_through_permuted_maps(scatter, monkeypatch):
    from firedrake_amd.configuration import configuration
    monkeypatch.setitem(configuration, "mat_scatter", scatter)
    coords, cells = structured_tri_mesh(24, 17, perturb=0.2)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    pr, pc = op2.PermutedMap(m, [2, 0, 1]), op2.PermutedMap(m, [1, 2, 0])
    x = op2.Dat(nodes ** 2, coords)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    km = op2.Kernel("static void kpm(double *A, const double *x) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) "
                    "A[i*3+j] += (i+1)*x[2*i] + 10*(j+1)*x[2*j+1]; }", "kpm")
    ref = oracle_run(km, ele, mat(op2.INC, (pr, pc)), x(op2.READ, m))[0]
    op2.par_loop(km, ele, mat(op2.INC, (pr, pc)), x(op2.READ, m))
    _, _, v = mat.csr()
    assert np.abs(v - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


def test_held_host_views_stay_coherent():
    """A view obtained from ``dat.data`` BEFORE a parloop keeps aliasing the Dat: it shows what the device wrote, and a
    write through it afterwards is seen by the next parloop (the reference has one buffer; here the mirror is kept
    coherent while such a view is alive, op2types._Mirrored)."""
    n = 1000
    s = op2.Set(n)
    d, e = op2.Dat(s, np.arange(n, dtype=np.float64)), op2.Dat(s)
    v = d.data                                   # writable view, held across the loops below
    ro = e.data_ro                               # read-only view of the output, held as well
    twice = op2.Kernel("static void tw(double *o, const double *i) { o[0] = 2.0*i[0]; }", "tw")
    inc = op2.Kernel("static void pl(double *x) { x[0] += 1.0; }", "pl")
    op2.par_loop(twice, s, e(op2.WRITE), d(op2.READ))
    assert np.array_equal(e.data_ro, 2.0 * np.arange(n)) and np.array_equal(ro, 2.0 * np.arange(n))
    op2.par_loop(inc, s, d(op2.RW))              # the device writes d ...
    assert np.array_equal(v, np.arange(n) + 1.0)  # ... and the held view shows it
    v[:] = 7.0                                   # host write through the OLD view, no new .data access
    op2.par_loop(twice, s, e(op2.WRITE), d(op2.READ))
    assert np.array_equal(e.data_ro, np.full(n, 14.0)) and np.array_equal(ro, np.full(n, 14.0))
    g = op2.Global(1, 1.5)
    gv = g.data
    op2.par_loop(op2.Kernel("static void sc(double *o, const double *g) { o[0] *= g[0]; }", "sc"), s, e(op2.RW), g(op2.READ))
    gv[0] = 2.0
    op2.par_loop(op2.Kernel("static void sc(double *o, const double *g) { o[0] *= g[0]; }", "sc"), s, e(op2.RW), g(op2.READ))
    assert np.array_equal(e.data_ro, np.full(n, 14.0 * 1.5 * 2.0))
    del v, gv, ro
    op2.par_loop(inc, s, d(op2.RW))
    assert not d._rw_handed                      # last view gone: back to lazy mirroring


def test_plan_that_does_not_fit_demotes_the_wrapper(monkeypatch):
    """A plan that exceeds LDS / builder capacity must not fail the loop: ocr -> staged (matrix plans) -> direct, decided
    before anything is launched (Parloop._ensure_geometry); results unchanged."""
    from firedrake_amd import forms, mesh as fmesh
    from firedrake_amd.parloop import Parloop, PlanDoesNotFit
    from test_gpu_forms import _oracle_problem
    m = fmesh.UnitCubeMesh(8, degrees=(1,), tile=(4, 4, 2), perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    ro, Ao = _oracle_problem(prob, True)
    real_ocr, real_staged = Parloop._ocr_geometry, Parloop._staged_geometry

    def no_ocr(self, *a, **k):
        raise PlanDoesNotFit("test: owner-computes-rows plan does not fit")
    monkeypatch.setattr(Parloop, "_ocr_geometry", no_ocr)
    A = prob.assemble_jacobian().toscipy()
    assert prob.jacobian()[1]._prepared["cw"].src.mode.startswith("staged")
    assert np.abs(A.data - Ao.data).max() <= 1e-12 * np.abs(Ao.data).max()

    def no_staged(self, *a, **k):
        raise PlanDoesNotFit("test: staged plan does not fit")
    monkeypatch.setattr(Parloop, "_staged_geometry", no_staged)
    prob2 = forms.PoissonProblem(m, 1, bcs=True)
    r = prob2.assemble_residual()
    A2 = prob2.assemble_jacobian().toscipy()
    assert prob2.res_loop._prepared["cw"].src.mode == "direct" and prob2.jacobian()[1]._prepared["cw"].src.mode == "direct"
    assert np.abs(r.data_ro - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max())
    assert np.abs(A2.data - Ao.data).max() <= 1e-12 * np.abs(Ao.data).max()
    monkeypatch.setattr(Parloop, "_ocr_geometry", real_ocr)
    monkeypatch.setattr(Parloop, "_staged_geometry", real_staged)
