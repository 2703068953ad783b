"""Dual-evaluation (interpolation) parloops -- firedrake/interpolation.py:1087-1167 restated for C-string expressions.
CPU: generated kernel through the host-sim and the oracle against the expression evaluated with numpy at the node
positions.  GPU: the same through the HIP backend, and the DG-advection demo's ``interpolate`` statements."""
import numpy as np
import pytest

from firedrake_amd import mesh as fmesh, op2
from firedrake_amd.interpolation import Interpolator, Space, simplex_lagrange, simplex_node_points
from helpers import oracle_run


def _tri_case(n=5):
    m = fmesh.UnitSquareMesh(n, n - 1, degrees=(1, 2), perturb=0.1)
    V1, V2 = m.space(1), m.space(2)
    xs = Space(m.coord_space.cell_node_map, simplex_node_points(2, 1), simplex_lagrange(2, 1), 2)
    s1 = Space(V1.cell_node_map, simplex_node_points(2, 1), simplex_lagrange(2, 1), 1)
    s2 = Space(V2.cell_node_map, simplex_node_points(2, 2), simplex_lagrange(2, 2), 1)
    return m, V1, V2, xs, s1, s2


def _affine_node_positions(m, V, dim, degree):
    """Physical position of every node of V on the affine (P1-coordinate) mesh: the image of its reference point."""
    N = simplex_lagrange(dim, 1)(simplex_node_points(dim, degree))                 # (nk, dim+1)
    x = np.asarray(m.coordinates.data_ro_with_halos)
    cx = x[m.coord_space.cell_node_map.values_with_halo]                            # (ncell, dim+1, gdim)
    pos = np.zeros((V.node_set.total_size, dim))
    pos[V.cell_node_map.values_with_halo] = np.einsum("kv,cvd->ckd", N, cx)
    return pos


def _loop(it, out):
    it.interpolate  # noqa: B018  (attribute exists)
    tmap = it.target.node_map
    args = [out(op2.WRITE, tmap), it.coordinates(op2.READ, it.coord_space.node_map)]
    args += [d(op2.READ, sp.node_map) for d, sp in it.coefficients]
    args += [g(op2.READ) for g in it.constants]
    return tmap.iterset, args


def test_expression_into_p2_on_host():
    from hostsim import run_direct
    m, V1, V2, xs, s1, s2 = _tri_case()
    it = Interpolator(["sin(3.0*X[0])*X[1] + c0[0]"], s2, m.coordinates, xs, constants=[op2.Global(1, 0.25)])
    out = V2.dat()
    iterset, args = _loop(it, out)
    got = run_direct(op2.LegacyParloop(it.kernel, iterset, *args))[0]
    p = _affine_node_positions(m, V2, 2, 2)
    exact = np.sin(3.0 * p[:, 0]) * p[:, 1] + 0.25
    assert np.abs(got - exact).max() < 1e-14
    ref = oracle_run(it.kernel, iterset, *args)[0]
    assert np.abs(got - ref).max() < 1e-15


def test_coefficient_p1_into_p2_and_vector_target_on_host():
    from hostsim import run_direct
    m, V1, V2, xs, s1, s2 = _tri_case()
    p1, p2 = _affine_node_positions(m, V1, 2, 1), _affine_node_positions(m, V2, 2, 2)
    f = V1.dat(data=2.0 * p1[:, 0] - 3.0 * p1[:, 1] + 0.5)            # a linear function: P1 -> P2 is exact
    it = Interpolator(["w0[0]*w0[0]"], s2, m.coordinates, xs, coefficients=[(f, s1)])
    iterset, args = _loop(it, V2.dat())
    got = run_direct(op2.LegacyParloop(it.kernel, iterset, *args))[0]
    assert np.abs(got - (2.0 * p2[:, 0] - 3.0 * p2[:, 1] + 0.5) ** 2).max() < 1e-13
    sv = Space(V2.cell_node_map, simplex_node_points(2, 2), simplex_lagrange(2, 2), 2)
    it = Interpolator(["0.5 - X[1]", "X[0] - 0.5"], sv, m.coordinates, xs)
    iterset, args = _loop(it, V2.dat(dim=2))
    got = run_direct(op2.LegacyParloop(it.kernel, iterset, *args))[0]
    assert np.abs(got - np.stack([0.5 - p2[:, 1], p2[:, 0] - 0.5], axis=1)).max() < 1e-15


@pytest.mark.gpu
def test_interpolation_on_gpu():
    m, V1, V2, xs, s1, s2 = _tri_case(60)
    p1, p2 = _affine_node_positions(m, V1, 2, 1), _affine_node_positions(m, V2, 2, 2)
    c = op2.Global(1, 0.25)
    it = Interpolator(["sin(3.0*X[0])*X[1] + c0[0]"], s2, m.coordinates, xs, constants=[c])
    out = it.interpolate(V2.dat())
    assert np.abs(out.data_ro - (np.sin(3.0 * p2[:, 0]) * p2[:, 1] + 0.25)).max() < 1e-13
    iterset, args = _loop(it, V2.dat())
    assert np.abs(out.data_ro - oracle_run(it.kernel, iterset, *args)[0]).max() < 1e-13
    c.data[...] = -1.0                                                 # constants are run-time arguments
    it.interpolate(out)
    assert np.abs(out.data_ro - (np.sin(3.0 * p2[:, 0]) * p2[:, 1] - 1.0)).max() < 1e-13
    f = V1.dat(data=2.0 * p1[:, 0] - 3.0 * p1[:, 1] + 0.5)
    it = Interpolator(["w0[0]*w0[0]"], s2, m.coordinates, xs, coefficients=[(f, s1)])
    out = it.interpolate(V2.dat())
    assert np.abs(out.data_ro - (2.0 * p2[:, 0] - 3.0 * p2[:, 1] + 0.5) ** 2).max() < 1e-12
    # tets, P1 -> P2
    m3 = fmesh.UnitCubeMesh(6, degrees=(1, 2), perturb=0.1)
    W1, W2 = m3.space(1), m3.space(2)
    xs3 = Space(m3.coord_space.cell_node_map, simplex_node_points(3, 1), simplex_lagrange(3, 1), 3)
    t2 = Space(W2.cell_node_map, simplex_node_points(3, 2), simplex_lagrange(3, 2), 1)
    it = Interpolator(["X[0]*X[1] + exp(X[2])"], t2, m3.coordinates, xs3)
    out = it.interpolate(W2.dat())
    q = _affine_node_positions(m3, W2, 3, 2)
    n = W2.node_set.size
    assert np.abs(out.data_ro - (q[:n, 0] * q[:n, 1] + np.exp(q[:n, 2]))).max() < 1e-13


@pytest.mark.gpu
def test_dg_advection_demo_fields_interpolated_on_device():
    from firedrake_amd import forms
    m = fmesh.make_quad_mesh(40, perturb=0.1)
    prob = forms.DGAdvectionProblem(m)
    u_host = prob.u.data_ro.copy()
    prob.interpolate_demo_fields()
    assert np.abs(prob.u.data_ro - u_host).max() < 1e-15
    p = m.dq_points
    bell = 0.25 * (1 + np.cos(np.pi * np.minimum(np.hypot(p[:, 0] - 0.25, p[:, 1] - 0.5) / 0.15, 1.0)))
    cone = 1.0 - np.minimum(np.hypot(p[:, 0] - 0.5, p[:, 1] - 0.25) / 0.15, 1.0)
    slot = np.where(np.hypot(p[:, 0] - 0.5, p[:, 1] - 0.75) < 0.15,
                    np.where((p[:, 0] > 0.475) & (p[:, 0] < 0.525) & (p[:, 1] < 0.85), 0.0, 1.0), 0.0)
    assert np.abs(prob.q.data_ro - (1.0 + bell + cone + slot)).max() < 1e-13
    assert slot.sum() > 0
