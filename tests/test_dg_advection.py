"""Config C4 (DG advection demo form L1): CPU identities of the hand-restated facet kernels through the
oracle; GPU parity of the three-parloop assembly (cell + ds + dS, arity-8 maps, direct uint32 facet Dat)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import ODat, OGlobal, READ, INC
from firedrake_amd import forms, mesh as fmesh, op2
from firedrake_amd.configuration import configuration


def _oracle_rhs(prob, q=None):
    m = prob.mesh
    kc, ke, ki = forms.dg_advection_kernels()
    L = np.zeros(m.dq_set.total_size)
    x = np.array(m.coordinates.data_ro)
    qv = np.array(prob.q.data_ro) if q is None else q
    uv = np.array(prob.u.data_ro)
    dtc, qin = np.array(prob.dtc.data_ro), np.array(prob.q_in.data_ro)
    oracle.par_loop(kc.code, kc.name, 0, m.cell_set.size, [ODat(L, INC, m.cell_dq.values), ODat(x, READ, m.cell_q1.values),
                    ODat(qv, READ, m.cell_dq.values), ODat(uv, READ, m.cell_q1.values), OGlobal(dtc, READ)])
    oracle.par_loop(ke.code, ke.name, 0, m.ext_facet_set.size, [ODat(L, INC, m.ext_dq.values), ODat(x, READ, m.ext_q1.values),
                    ODat(qv, READ, m.ext_dq.values), ODat(uv, READ, m.ext_q1.values), OGlobal(dtc, READ), OGlobal(qin, READ),
                    ODat(np.array(m.ext_local_facet.data_ro), READ)])
    oracle.par_loop(ki.code, ki.name, 0, m.int_facet_set.size, [ODat(L, INC, m.int_dq.values), ODat(x, READ, m.int_q1.values),
                    ODat(qv, READ, m.int_dq.values), ODat(uv, READ, m.int_q1.values), OGlobal(dtc, READ),
                    ODat(np.array(m.int_local_facet.data_ro), READ)])
    return L


def test_facet_counts_match_demo_mesh():
    m = fmesh.make_quad_mesh(40)
    # 1600 cells, 6400 DoFs, 3120 interior + 160 exterior facets
    assert (m.cell_set.size, m.dq_set.size, m.int_facet_set.size, m.ext_facet_set.size) == (1600, 6400, 3120, 160)
    lf = m.int_local_facet.data_ro
    assert set(map(tuple, lf.tolist())) == {(1, 0), (3, 2)}


@pytest.mark.parametrize("perturb", [0.0, 0.2])
def test_conservation_and_constant_state(perturb):
    """q = q_in = 1 with a divergence-free velocity: every row of L1 vanishes? No -- but the total does
    (mass conservation, tests/firedrake/regression/test_dg_advection.py:70-73), and upwinding a constant
    state gives the same flux from both sides."""
    m = fmesh.make_quad_mesh(12, tile=(4, 4), perturb=perturb)
    prob = forms.DGAdvectionProblem(m)
    L = _oracle_rhs(prob, q=np.ones(m.dq_set.size))
    assert abs(L.sum()) < 1e-13
    # general q: sum_i L_i = -dt * net boundary flux; interior jumps cancel.  Check against direct quadrature
    L2 = _oracle_rhs(prob)
    assert np.isfinite(L2).all() and abs(L2).max() > 0
    # interior-facet contributions alone sum to zero
    ki = forms.dg_advection_kernels()[2]
    Li = np.zeros(m.dq_set.size)
    oracle.par_loop(ki.code, ki.name, 0, m.int_facet_set.size,
                    [ODat(Li, INC, m.int_dq.values), ODat(np.array(m.coordinates.data_ro), READ, m.int_q1.values),
                     ODat(np.array(prob.q.data_ro), READ, m.int_dq.values), ODat(np.array(prob.u.data_ro), READ, m.int_q1.values),
                     OGlobal(np.array(prob.dtc.data_ro), READ), ODat(np.array(m.int_local_facet.data_ro), READ)])
    assert abs(Li.sum()) < 1e-14 * max(1.0, abs(Li).sum())


@pytest.mark.gpu
@pytest.mark.parametrize("n,perturb", [(40, 0.0), (40, 0.15), (7, 0.1)])
@pytest.mark.parametrize("mode", ["auto", "direct"])
def test_dg_rhs_matches_oracle(n, perturb, mode, monkeypatch):
    monkeypatch.setitem(configuration, "mode", mode)
    m = fmesh.make_quad_mesh(n, perturb=perturb)
    prob = forms.DGAdvectionProblem(m)
    L = prob.assemble_rhs()
    ref = _oracle_rhs(prob)
    assert_allclose(L.data_ro, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
    L = prob.assemble_rhs()           # reuse (frozen halo, zero + three loops)
    assert_allclose(L.data_ro, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.gpu
def test_dg_rhs_conservation_at_scale():
    m = fmesh.make_quad_mesh(512)
    prob = forms.DGAdvectionProblem(m)
    prob.q.assign(1.0)
    L = prob.assemble_rhs()
    assert abs(L.data_ro.sum()) < 1e-11


@pytest.mark.gpu
def test_dg_advection_time_loop_conserves_mass_and_is_l2_stable():
    """Mirror of tests/firedrake/regression/test_dg_advection.py:5-83 on the demo's own problem: ten SSPRK3 steps
    on the device; the L2 norm must not grow and the mass must be conserved."""
    m = fmesh.make_quad_mesh(40)
    st = forms.DGAdvectionStepper(m)
    mass0, l2_0 = st.integrals()
    for _ in range(10):
        st.step()
    mass1, l2_1 = st.integrals()
    assert l2_1 < l2_0
    assert np.isclose(mass1, mass0, rtol=1e-10)
    # one stage against the oracle: dq = M^{-1} L(q)
    ks = forms.dg_mass_solve_kernel()
    st._solve()
    L = _oracle_rhs(st.prob)
    dq = np.zeros(m.dq_set.size)
    oracle.par_loop(ks.code, ks.name, 0, m.cell_set.size,
                    [ODat(dq, oracle.WRITE, m.cell_dq.values), ODat(np.array(m.coordinates.data_ro), READ, m.cell_q1.values),
                     ODat(L, READ, m.cell_dq.values)])
    assert_allclose(st.dq.data_ro, dq, rtol=0, atol=1e-11 * np.abs(dq).max())


@pytest.mark.parametrize("lane_strided,direct_noreuse", [(1, False), (0, False), (1, True)])
def test_dg_rhs_staged_wrappers_on_host(lane_strided, direct_noreuse, monkeypatch):
    """The three staged wrappers of the demo's RHS (cell, exterior-facet and interior-facet loops: arity-8 maps, a
    direct uint32 facet-number Dat addressed through the lane-order entity formula, READ Globals) executed by the
    multi-threaded host-sim, chained like assemble_rhs, against the oracle."""
    from hostsim import run_staged
    monkeypatch.setitem(configuration, "lane_strided", lane_strided)
    m = fmesh.make_quad_mesh(9, tile=(4, 4), perturb=0.15)
    prob = forms.DGAdvectionProblem(m)
    L = np.zeros(m.dq_set.total_size)
    for loop, epb in zip(prob.loops, (30, 16, 70)):
        prob.L._host_rw()[...] = L                       # the loops accumulate into the same Dat
        # direct_noreuse: the DQ1 map of the CELL loop has no reuse inside a block (every node belongs to one cell), so its READ and
        # INC arguments bypass LDS ("_d" variant) beside the staged Q1 fields; the facet loops keep every map staged
        L = run_staged(loop, epb=epb, direct_noreuse=direct_noreuse)[0]
    ref = _oracle_rhs(prob)
    assert np.abs(L - ref).max() <= 1e-12 * np.abs(ref).max()


def test_dg_cell_and_facet_kernels_known_answers():
    """Known answers on the unit square cell with u = (1, 0), q = 1 (vertex a*2 + b at (a, b)):
    cell term dt * int u.grad(phi_i) = dt * (-1/2, -1/2, 1/2, 1/2); the exterior facets at x = 1 (outflow, facet 1)
    give -dt * int phi_i q u.n = -dt * (0, 0, 1/2, 1/2), at x = 0 (inflow of q_in, facet 0) +dt * q_in * (1/2, 1/2, 0, 0),
    and the facets y = 0, 1 (u.n = 0) nothing -- the upwind form of demos/DG_advection/DG_advection.py.rst."""
    kc, ke, _ = forms.dg_advection_kernels()
    xc = np.array([[0.0, 0.0], [0.0, 1.0], [1.0, 0.0], [1.0, 1.0]])
    q, u = np.ones(4), np.tile([1.0, 0.0], (4, 1))
    ident = np.arange(4, dtype=np.int32).reshape(1, 4)
    dt = 0.125
    L = np.zeros(4)
    oracle.par_loop(kc.code, kc.name, 0, 1, [ODat(L, INC, ident), ODat(xc.copy(), READ, ident), ODat(q.copy(), READ, ident),
                                             ODat(u.copy(), READ, ident), OGlobal(np.array([dt]), READ)])
    assert_allclose(L, dt * np.array([-0.5, -0.5, 0.5, 0.5]), atol=1e-15)
    expect = {0: 0.75 * dt * np.array([0.5, 0.5, 0, 0]), 1: -dt * np.array([0, 0, 0.5, 0.5]), 2: np.zeros(4), 3: np.zeros(4)}
    for f, exp in expect.items():
        L = np.zeros(4)
        oracle.par_loop(ke.code, ke.name, 0, 1, [ODat(L, INC, ident), ODat(xc.copy(), READ, ident), ODat(q.copy(), READ, ident),
                                                 ODat(u.copy(), READ, ident), OGlobal(np.array([dt]), READ),
                                                 OGlobal(np.array([0.75]), READ), ODat(np.array([[f]], dtype=np.uint32), READ)])
        assert_allclose(L, exp, atol=1e-15)


def test_dg_interior_facet_kernel_known_answer():
    """Two unit cells side by side, u = (1, 0): the facet x = 1 is facet 1 of the '+' cell and facet 0 of the '-' cell;
    upwinding takes q('+'): -dt * int (phi('+') - phi('-')) * q('+') dS = dt * (0, 0, -q/2, -q/2 | q/2, q/2, 0, 0)."""
    ki = forms.dg_advection_kernels()[2]
    xc = np.array([[0, 0], [0, 1], [1, 0], [1, 1], [1, 0], [1, 1], [2, 0], [2, 1]], dtype=float)
    ident = np.arange(8, dtype=np.int32).reshape(1, 8)
    q = np.array([3.0] * 4 + [7.0] * 4)                       # q('+') = 3, q('-') = 7 (not used: it is downwind)
    u = np.tile([1.0, 0.0], (8, 1))
    dt = 0.25
    L = np.zeros(8)
    oracle.par_loop(ki.code, ki.name, 0, 1, [ODat(L, INC, ident), ODat(xc, READ, ident), ODat(q, READ, ident), ODat(u, READ, ident),
                                             OGlobal(np.array([dt]), READ), ODat(np.array([[1, 0]], dtype=np.uint32), READ)])
    assert_allclose(L, dt * 3.0 * np.array([0, 0, -0.5, -0.5, 0.5, 0.5, 0, 0]), atol=1e-15)
    # reversed flow: the '-' cell is upwind
    L = np.zeros(8)
    oracle.par_loop(ki.code, ki.name, 0, 1, [ODat(L, INC, ident), ODat(xc, READ, ident), ODat(q, READ, ident), ODat(-u, READ, ident),
                                             OGlobal(np.array([dt]), READ), ODat(np.array([[1, 0]], dtype=np.uint32), READ)])
    assert_allclose(L, dt * 7.0 * np.array([0, 0, 0.5, 0.5, -0.5, -0.5, 0, 0]), atol=1e-15)
