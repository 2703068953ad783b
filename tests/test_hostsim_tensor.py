"""Host-sim parity of the tensor-product wrappers (codegen modes tp_action / tp_matrix): the generated wrapper and the device
templates of firedrake_amd/csrc/fd_tensor.h, compiled against tests/hostsim/mt/fd_wrapper.h (one OS thread per lane, the fp64
MFMA restated from its operand layout), against the oracle running the dense quadrature kernel through the extruded wrapper
restatement (builder.py:94-124, 790-831).  Test infrastructure only: the product path has no host route."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import hostsim
from firedrake_amd import forms, mesh as fmesh, op2
from test_gpu_q4_hex import _oracle_action, _oracle_matrix


@pytest.mark.parametrize("n,layers", [(1, 1), (2, 3)])
def test_q4_action_template_on_the_host(n, layers):
    """(2, 3): 12 cells = two full workgroups of five cells and one with two -- the idle cell slots and lanes 125..127."""
    m = fmesh.make_extruded_hex_mesh(n, layers, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m)
    y = hostsim.run_tensor(prob.act_loop)[0]
    ref = _oracle_action(m, prob.u.data_ro)
    assert_allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("bcs", [False, True])
def test_q4_mfma_matrix_template_on_the_host(bcs):
    """One column of two cells (bottom and top variants of the offset table); with BCs every boundary row and column is
    dropped through the lgmaps, which on a 1 x 1 x 2 mesh leaves the 2 x 9 + 9 interior nodes."""
    m = fmesh.make_extruded_hex_mesh(1, 2, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=bcs)
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None)
    v = csr.values.copy()
    if bcs:                                   # the unit diagonal of the BC rows is a separate pass (fd_csr_set_diagonal)
        rp, ci = csr.rowptr, csr.colidx
        for b in prob.bc_nodes:
            v[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


@pytest.mark.parametrize("degree,nq,n,layers", [(1, 2, 3, 3), (2, 3, 2, 3), (3, 4, 2, 2), (2, 4, 2, 2), (3, 3, 1, 2), (5, 6, 1, 1)])
def test_qk_action_template_on_the_host(degree, nq, n, layers):
    """The action template for other degrees and quadrature sizes: index cubes of extent max(k+1, nq) (more Gauss points than
    nodes per axis and fewer), 32 / 14 / 8 / 3 cells per workgroup."""
    m = fmesh.make_extruded_hex_mesh(n, layers, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, nq=nq)
    y = hostsim.run_tensor(prob.act_loop)[0]
    ref = _oracle_action(m, prob.u.data_ro, prob.kact)
    assert_allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max())


@pytest.mark.parametrize("degree,nq,bcs", [(1, 2, True), (2, 3, False), (2, 4, True), (3, 4, True)])
def test_qk_mfma_matrix_template_on_the_host(degree, nq, bcs):
    """The MFMA template for 1, 2 and 4 tiles per side (one, two and four wavefronts per workgroup), padded and exact.  (Q5 -- 14
    tiles, seven workgroups of two wavefronts per cell -- takes a minute of barriers per cell here; it runs in the -m gpu suite.)"""
    m = fmesh.make_extruded_hex_mesh(2 if degree < 3 else 1, 2, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, bcs=bcs, nq=nq)
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac)
    v = csr.values.copy()
    if bcs:
        rp, ci = csr.rowptr, csr.colidx
        for b in prob.bc_nodes:
            v[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


@pytest.mark.parametrize("degree,nq", [(2, 3), (3, 4)])
def test_convection_diffusion_reaction_weights_on_the_host(degree, nq):
    """A non-symmetric point weight (alpha grad u . grad v + (b . grad u) v + beta u v): the MFMA template's A operand is
    Phi^T W, the action applies W between the forward and the transposed passes -- both must keep rows and columns apart."""
    m = fmesh.make_extruded_hex_mesh(1, 2, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, nq=nq, alpha=0.7, beta=1.3, velocity=(1.0, -2.0, 0.5))
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_matrix(m, None, prob.kjac)
    assert_allclose(csr.values, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    A = ref.toscipy()
    assert abs(A - A.T).max() > 1e-3 * abs(A).max()              # genuinely non-symmetric
    y = hostsim.run_tensor(prob.act_loop)[0]
    yref = _oracle_action(m, prob.u.data_ro, prob.kact)
    assert_allclose(y, yref, rtol=0, atol=1e-12 * np.abs(yref).max())
    assert_allclose(y, A @ np.asarray(prob.u.data_ro), rtol=0, atol=1e-11 * np.abs(yref).max())


@pytest.mark.parametrize("degree,nq,bcs", [(2, 3, False), (3, 4, True), (4, 5, False)])
def test_coefficient_arguments_through_the_tensor_templates_on_the_host(degree, nq, bcs):
    """a(du, v) = int kappa(w0) grad(du).grad(v) + c(u0) du v dx: two coefficient fields on the Q_k map -- a variable diffusivity
    and the linearisation point of a nonlinear reaction term, the w_k arguments of a TSFC Jacobian
    (tsfc/kernel_interface/firedrake_loopy.py:432-522) -- evaluated at the Gauss points by the templates (sum-factorised, through
    LDS in the matrix kernel, through the axis passes in the action) against the oracle's dense kernel that tabulates them
    point by point."""
    m = fmesh.make_extruded_hex_mesh(1, 2, degree, perturb=0.1)
    prob = forms.CoefficientHexProblem(m, bcs=bcs, nq=nq)
    from firedrake_amd.codegen import tensor_eligible
    assert tensor_eligible(prob.jac_loop.global_kernel) == "matrix" and tensor_eligible(prob.act_loop.global_kernel) == "action"
    coefs = (prob.w0.data_ro, prob.u0.data_ro)
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac, coefs)
    v = csr.values.copy()
    if bcs:
        rp, ci = csr.rowptr, csr.colidx
        for b in prob.bc_nodes:
            v[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    # the coefficients matter: the constant-coefficient operator differs by O(1)
    plain = _oracle_matrix(m, prob.bc_nodes if bcs else None, forms.helmholtz_hex_jacobian_kernel(degree, nq))
    assert np.abs(plain.values - ref.values).max() > 0.05 * np.abs(ref.values).max()
    y = hostsim.run_tensor(prob.act_loop)[0]
    yref = _oracle_action(m, prob.u.data_ro, prob.kact, coefs)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())


@pytest.mark.parametrize("degree,nq,bcs", [(2, 3, False), (3, 4, True), (4, 5, True)])
def test_a_coefficient_on_the_q1_map_through_the_tensor_templates_on_the_host(degree, nq, bcs):
    """The diffusivity as a field in the Q1 space of the COORDINATES (8 vertex values per cell on the coordinate map: a piecewise-
    trilinear material under a high-order discretisation) next to a Q_k linearisation point: the templates stage its vertex values
    beside the coordinates and interpolate them at the Gauss points (trilinear in the matrix kernel, affine along a line in the
    action), the generated callback restores TSFC's argument order (here: Q1 first, Q_k second, the reverse of the templates')."""
    m = fmesh.make_extruded_hex_mesh(2, 2, degree, perturb=0.1)
    prob = forms.CoefficientHexProblem(m, bcs=bcs, nq=nq, q1_diffusivity=True)
    from firedrake_amd.codegen import tensor_coefficient_spaces, tensor_eligible
    assert tensor_eligible(prob.jac_loop.global_kernel) == "matrix" and tensor_eligible(prob.act_loop.global_kernel) == "action"
    assert tensor_coefficient_spaces(prob.jac_loop.global_kernel, "matrix") == ["1", "k"]
    assert tensor_coefficient_spaces(prob.act_loop.global_kernel, "action") == ["1", "k"]
    coefs = ((prob.w0.data_ro_with_halos, "1"), prob.u0.data_ro)
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac, coefs)
    v = csr.values.copy()
    if bcs:
        rp, ci = csr.rowptr, csr.colidx
        for b in prob.bc_nodes:
            v[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    # the Q1 field is not the Q_k field of the same function: the two operators differ
    other = forms.CoefficientHexProblem(m, bcs=bcs, nq=nq)
    ref_k = _oracle_matrix(m, prob.bc_nodes if bcs else None, other.kjac, (other.w0.data_ro, other.u0.data_ro))
    assert np.abs(ref_k.values - ref.values).max() > 1e-6 * np.abs(ref.values).max()
    y = hostsim.run_tensor(prob.act_loop)[0]
    yref = _oracle_action(m, prob.u.data_ro, prob.kact, coefs)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())


@pytest.mark.parametrize("degree,nq,n,bcs", [(2, 3, 2, False), (2, 3, 2, True), (3, 4, 1, True), (1, 2, 2, False)])
def test_matrix_template_accumulates_on_top_of_existing_values(degree, nq, n, bcs):
    """hex_qk_matrix adds into every row with atomics: a loop without a zero() accumulates on top of what the Mat holds (Mat INC,
    mat.py:851-855); dropped boundary rows and columns stay untouched."""
    m = fmesh.make_extruded_hex_mesh(n, 2, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, bcs=bcs, nq=nq)
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac)
    rp, ci = ref.rowptr, ref.colidx
    tol = 1e-12 * np.abs(ref.values).max()
    start = np.random.default_rng(5).standard_normal(ref.values.shape)
    acc = hostsim.run_tensor(prob.jac_loop, initial=start)[0]
    expect = ref.values.copy()
    for b in (prob.bc_nodes if bcs else ()):
        expect[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 0.0
    assert_allclose(acc.values, start + expect, rtol=0, atol=tol)


def _oracle_loop(pl):
    """the loop's own arguments through the oracle (generic adapter: blocked Mats, coefficient arguments on either map)"""
    from firedrake_amd.parloop import MatParloopArg
    from helpers import oracle_run
    args = [pa.data(acc, pa.maps, lgmaps=pa.lgmaps) if isinstance(pa, MatParloopArg) else pa.data(acc, pa.map_)
            for pa, acc in zip(pl.arguments, pl.accesses)]
    for pa, acc in zip(pl.arguments, pl.accesses):
        if int(acc) != int(op2.READ) and not isinstance(pa, MatParloopArg):
            pa.data.zero()                                     # (the oracle starts from the carrier's values)
    return oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]


@pytest.mark.parametrize("degree,nq,bcs,q1_state", [(2, 3, False, False), (3, 4, True, False), (2, 4, True, True), (1, 2, False, False)])
def test_coefficient_gradients_through_the_tensor_templates_on_the_host(degree, nq, bcs, q1_state):
    """The Newton Jacobian of int (1 + |grad u|^2) grad(u).grad(v) dx at u0: the point weight needs grad(u0) at the Gauss points
    (tsfc/fem.py:742-805 tabulates the derivative tables for coefficients exactly like for arguments).  The templates evaluate the
    reference gradient sum-factorised next to the value (matrix: three more results of the same LDS passes; action: the coefficient
    rides the axis passes like u itself) -- or, for a state in the Q1 space of the coordinates, from the staged vertex values --
    and hand it to the callback as DC; against the oracle's dense kernel that sums grad(phi_i) u0_i point by point."""
    m = fmesh.make_extruded_hex_mesh(2 if degree < 3 else 1, 2, degree, perturb=0.1)
    prob = forms.NonlinearDiffusionHexProblem(m, bcs=bcs, nq=nq, q1_state=q1_state)
    from firedrake_amd.codegen import tensor_eligible
    assert tensor_eligible(prob.jac_loop.global_kernel) == "matrix" and tensor_eligible(prob.act_loop.global_kernel) == "action"
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_loop(prob.jac_loop)
    assert np.array_equal(csr.rowptr, ref.rowptr) and np.array_equal(csr.colidx, ref.colidx)
    assert_allclose(csr.values, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    # the state matters: the Laplacian (u0 = 0) differs by O(1), and the rank-one term makes it more than a rescaling
    lap = _oracle_matrix(m, prob.bc_nodes if bcs else None, forms.helmholtz_hex_jacobian_kernel(degree, nq, alpha=1.0, beta=0.0))
    if bcs:
        rp, ci = lap.rowptr, lap.colidx
        for b in prob.bc_nodes:
            lap.values[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 0.0
    assert np.abs(lap.values - ref.values).max() > 0.05 * np.abs(ref.values).max()
    y = hostsim.run_tensor(prob.act_loop)[0]
    yref = _oracle_loop(prob.act_loop)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())


@pytest.mark.parametrize("degree,nq,bcs", [(2, 3, False), (1, 2, True), (3, 4, True)])
def test_a_vector_valued_space_through_the_tensor_templates_on_the_host(degree, nq, bcs):
    """Linear elasticity on (Q_k)^3: Mat dims (3, 3) -- element tensor t[(i*3 + p)][(j*3 + r)], MatSetValuesBlockedLocal
    (builder.py:573-625) -- and Dats of dim 3.  The matrix template runs one scalar MFMA contraction per (test, trial) component pair
    on its 4 x 4 slice of the 12 x 12 point weight and scatters into the blocked CSR; the action carries three components of u
    through the axis passes.  Against the oracle's dense 3 nd x 3 nd kernel; the operator is symmetric and annihilates the rigid
    translations (rho = 0)."""
    m = fmesh.make_extruded_hex_mesh(1 if degree > 2 else 2, 2, degree, perturb=0.1)
    prob = forms.ElasticityHexProblem(m, bcs=bcs, nq=nq)
    from firedrake_amd.codegen import tensor_eligible
    assert tensor_eligible(prob.jac_loop.global_kernel) == "matrix" and tensor_eligible(prob.act_loop.global_kernel) == "action"
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_loop(prob.jac_loop)
    assert np.array_equal(csr.rowptr, ref.rowptr) and np.array_equal(csr.colidx, ref.colidx)
    assert_allclose(csr.values, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    A = ref.toscipy()
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
    if not bcs:
        for c in range(3):
            t = np.zeros((m.node_set.total_size, 3))
            t[:, c] = 1.0
            assert np.abs(A @ t.ravel()).max() <= 1e-11 * abs(A).max()
    y = hostsim.run_tensor(prob.act_loop)[0]
    yref = _oracle_loop(prob.act_loop)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())
    if not bcs:
        assert_allclose(y.ravel(), A @ np.asarray(prob.u.data_ro).ravel(), rtol=0, atol=1e-10 * np.abs(yref).max())


@pytest.mark.parametrize("degree,nq,bcs,panel,chunk,wlds", [(3, 4, True, 3, 2, 1024), (2, 3, False, 1, 1, 1 << 20), (3, 5, False, 14, 8, 4096),
                                                             (4, 5, True, 7, 3, 1 << 20)])
def test_qk_mfma_matrix_column_chunks_and_weight_slabs_on_the_host(degree, nq, bcs, panel, chunk, wlds, monkeypatch):
    """Q6 and beyond (22+ tiles per side) cut a 16-row panel into column chunks -- one wavefront per (panel, chunk) -- and compute the
    point weights one q1 plane at a time once a cell's weights pass 48 KB of LDS (fd_tensor.h tp_col_splits / tp_weight_slabs).  On the
    host the REAL Q6 template takes seven minutes per cell, so the two mechanisms are run on small elements with the thresholds
    lowered: Q3 (4 tiles) as 2 chunks of 2 with plane-wise weights, Q2 (2 tiles) as 2 chunks of 1, Q3 with 5 points plane-wise only,
    Q4 (8 tiles) as 3 chunks of 3 -- the last chunk's third tile lies beyond the matrix;
    Q6 / Q7 themselves run in the -m gpu suite (tests/test_gpu_q4_hex.py)."""
    from firedrake_amd.codegen import configuration, tensor_geometry
    monkeypatch.setitem(configuration, "tp_max_panel_tiles", panel)
    monkeypatch.setitem(configuration, "tp_chunk_tiles", chunk)
    monkeypatch.setitem(configuration, "tp_weight_lds", wlds)
    g = tensor_geometry(degree, nq)
    assert g["col_splits"] == (1 if g["tiles"] <= panel else -(-g["tiles"] // chunk)) and g["matrix_groups"] * g["matrix_threads"] == 64 * g["tiles"] * g["col_splits"]
    m = fmesh.make_extruded_hex_mesh(1, 2, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, bcs=bcs, nq=nq, alpha=0.7, beta=1.3, velocity=(1.0, -2.0, 0.5))
    csr = hostsim.run_tensor(prob.jac_loop)[0]
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac)
    v = csr.values.copy()
    if bcs:
        rp, ci = csr.rowptr, csr.colidx
        for b in prob.bc_nodes:
            v[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


@pytest.mark.parametrize("degree,nq", [(6, 8), (7, 9), (8, 10)])
def test_high_order_action_template_on_the_host(degree, nq):
    """Q6, Q7, Q8: 64 / 81 / 100 lanes per cell (one or two cells per workgroup), lines of 7 to 9 nodes and 8 to 10 points."""
    m = fmesh.make_extruded_hex_mesh(1, 2, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, nq=nq, alpha=0.7, beta=1.3, velocity=(1.0, -2.0, 0.5))
    y = hostsim.run_tensor(prob.act_loop)[0]
    ref = _oracle_action(m, prob.u.data_ro, prob.kact)
    assert_allclose(y, ref, rtol=0, atol=1e-12 * np.abs(ref).max())
