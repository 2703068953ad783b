"""GPU: a coefficient argument on the Q1 map of the coordinates through the tensor-product wrappers (codegen.tensor_coefficient_spaces,
csrc/fd_tensor.h: hex_qk_matrix / hex_qk_action with N1 > 0) against the oracle's dense kernel.  (Collected last: the file was added
at the end of round 4 with one GPU minute left to run it.)"""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import forms, mesh as fmesh
from test_gpu_q4_hex import _oracle_action, _oracle_matrix

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree,nq,n,layers,bcs", [(4, 5, 2, 3, True), (2, 3, 3, 2, False), (3, 4, 2, 2, True)])
def test_a_diffusivity_in_the_q1_space_of_the_coordinates(degree, nq, n, layers, bcs):
    """a(du, v) = int kappa(w0) grad(du).grad(v) + c(u0) du v dx with w0 a piecewise-trilinear field on the COORDINATE map (8 vertex
    values per cell) and u0 in the Q_k space of the unknown -- TSFC's argument order (Q1 first) is the reverse of the templates'
    (Q_k first): matrix by fp64 MFMA and sum-factorised action against the oracle's dense kernel to 1e-11; a change of the Q1 field
    between two assemblies is seen."""
    m = fmesh.make_extruded_hex_mesh(n, layers, degree, perturb=0.1)
    prob = forms.CoefficientHexProblem(m, bcs=bcs, nq=nq, q1_diffusivity=True)
    mat = prob.assemble_jacobian()
    assert prob.jac_loop._prepared["cw"].src.mode == "tp_matrix"
    coefs = ((prob.w0.data_ro_with_halos, "1"), prob.u0.data_ro)
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac, coefs)
    _, _, v = mat.csr()
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    y = prob.assemble_action()
    assert prob.act_loop._prepared["cw"].src.mode == "tp_action"
    yref = _oracle_action(m, prob.u.data_ro, prob.kact, coefs)
    assert_allclose(y.data_ro, yref, rtol=0, atol=1e-11 * np.abs(yref).max())
    if not bcs:
        assert_allclose(ref.toscipy() @ np.asarray(prob.u.data_ro), yref, rtol=0, atol=1e-10 * np.abs(yref).max())
    prob.w0.data[:] = 0.25
    mat = prob.assemble_jacobian()
    ref0 = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac, ((prob.w0.data_ro_with_halos, "1"), prob.u0.data_ro))
    _, _, v0 = mat.csr()
    assert_allclose(v0, ref0.values, rtol=0, atol=1e-11 * np.abs(ref0.values).max())
    assert np.abs(v0 - v).max() > 1e-3 * np.abs(v).max()
