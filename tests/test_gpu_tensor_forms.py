"""GPU: the wider tensor-product descriptors -- coefficient GRADIENTS at the Gauss points (the Newton Jacobian of a form nonlinear in
grad(u); tsfc/fem.py:742-805) and VECTOR-VALUED Q_k spaces (Mat dims (3, 3), MatSetValuesBlockedLocal, builder.py:573-625) -- through
the backend-picked wrappers (tp_matrix on the fp64 matrix cores, tp_action sum-factorised) against the oracle running the dense C text
of the same local kernels, to 1e-11."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import forms, mesh as fmesh, op2
from firedrake_amd.parloop import MatParloopArg
from helpers import oracle_run

pytestmark = pytest.mark.gpu


def _oracle_loop(pl):
    args = [pa.data(acc, pa.maps, lgmaps=pa.lgmaps) if isinstance(pa, MatParloopArg) else pa.data(acc, pa.map_)
            for pa, acc in zip(pl.arguments, pl.accesses)]
    for pa, acc in zip(pl.arguments, pl.accesses):
        if int(acc) != int(op2.READ) and not isinstance(pa, MatParloopArg):
            pa.data.zero()
    return oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]


def _check(prob, bcs):
    mat = prob.assemble_jacobian()
    assert prob.jac_loop._prepare()["cw"].src.mode == "tp_matrix"
    rp, ci, v = mat.csr()
    v = np.array(v)
    y = np.array(prob.assemble_action().data_ro)
    assert prob.act_loop._prepare()["cw"].src.mode == "tp_action"
    ref = _oracle_loop(prob.jac_loop)
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    if bcs and hasattr(prob, "_bc_rows"):            # (the scalar problems set the unit diagonal of the BC rows in a separate pass)
        for b in prob.bc_nodes:
            ref.values[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    yref = _oracle_loop(prob.act_loop)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())
    return ref, yref


@pytest.mark.parametrize("degree,nq,n,layers,bcs,q1_state", [(2, 3, 3, 3, True, False), (3, 4, 2, 3, False, False), (3, 5, 2, 2, True, True),
                                                             (4, 5, 2, 2, True, False)])
def test_nonlinear_diffusion_jacobian_with_coefficient_gradients(degree, nq, n, layers, bcs, q1_state):
    m = fmesh.make_extruded_hex_mesh(n, layers, degree, perturb=0.1)
    prob = forms.NonlinearDiffusionHexProblem(m, bcs=bcs, nq=nq, q1_state=q1_state)
    ref, _ = _check(prob, bcs)
    A = ref.toscipy()
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()            # kappa K K^T + 2 (K g)(K g)^T is symmetric


@pytest.mark.parametrize("degree,nq,n,layers,bcs", [(2, 3, 3, 3, False), (2, 3, 2, 4, True), (1, 2, 4, 3, True), (3, 4, 2, 2, True)])
def test_linear_elasticity_on_a_vector_valued_space(degree, nq, n, layers, bcs):
    m = fmesh.make_extruded_hex_mesh(n, layers, degree, perturb=0.1)
    prob = forms.ElasticityHexProblem(m, bcs=bcs, nq=nq)
    ref, yref = _check(prob, bcs)
    A = ref.toscipy()
    assert abs(A - A.T).max() <= 1e-12 * abs(A).max()
    if not bcs:
        assert_allclose(yref.ravel(), A @ np.asarray(prob.u.data_ro).ravel(), rtol=0, atol=1e-10 * np.abs(yref).max())
        t = np.zeros((m.node_set.total_size, 3))
        t[:, 1] = 1.0
        assert np.abs(A @ t.ravel()).max() <= 1e-11 * abs(A).max()       # rigid translations are in the kernel (rho = 0)


def test_elasticity_matrix_twice_and_after_other_values():
    """the blocked scatter adds into whatever the Mat holds: two loops without zeroing in between give twice the matrix"""
    m = fmesh.make_extruded_hex_mesh(2, 3, 2, perturb=0.1)
    prob = forms.ElasticityHexProblem(m, nq=3)
    v1 = np.array(prob.assemble_jacobian().csr()[2])
    prob.jac_loop()
    v2 = np.array(prob.mat.csr()[2])
    assert_allclose(v2, 2 * v1, rtol=0, atol=1e-12 * np.abs(v1).max())
