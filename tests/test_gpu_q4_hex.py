"""Config C3 parity: the fp64-MFMA Q4 hexahedron kernel against the oracle running the dense quadrature
C kernel through the extruded wrapper restatement (builder.py:94-124, 790-831)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import ODat, OMat, READ, INC
from firedrake_amd import forms, mesh as fmesh


def _oracle_matrix(m):
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    nn = m.node_set.total_size
    csr = oracle.build_sparsity(nn, nn, [(cm, cm, m.layers, m.cell_node_map.offset, m.cell_node_map.offset)])
    k = forms.helmholtz_q4_hex_jacobian_kernel()
    oracle.par_loop(k.code, k.name, 0, m.base_set.size,
                    [OMat(csr, INC, cm, cm, roffset=m.cell_node_map.offset, coffset=m.cell_node_map.offset),
                     ODat(np.array(m.coordinates.data_ro_with_halos), READ, xm, offset=m.coord_map.offset)],
                    layers=(0, m.layers + 1))
    return csr


def test_q4_tables_and_oracle_kernel_identities():
    """CPU: partition of unity, exact derivative of a quartic, sum(mass part)=volume via K*1=M*1 split."""
    L, DL, qp, qw = forms.q4_tables()
    assert_allclose(L.sum(axis=1), 1.0, atol=1e-14)
    assert_allclose(DL.sum(axis=1), 0.0, atol=1e-12)
    assert_allclose(qw.sum(), 1.0, atol=1e-15)
    m = fmesh.make_extruded_hex_mesh(2, 2, 4, perturb=0.1)
    A = _oracle_matrix(m).toscipy()
    one = np.ones(A.shape[0])
    # (grad 1 = 0) => A*1 = M*1 and 1^T M 1 = |Omega| = 1
    assert_allclose(one @ (A @ one), 1.0, rtol=1e-12)
    assert abs(A - A.T).max() < 1e-13 * abs(A).max()
    # u = x (in Q4 since the geometry is trilinear only approximately -> use the unperturbed mesh)
    m0 = fmesh.make_extruded_hex_mesh(2, 2, 4, perturb=0.0)
    A0 = _oracle_matrix(m0).toscipy()
    x = m0.node_points[:, 0]
    # x^T A x = int |grad x|^2 + x^2 = 1 + 1/3
    assert_allclose(x @ (A0 @ x), 1.0 + 1.0 / 3.0, rtol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("n,layers", [(2, 2), (3, 2), (2, 5)])
def test_q4_mfma_matches_oracle(n, layers):
    m = fmesh.make_extruded_hex_mesh(n, layers, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m)
    mat = prob.assemble_jacobian()
    ref = _oracle_matrix(m)
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    # 1e-11 * max|A| for Q4 (different contraction order under MFMA)
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    v2 = prob.assemble_jacobian().csr()[2]
    assert_allclose(v2, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())


@pytest.mark.gpu
def test_q4_mfma_properties_at_scale():
    """n=8: 512 cells, 35937 DoFs -- properties instead of the oracle."""
    m = fmesh.make_extruded_hex_mesh(8, 8, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m)
    A = prob.assemble_jacobian().toscipy()
    one = np.ones(A.shape[0])
    assert_allclose(one @ (A @ one), 1.0, rtol=1e-11)
    assert abs(A - A.T).max() < 1e-12 * abs(A).max()


@pytest.mark.gpu
def test_q4_mfma_full_size_properties():
    """BASELINE.json configs[2] at the benchmark size (n = 32: 32768 cells, 2 146 689 DoFs, nnz 4.5e8): the matrix stays on
    the device; properties through the device SpMV: 1'A1 = |Omega| (grad 1 = 0), x'A1 = int x = 1/2, symmetry x'Ay = y'Ax."""
    from firedrake_amd import op2
    m = fmesh.make_extruded_hex_mesh(32, 32, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m)
    mat = prob.assemble_jacobian()
    assert m.node_set.size == 2146689 and mat.sparsity.nz == 454756609
    nn = m.node_set.size
    rng = np.random.default_rng(1)
    one, xs = np.ones(nn), m.node_points[:, 0].copy()
    a, b = rng.standard_normal(nn), rng.standard_normal(nn)
    d1, da, db = op2.Dat(m.node_set, one), op2.Dat(m.node_set, a), op2.Dat(m.node_set, b)
    t1, ta, tb = op2.Dat(m.node_set), op2.Dat(m.node_set), op2.Dat(m.node_set)
    mat.mult(d1, t1); mat.mult(da, ta); mat.mult(db, tb)
    A1 = np.array(t1.data_ro)
    assert abs(one @ A1 - 1.0) < 1e-10
    # the geometry is perturbed, so int x dx is not exactly 1/2; but A1 = M1 is the lumped mass: positive, sums to 1
    assert A1.min() > 0
    lhs, rhs = float(b @ ta.data_ro), float(a @ tb.data_ro)
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs))
    del xs
