"""GPU parity against the ORACLE at the sizes BASELINE.json quotes (not identities): C2 Poisson CG1 on 215^3 cubes
(59.6 M tets, 10.08 M DoFs, 150 M nonzeros) with boundary conditions, in the benchmark's tiled numbering and in
reference-like / random numberings with no producer hints; C5's per-GPU share (CG2 on 107^3 cubes, 9.94 M DoFs);
C4 DG advection at 2048^2 cells.  Plan capacity limits, 16-bit offsets and block splits only trigger at this scale.
The oracle runs single-threaded in seconds per form.  Tolerances: SURVEY.md Appendix D."""
import numpy as np
import pytest

import oracle
from oracle import ODat, OMat, READ, INC
from firedrake_amd import forms, mesh as fmesh

pytestmark = pytest.mark.gpu


def _oracle_poisson(prob):
    """(residual with BC rows zeroed, CSR values with BC rows/cols dropped and unit BC diagonal, oracle CSR)."""
    m = prob.mesh
    V, X = prob.V, m.coord_space
    cm, xm = V.cell_node_map.values_with_halo, X.cell_node_map.values_with_halo
    nn = V.node_set.total_size
    coords = np.array(m.coordinates.data_ro_with_halos)
    r = np.zeros(nn)
    oracle.par_loop(prob.kres.code, prob.kres.name, 0, m.cell_set.size,
                    [ODat(r, INC, cm), ODat(coords, READ, xm), ODat(np.array(prob.u.data_ro_with_halos), READ, cm),
                     ODat(np.array(prob.f.data_ro_with_halos), READ, cm)])
    r[prob.bc_nodes] = 0.0
    csr = oracle.build_sparsity(nn, nn, [(cm, cm)])
    lg = np.arange(nn, dtype=np.int32)
    lg[prob.bc_nodes] = -1
    oracle.par_loop(prob.kjac.code, prob.kjac.name, 0, m.cell_set.size,
                    [OMat(csr, INC, cm, cm, row_lgmap=lg, col_lgmap=lg), ODat(coords, READ, xm)])
    rp, ci = csr.rowptr, csr.colidx
    for b in prob.bc_nodes:                              # assemble.py:1501-1507: weight 1 on the BC diagonal
        csr.values[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    return r, csr


def _compare_poisson(prob):
    r = np.array(prob.assemble_residual().data_ro)
    mat = prob.assemble_jacobian()
    rp, ci, v = mat.csr()
    ro, csr = _oracle_poisson(prob)
    assert np.array_equal(rp, csr.rowptr) and np.array_equal(ci, csr.colidx)          # bit-exact pattern
    assert np.abs(r - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max())
    assert np.abs(v - csr.values).max() <= 1e-12 * np.abs(csr.values).max()
    # a second assembly into the same tensors (pending zero fused / plans reused) reproduces it
    v2 = prob.assemble_jacobian().csr()[2]
    assert np.abs(v2 - csr.values).max() <= 1e-12 * np.abs(csr.values).max()
    return len(v)


@pytest.mark.parametrize("numbering,n", [("tiled", 215), ("lexicographic", 215), ("random", 215)])
def test_c2_against_oracle(numbering, n):
    """BASELINE.json configs[1] with Dirichlet BCs on the whole boundary."""
    m = fmesh.UnitCubeMesh(n, degrees=(1,), perturb=0.1, numbering=numbering)
    assert (not hasattr(m.space(1).cell_node_map, "preferred_blocks")) == (numbering != "tiled")
    prob = forms.PoissonProblem(m, 1, bcs=True)
    nnz = _compare_poisson(prob)
    if n == 215:
        assert prob.V.node_set.size == 10077696 and m.cell_set.size == 59630250 and nnz == 150048286


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic"])
def test_c5_per_gpu_share_against_oracle(numbering):
    """BASELINE.json configs[4], one GPU's share: CG2 on 107^3 cubes (7.35 M tets, 9.94 M DoFs)."""
    m = fmesh.UnitCubeMesh(107, degrees=(2,), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 2, bcs=True)
    assert prob.V.node_set.size == 215 ** 3
    _compare_poisson(prob)


def test_c4_full_size_against_oracle():
    """BASELINE.json configs[3] at the timing size: DQ1 on 2048^2 quadrilaterals, the three-parloop RHS."""
    from test_dg_advection import _oracle_rhs
    m = fmesh.make_quad_mesh(2048, perturb=0.1)
    prob = forms.DGAdvectionProblem(m)
    L = np.array(prob.assemble_rhs().data_ro)
    ref = _oracle_rhs(prob)
    assert m.cell_set.size == 2048 ** 2 and m.int_facet_set.size == 2 * 2048 * 2047
    assert np.abs(L - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
