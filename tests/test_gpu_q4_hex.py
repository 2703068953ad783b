"""Config C3 parity: the fp64-MFMA Q4 hexahedron kernel against the oracle running the dense quadrature
C kernel through the extruded wrapper restatement (builder.py:94-124, 790-831)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import ODat, OMat, READ, INC
from firedrake_amd import forms, mesh as fmesh


def _coef_args(m, coefs):
    """Oracle arguments of the coefficient fields: an array lives on the Q_k map, a pair (array, "1") on the Q1 map of the coordinates."""
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    out = []
    for c in coefs:
        if isinstance(c, tuple):
            out.append(ODat(np.array(c[0]), READ, xm, offset=m.coord_map.offset))
        else:
            out.append(ODat(np.array(c), READ, cm, offset=m.cell_node_map.offset))
    return out


def _oracle_matrix(m, bc_nodes=None, k=None, coefs=()):
    """Dense Q4 kernel through the oracle's extruded wrapper; BC rows/columns dropped through the lgmaps and the unit
    diagonal set afterwards, as assemble.py:1501-1507 / 2075-2108 do."""
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    nn = m.node_set.total_size
    csr = oracle.build_sparsity(nn, nn, [(cm, cm, m.layers, m.cell_node_map.offset, m.cell_node_map.offset)])
    k = k or forms.helmholtz_q4_hex_jacobian_kernel()
    lg = None
    if bc_nodes is not None and len(bc_nodes):
        lg = np.arange(nn, dtype=np.int32)
        lg[bc_nodes] = -1
    oracle.par_loop(k.code, k.name, 0, m.base_set.size,
                    [OMat(csr, INC, cm, cm, roffset=m.cell_node_map.offset, coffset=m.cell_node_map.offset, row_lgmap=lg, col_lgmap=lg),
                     ODat(np.array(m.coordinates.data_ro_with_halos), READ, xm, offset=m.coord_map.offset)]
                    + _coef_args(m, coefs),
                    layers=(0, m.layers + 1))
    if lg is not None:
        rp, ci = csr.rowptr, csr.colidx
        for b in bc_nodes:
            csr.values[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = 1.0
    return csr


def _oracle_action(m, u, k=None, coefs=()):
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    y = np.zeros(m.node_set.total_size)
    k = k or forms.helmholtz_q4_hex_action_kernel()
    oracle.par_loop(k.code, k.name, 0, m.base_set.size,
                    [ODat(y, INC, cm, offset=m.cell_node_map.offset),
                     ODat(np.array(m.coordinates.data_ro_with_halos), READ, xm, offset=m.coord_map.offset),
                     ODat(np.array(u), READ, cm, offset=m.cell_node_map.offset)]
                    + _coef_args(m, coefs), layers=(0, m.layers + 1))
    return y


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
@pytest.mark.parametrize("n,layers", [(2, 2), (3, 2), (2, 5), (3, 1), (8, 8)])
@pytest.mark.parametrize("bcs", [False, True])
def test_q4_mfma_matches_oracle(n, layers, bcs):
    """The Q4 matrix through an ordinary parloop (Mat INC over the extruded cell set, BC lgmaps): GlobalKernel.compile
    selects the fp64-MFMA wrapper from the TensorProductLocalKernel descriptor; the oracle runs the kernel's dense C text."""
    m = fmesh.make_extruded_hex_mesh(n, layers, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=bcs)
    assert prob.jac_loop._prepare()["cw"].src.mode == "tp_matrix"
    mat = prob.assemble_jacobian()
    ref = _oracle_matrix(m, prob.bc_nodes)
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    # 1e-11 * max|A| for Q4 (different contraction order under MFMA)
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    v2 = prob.assemble_jacobian().csr()[2]
    assert_allclose(v2, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())


@pytest.mark.gpu
@pytest.mark.parametrize("n,layers", [(2, 2), (3, 4), (8, 8)])
def test_q4_action_matches_oracle_and_the_assembled_matrix(n, layers):
    """y = A u by the sum-factorised action wrapper (tp_action) against the oracle's dense element-matrix-times-vector,
    and against the MFMA-assembled matrix applied with the device SpMV (A*x == action(a, x), test_matrix_free.py:97-123)."""
    from firedrake_amd import op2
    m = fmesh.make_extruded_hex_mesh(n, layers, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m)
    assert prob.act_loop._prepare()["cw"].src.mode == "tp_action"
    y = np.array(prob.assemble_action().data_ro)
    ref = _oracle_action(m, prob.u.data_ro)
    assert_allclose(y, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    y2 = np.array(prob.assemble_action().data_ro)                   # tensor reuse: zero + INC again
    assert_allclose(y2, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    mat = prob.assemble_jacobian()
    t = op2.Dat(m.node_set)
    mat.mult(prob.u, t)
    assert_allclose(t.data_ro, y, rtol=0, atol=1e-11 * np.abs(ref).max())


@pytest.mark.gpu
def test_q4_direct_wrapper_fallback(monkeypatch):
    """FDHIP_TENSOR_WRAPPERS=0: the same parloops run the kernel's C text through the generic direct wrapper."""
    from firedrake_amd.configuration import configuration
    monkeypatch.setitem(configuration, "tensor_wrappers", 0)
    m = fmesh.make_extruded_hex_mesh(2, 2, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=True)
    assert prob.jac_loop._prepare()["cw"].src.mode == "direct"
    v = prob.assemble_jacobian().csr()[2]
    ref = _oracle_matrix(m, prob.bc_nodes)
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())


@pytest.mark.gpu
def test_q4_full_size_sampled_rows_against_oracle():
    """BASELINE.json configs[2] at n = 32 (32768 cells): the rows of the 27 cell-interior DoFs of a cell receive
    contributions from that cell alone, so they equal the corresponding rows of its element matrix -- checked against
    the oracle's dense kernel for 256 random cells (bottom, interior and top layers included), with BCs in place."""
    from test_forms_identities import _element_tensor
    n = 32
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=True)
    mat = prob.assemble_jacobian()
    sp = mat.sparsity
    rp = sp.rowptr
    rng = np.random.default_rng(5)
    cols_ = np.concatenate([rng.integers(0, m.base_set.size, 250), [0, 1, m.base_set.size - 1, 5, 7, 9]])
    lays = np.concatenate([rng.integers(0, n, 250), [0, n - 1, 0, n - 1, 1, n - 2]])
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    coords = np.array(m.coordinates.data_ro_with_halos)
    interior = [(a * 5 + b) * 5 + c for a in (1, 2, 3) for b in (1, 2, 3) for c in (1, 2, 3)]
    from firedrake_amd import _lib
    bc = np.zeros(m.node_set.total_size, dtype=bool)
    bc[prob.bc_nodes] = True
    worst = 0.0
    for col, lay in zip(cols_, lays):
        nodes = cm[col] + 4 * lay
        verts = coords[xm[col] + lay]
        Ae = _element_tensor(prob.kjac, 125, verts).reshape(125, 125)
        scale = np.abs(Ae).max()
        for i in interior:
            r = nodes[i]
            assert not bc[r]
            n0, n1 = int(rp[r]), int(rp[r + 1])
            cidx = np.empty(n1 - n0, dtype=np.int32)
            vals = np.empty(n1 - n0)
            _lib.call("fd_memcpy_d2h", cidx.ctypes.data, sp._colidx.ptr + 4 * n0, cidx.nbytes, None)
            _lib.call("fd_memcpy_d2h", vals.ctypes.data, mat._values_dev().ptr + 8 * n0, vals.nbytes, None)
            assert np.array_equal(np.sort(nodes), cidx)                # an interior DoF couples to its own cell only
            expect = np.where(bc[nodes], 0.0, Ae[i])                   # BC columns dropped
            got = vals[np.searchsorted(cidx, nodes)]
            worst = max(worst, np.abs(got - expect).max() / scale)
    assert worst <= 1e-11


@pytest.mark.gpu
def test_q4_full_size_shared_rows_against_oracle():
    """BASELINE.json configs[2] at n = 32: rows of SHARED DoFs -- cell vertices (8 cells), edge and face DoFs (4 / 2 cells),
    where the atomic scatter of several workgroups collides -- against the sum of the oracle's dense element tensors of every
    cell touching the DoF, with BCs in place (the cell-interior rows are the test above)."""
    from test_forms_identities import _element_tensor
    from firedrake_amd import _lib
    n = 32
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=True)
    mat = prob.assemble_jacobian()
    sp = mat.sparsity
    rp = sp.rowptr
    cm, xm = np.asarray(m.cell_node_map.values_with_halo), np.asarray(m.coord_map.values_with_halo)
    coords = np.array(m.coordinates.data_ro_with_halos)
    bc = np.zeros(m.node_set.total_size, dtype=bool)
    bc[prob.bc_nodes] = True
    # base node (bottom of its vertical line: node - 4*layer lies on a column's bottom cell) -> [(column, local index)]
    touching = {}
    for col in range(cm.shape[0]):
        for i, g in enumerate(cm[col]):
            touching.setdefault(int(g), []).append((col, i))
    rng = np.random.default_rng(11)
    local = {"vertex": [(a * 5 + b) * 5 + c for a in (0, 4) for b in (0, 4) for c in (0, 4)],
             "edge": [(0 * 5 + 0) * 5 + 2, (4 * 5 + 2) * 5 + 0, (2 * 5 + 4) * 5 + 4],
             "face": [(0 * 5 + 2) * 5 + 2, (2 * 5 + 2) * 5 + 4, (2 * 5 + 0) * 5 + 1]}
    tensors = {}

    def tensor(col, lay):
        if (col, lay) not in tensors:
            tensors[(col, lay)] = _element_tensor(prob.kjac, 125, coords[xm[col] + lay]).reshape(125, 125)
        return tensors[(col, lay)]

    worst, seen = 0.0, {1: 0, 2: 0, 4: 0, 8: 0}
    for kind, idxs in local.items():
        for _ in range(10):
            col, lay, i = int(rng.integers(0, cm.shape[0])), int(rng.integers(0, n)), int(rng.choice(idxs))
            r = int(cm[col, i] + 4 * lay)
            if bc[r]:
                continue
            # every (column', layer', i') with cm[column', i'] + 4*layer' == r
            cells = [(c2, l2, i2) for l2 in range(max(lay - 1, 0), min(lay + 2, n)) for (c2, i2) in touching.get(r - 4 * l2, [])]
            assert (col, lay, i) in cells
            seen[len(cells)] = seen.get(len(cells), 0) + 1
            expect = {}
            scale = 0.0
            for (c2, l2, i2) in cells:
                Ae = tensor(c2, l2)
                scale = max(scale, np.abs(Ae).max())
                for j, g in enumerate(cm[c2] + 4 * l2):
                    expect[int(g)] = expect.get(int(g), 0.0) + (0.0 if bc[g] else Ae[i2, j])
            n0, n1 = int(rp[r]), int(rp[r + 1])
            cidx, vals = np.empty(n1 - n0, dtype=np.int32), np.empty(n1 - n0)
            _lib.call("fd_memcpy_d2h", cidx.ctypes.data, sp._colidx.ptr + 4 * n0, cidx.nbytes, None)
            _lib.call("fd_memcpy_d2h", vals.ctypes.data, mat._values_dev().ptr + 8 * n0, vals.nbytes, None)
            assert np.array_equal(cidx, np.array(sorted(expect), dtype=np.int32))          # the row's pattern = union of its cells' nodes
            worst = max(worst, np.abs(vals - np.array([expect[g] for g in sorted(expect)])).max() / scale)
    assert seen.get(8, 0) > 0 and seen.get(4, 0) > 0 and seen.get(2, 0) > 0, seen
    assert worst <= 1e-11, worst


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


@pytest.mark.gpu
@pytest.mark.parametrize("degree,nq,n,layers", [(1, 2, 4, 5), (2, 3, 3, 4), (3, 4, 3, 3), (2, 4, 2, 3), (3, 3, 2, 2), (5, 6, 2, 2),
                                                (6, 7, 2, 2), (6, 8, 1, 3), (7, 9, 1, 2), (8, 10, 1, 2)])
def test_qk_tensor_wrappers_match_the_oracle(degree, nq, n, layers):
    """The tensor-product templates for other degrees / quadrature sizes (1, 2, 4 and 14 tiles per side of the element matrix;
    32, 14, 8 and 3 cells per action workgroup): MFMA matrix with BC lgmaps and sum-factorised action against the oracle's dense
    kernel, and A u == action(u) with the device SpMV.  Q6, Q7, Q8 (22, 32, 46 tiles per side): 16-row panels cut into column
    chunks of 8 tiles, one wavefront per (panel, chunk); with 8+ points per axis the point weights are computed plane by plane."""
    from firedrake_amd import op2
    m = fmesh.make_extruded_hex_mesh(n, layers, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, bcs=True, nq=nq)
    assert (prob.jac_loop._prepare()["cw"].src.mode, prob.act_loop._prepare()["cw"].src.mode) == ("tp_matrix", "tp_action")
    v = prob.assemble_jacobian().csr()[2]
    ref = _oracle_matrix(m, prob.bc_nodes, prob.kjac)
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    y = np.array(prob.assemble_action().data_ro)
    yref = _oracle_action(m, prob.u.data_ro, prob.kact)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())
    free = forms.HelmholtzHexProblem(m, bcs=False, nq=nq)
    t = op2.Dat(m.node_set)
    free.assemble_jacobian().mult(free.u, t)
    assert_allclose(t.data_ro, y, rtol=0, atol=1e-11 * np.abs(yref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("degree,nq", [(2, 3), (4, 5)])
def test_convection_diffusion_reaction_through_the_tensor_wrappers(degree, nq):
    """Non-symmetric point weights (alpha grad u . grad v + (b . grad u) v + beta u v) through the MFMA matrix and the
    sum-factorised action against the oracle's dense kernel."""
    from firedrake_amd import op2
    m = fmesh.make_extruded_hex_mesh(3, 3, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, nq=nq, alpha=0.7, beta=1.3, velocity=(1.0, -2.0, 0.5))
    assert (prob.jac_loop._prepare()["cw"].src.mode, prob.act_loop._prepare()["cw"].src.mode) == ("tp_matrix", "tp_action")
    mat = prob.assemble_jacobian()
    ref = _oracle_matrix(m, None, prob.kjac)
    assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    y = np.array(prob.assemble_action().data_ro)
    yref = _oracle_action(m, prob.u.data_ro, prob.kact)
    assert_allclose(y, yref, rtol=0, atol=1e-11 * np.abs(yref).max())
    t = op2.Dat(m.node_set)
    mat.mult(prob.u, t)
    assert_allclose(t.data_ro, y, rtol=0, atol=1e-11 * np.abs(yref).max())


@pytest.mark.gpu
@pytest.mark.parametrize("degree,nq,n,layers,bcs", [(4, 5, 2, 3, True), (2, 3, 3, 2, False), (3, 5, 2, 2, True)])
def test_coefficient_arguments_through_the_tensor_wrappers(degree, nq, n, layers, bcs):
    """a(du, v) = int kappa(w0) grad(du).grad(v) + c(u0) du v dx on Q_k hexahedra: the coefficient arguments of a TSFC Jacobian
    (tsfc/kernel_interface/firedrake_loopy.py:432-522) go through the fp64-MFMA matrix wrapper and the sum-factorised action
    wrapper -- NOT the direct fallback -- and match the oracle's dense kernel to 1e-11; A u == action(a, u)."""
    m = fmesh.make_extruded_hex_mesh(n, layers, degree, perturb=0.1)
    prob = forms.CoefficientHexProblem(m, bcs=bcs, nq=nq)
    mat = prob.assemble_jacobian()
    assert prob.jac_loop._prepared["cw"].src.mode == "tp_matrix"
    coefs = (prob.w0.data_ro, prob.u0.data_ro)
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac, coefs)
    _, _, v = mat.csr()
    assert_allclose(v, ref.values, rtol=0, atol=1e-11 * np.abs(ref.values).max())
    y = prob.assemble_action()
    assert prob.act_loop._prepared["cw"].src.mode == "tp_action"
    yref = _oracle_action(m, prob.u.data_ro, prob.kact, coefs)
    assert_allclose(y.data_ro, yref, rtol=0, atol=1e-11 * np.abs(yref).max())
    if not bcs:
        assert_allclose(ref.toscipy() @ np.asarray(prob.u.data_ro), yref, rtol=0, atol=1e-10 * np.abs(yref).max())
    # a coefficient that changes between calls (the Newton state) is seen by the next assembly
    prob.u0.data[:] = 0.0
    mat = prob.assemble_jacobian()
    ref0 = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac, (prob.w0.data_ro, prob.u0.data_ro))
    _, _, v0 = mat.csr()
    assert_allclose(v0, ref0.values, rtol=0, atol=1e-11 * np.abs(ref0.values).max())
    assert np.abs(v0 - v).max() > 1e-3 * np.abs(v).max()


@pytest.mark.gpu
@pytest.mark.parametrize("degree,nq,bcs", [(4, 5, True), (2, 3, False), (1, 2, True)])
def test_matrix_loop_after_zero_and_without(degree, nq, bcs):
    """The tensor-product matrix loop after Mat.zero() gives the oracle's values whatever the Mat held before; a second loop
    without zero() adds on top (Mat INC, mat.py:851-855); zeroing ahead of the loop (what the kernel-only timing does) shows a
    matrix of zeros to whoever looks in between."""
    m = fmesh.make_extruded_hex_mesh(3, 3, degree, perturb=0.1)
    prob = forms.HelmholtzHexProblem(m, bcs=bcs, nq=nq)
    ref = _oracle_matrix(m, prob.bc_nodes if bcs else None, prob.kjac)
    tol = 1e-11 * np.abs(ref.values).max()
    loop, mat = prob.jac_loop, prob.mat
    prob.assemble_jacobian()
    prob.assemble_jacobian()                  # the second time over stale values
    v = mat.csr()[2]
    assert_allclose(v, ref.values, rtol=0, atol=tol)
    loop()                                    # no zero(): accumulates
    twice = mat.csr()[2]
    bcrows = np.zeros(len(v), dtype=bool)
    if bcs:
        rp, ci = ref.rowptr, ref.colidx
        for b in prob.bc_nodes:
            bcrows[rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b)] = True
    assert_allclose(twice[~bcrows], 2 * ref.values[~bcrows], rtol=0, atol=2 * tol)
    mat.zero()
    loop.zero_ahead()
    assert not mat.csr()[2].any()
    loop()
    got = mat.csr()[2]
    assert_allclose(got[~bcrows], ref.values[~bcrows], rtol=0, atol=tol)
    import torch
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    prob.assemble_jacobian(events=ev)
    assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=tol)
