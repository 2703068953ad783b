"""CPU: the hand-restated TSFC-equivalent local kernels satisfy the analytic identities the
reference's regression tests rely on (tests/firedrake/regression/test_assemble.py:60-73,
test_poisson_strong_bcs.py:74-92, test_matrix_free.py:97-123) -- evaluated through the oracle --
and the synthetic mesh generators produce the layouts of SURVEY.md Appendix C."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import ODat, OMat, READ, INC
from firedrake_amd import forms, mesh as fmesh


def _mesh(dim, n, degree, perturb=0.15):
    if dim == 2:
        return fmesh.UnitSquareMesh(n, n, degrees=(degree,), tile=(4, 4), perturb=perturb)
    return fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(2, 2, 2), perturb=perturb)


def _assemble(m, degree):
    V, X = m.space(degree), m.coord_space
    cm, xm = V.cell_node_map.values_with_halo, X.cell_node_map.values_with_halo
    nn = V.node_set.total_size
    coords = np.array(m.coordinates.data_ro_with_halos)
    K = oracle.build_sparsity(nn, nn, [(cm, cm)])
    kj = forms.poisson_jacobian_kernel(m.gdim, degree)
    oracle.par_loop(kj.code, kj.name, 0, m.cell_set.size, [OMat(K, INC, cm, cm), ODat(coords, READ, xm)])
    M = oracle.build_sparsity(nn, nn, [(cm, cm)])
    km = forms.mass_kernel(m.gdim, degree)
    oracle.par_loop(km.code, km.name, 0, m.cell_set.size, [OMat(M, INC, cm, cm), ODat(coords, READ, xm)])
    return V, cm, xm, coords, K.toscipy(), M.toscipy()


@pytest.mark.parametrize("dim,degree,n", [(2, 1, 6), (2, 2, 5), (3, 1, 4), (3, 2, 3)])
def test_stiffness_and_mass_identities(dim, degree, n):
    m = _mesh(dim, n, degree)
    V, cm, xm, coords, K, M = _assemble(m, degree)
    nn = V.node_set.total_size
    assert nn == V.global_dofs == (degree * n + 1) ** dim
    amax = abs(K).max()
    assert_allclose(K @ np.ones(nn), 0, atol=1e-12 * amax)          # constants in the null space
    assert abs(K - K.T).max() <= 1e-13 * amax                       # symmetric
    assert_allclose(M.sum(), 1.0, rtol=1e-12)                       # sum(M) = |Omega|
    # residual with f = 0 equals K u  (A*x == action(a, x))
    pts = V.node_points
    u = np.sin(2 * pts[:, 0]) + pts[:, 1] ** 2
    r = np.zeros(nn)
    kr = forms.poisson_residual_kernel(dim, degree)
    oracle.par_loop(kr.code, kr.name, 0, m.cell_set.size,
                    [ODat(r, INC, cm), ODat(coords, READ, xm), ODat(u.copy(), READ, cm), ODat(np.zeros(nn), READ, cm)])
    assert_allclose(r, K @ u, atol=1e-12 * max(1, abs(K @ u).max()))
    # load term: F(0) = -M f
    f = np.cos(pts[:, 0]) * (1 + pts[:, 1])
    r2 = np.zeros(nn)
    oracle.par_loop(kr.code, kr.name, 0, m.cell_set.size,
                    [ODat(r2, INC, cm), ODat(coords, READ, xm), ODat(np.zeros(nn), READ, cm), ODat(f.copy(), READ, cm)])
    assert_allclose(r2, -(M @ f), atol=1e-12)


@pytest.mark.parametrize("dim,n", [(2, 6), (3, 3)])
def test_p2_residual_vanishes_for_exact_quadratic_solution(dim, n):
    """-lap(u) = f with u = x^2 (in P2), f = -2: interior residual rows are zero (cf. test_poisson_strong_bcs.py)."""
    m = _mesh(dim, n, 2, perturb=0.0)
    V, X = m.space(2), m.coord_space
    cm, xm = V.cell_node_map.values_with_halo, X.cell_node_map.values_with_halo
    nn = V.node_set.total_size
    pts = V.node_points
    r = np.zeros(nn)
    kr = forms.poisson_residual_kernel(dim, 2)
    oracle.par_loop(kr.code, kr.name, 0, m.cell_set.size,
                    [ODat(r, INC, cm), ODat(np.array(m.coordinates.data_ro_with_halos), READ, xm),
                     ODat(pts[:, 0] ** 2, READ, cm), ODat(np.full(nn, -2.0), READ, cm)])
    interior = np.setdiff1d(np.arange(nn), V.boundary_nodes)
    assert len(interior) > 0
    assert_allclose(r[interior], 0, atol=1e-12)


def test_unit_cube_mesh_layout():
    n = 5
    m = fmesh.UnitCubeMesh(n, degrees=(1, 2), tile=(2, 2, 2))
    assert m.cell_set.size == 6 * n ** 3 == m.ncells_global
    V1, V2 = m.space(1), m.space(2)
    assert V1.node_set.size == (n + 1) ** 3 and V2.node_set.size == (2 * n + 1) ** 3
    c = V1.cell_node_map.values
    assert c.shape == (6 * n ** 3, 4) and c.min() == 0 and c.max() == (n + 1) ** 3 - 1
    # every tet has volume 1/(6 n^3)
    X = np.array(m.coordinates.data_ro)[c]
    vol = np.abs(np.linalg.det(X[:, 1:] - X[:, :1])) / 6
    assert_allclose(vol, 1 / (6 * n ** 3), rtol=1e-12)
    # CG2: vertex dofs coincide with the CG1 points, edge dofs are midpoints (UFC edge order)
    c2 = V2.cell_node_map.values
    P2 = V2.node_points[c2]
    assert_allclose(P2[:, :4], X, atol=1e-14)
    for e, (a, b) in enumerate([(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)]):
        assert_allclose(P2[:, 4 + e], 0.5 * (X[:, a] + X[:, b]), atol=1e-14)
    # numbering locality: nodes of consecutive cells are close (cell-traversal numbering)
    assert np.median(c.max(axis=1) - c.min(axis=1)) < 4 * (n + 1) ** 2
    # nnz of the P1 operator: nV + 2 nE with nE = 7n^3 + 9n^2 + 3n (SURVEY.md 8 sizes)
    sp = oracle.build_sparsity((n + 1) ** 3, (n + 1) ** 3, [(V1.cell_node_map.values, V1.cell_node_map.values)])
    assert len(sp.colidx) == (n + 1) ** 3 + 2 * (7 * n ** 3 + 9 * n ** 2 + 3 * n)


def test_unit_square_mesh_layout():
    m = fmesh.UnitSquareMesh(64, 64)
    V = m.space(1)
    assert m.cell_set.size == 8192 and V.node_set.size == 4225              # config C1 sizes
    cm = V.cell_node_map.values
    sp = oracle.build_sparsity(4225, 4225, [(cm, cm)])
    assert len(sp.colidx) == 29057                                           # nV + 2 nE
    X = np.array(m.coordinates.data_ro)[cm]
    e1, e2 = X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]
    area = 0.5 * np.abs(e1[:, 0] * e2[:, 1] - e1[:, 1] * e2[:, 0])
    assert_allclose(area, 1 / 8192, rtol=1e-12)


@pytest.mark.parametrize("nranks", [2, 3])
def test_partitioned_cube_halo_lists_are_consistent(nranks):
    n = 6
    parts = [fmesh.UnitCubeMesh(n, degrees=(1, 2), tile=(2, 2, 2), rank=r, nranks=nranks) for r in range(nranks)]
    assert sum(p.cell_set.size for p in parts) == 6 * n ** 3
    for deg in (1, 2):
        assert sum(p.space(deg).node_set.size for p in parts) == (deg * n + 1) ** 3
        for r, p in enumerate(parts):
            V = p.space(deg)
            for nb, recv in V.halo.recv.items():
                send = parts[nb].space(deg).halo.send[r]
                assert len(send) == len(recv)
                # matching order: same physical points
                assert_allclose(parts[nb].space(deg).node_points[send], V.node_points[recv], atol=1e-14)
                assert recv.min() >= V.node_set.size            # ghosts live at the tail
                assert send.max() < parts[nb].space(deg).node_set.size
            # executed cells only reference local nodes; core cells reference no ghost
            cm = V.cell_node_map.values_with_halo
            assert cm[:p.cell_set.core_size].max() < V.node_set.size
            assert cm.max() < V.node_set.total_size


def _element_tensor(kernel, nd, coords):
    """One call of a local kernel on a single element, through the oracle's wrapper (a one-cell mesh)."""
    nv = len(coords)
    cm = np.arange(nd, dtype=np.int32).reshape(1, nd)
    xm = np.arange(nv, dtype=np.int32).reshape(1, nv)
    A = oracle.build_sparsity(nd, nd, [(cm, cm)])
    oracle.par_loop(kernel.code, kernel.name, 0, 1, [OMat(A, INC, cm, cm), ODat(np.ascontiguousarray(coords, dtype=float), READ, xm)])
    return A.todense()


def test_p1_element_tensors_are_the_textbook_ones():
    """Known answers for the hand-restated local kernels (no reference test stores element tensors -- SURVEY.md 8c): the
    P1 stiffness and mass matrices of the unit right triangle / tetrahedron, and their affine scaling laws."""
    tri = np.array([[0, 0], [1, 0], [0, 1]], dtype=float)
    tet = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1]], dtype=float)
    K2 = _element_tensor(forms.poisson_jacobian_kernel(2, 1), 3, tri)
    assert_allclose(K2, 0.5 * np.array([[2, -1, -1], [-1, 1, 0], [-1, 0, 1]]), atol=1e-15)
    M2 = _element_tensor(forms.mass_kernel(2, 1), 3, tri)
    assert_allclose(M2, (np.ones((3, 3)) + np.eye(3)) / 24.0, atol=1e-15)
    K3 = _element_tensor(forms.poisson_jacobian_kernel(3, 1), 4, tet)
    assert_allclose(K3, np.array([[3, -1, -1, -1], [-1, 1, 0, 0], [-1, 0, 1, 0], [-1, 0, 0, 1]]) / 6.0, atol=1e-15)
    M3 = _element_tensor(forms.mass_kernel(3, 1), 4, tet)
    assert_allclose(M3, (np.ones((4, 4)) + np.eye(4)) / 120.0, atol=1e-15)
    # x -> s*x: mass scales with s^d, stiffness with s^(d-2); a rotation leaves both unchanged
    s = 0.37
    assert_allclose(_element_tensor(forms.mass_kernel(3, 1), 4, s * tet), s ** 3 * M3, rtol=1e-13)
    assert_allclose(_element_tensor(forms.poisson_jacobian_kernel(3, 1), 4, s * tet), s * K3, rtol=1e-13)
    assert_allclose(_element_tensor(forms.poisson_jacobian_kernel(2, 1), 3, s * tri), K2, rtol=1e-13)
    th = 0.7
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    assert_allclose(_element_tensor(forms.poisson_jacobian_kernel(2, 1), 3, tri @ R.T + 3.0), K2, atol=1e-14)


def test_p2_element_mass_matrix_is_the_textbook_one():
    """P2 on the unit right triangle (vertices, then the edges opposite to them -- FIAT's order): the classical
    (1/360) mass matrix; vertices couple to their two adjacent edges with 0 and to the opposite edge with -4."""
    tri = np.array([[0, 0], [1, 0], [0, 1]], dtype=float)
    M = _element_tensor(forms.mass_kernel(2, 2), 6, tri) * 360.0
    vv = np.array([[6, -1, -1], [-1, 6, -1], [-1, -1, 6]])
    ee = np.array([[32, 16, 16], [16, 32, 16], [16, 16, 32]])
    ve = np.array([[-4, 0, 0], [0, -4, 0], [0, 0, -4]])                 # vertex i vs the edge opposite to vertex i
    assert_allclose(M, np.block([[vv, ve], [ve.T, ee]]), atol=1e-12)


def test_q4_hex_element_matrix_is_the_kronecker_product_of_exact_1d_matrices():
    """Known answer for the C3 local kernel (which the fp64-MFMA kernel is compared with on the GPU): on the box
    [0,a]x[0,b]x[0,c] the Q4 Helmholtz matrix is K1(a) (x) M1(b) (x) M1(c) + M1 (x) K1 (x) M1 + M1 (x) M1 (x) K1 +
    M1 (x) M1 (x) M1, with the 1-D CG4 mass / stiffness matrices on GLL nodes integrated EXACTLY here by polynomial
    algebra (independent of the kernel's quadrature tables; 5 Gauss points integrate degree 8 exactly)."""
    from numpy.polynomial import polynomial as P
    nodes = fmesh._gll_nodes(4)
    basis = []
    for i in range(5):
        p = np.array([1.0])
        for m in range(5):
            if m != i:
                p = P.polymul(p, np.array([-nodes[m], 1.0]) / (nodes[i] - nodes[m]))
        basis.append(p)

    def integ(p):
        q = P.polyint(p)
        return P.polyval(1.0, q) - P.polyval(0.0, q)
    M1 = np.array([[integ(P.polymul(bi, bj)) for bj in basis] for bi in basis])
    K1 = np.array([[integ(P.polymul(P.polyder(bi), P.polyder(bj))) for bj in basis] for bi in basis])
    a, b, c = 0.7, 1.3, 0.45
    exact = (np.kron(np.kron(K1 / a, M1 * b), M1 * c) + np.kron(np.kron(M1 * a, K1 / b), M1 * c)
             + np.kron(np.kron(M1 * a, M1 * b), K1 / c) + np.kron(np.kron(M1 * a, M1 * b), M1 * c))
    verts = np.array([[x * a, y * b, z * c] for x in (0, 1) for y in (0, 1) for z in (0, 1)], dtype=float)   # v = 4x+2y+z
    A = _element_tensor(forms.helmholtz_q4_hex_jacobian_kernel(), 125, verts)
    assert_allclose(A, exact, atol=1e-12 * np.abs(exact).max())


@pytest.mark.parametrize("dim", [2, 3])
def test_simplex_rules_are_exact_to_their_degree_with_positive_interior_points(dim):
    """forms.simplex_rule: every rule integrates all monomials up to its degree to rounding, has positive weights and points
    strictly inside the simplex; the symmetric rules (6 points / degree 4 on the triangle, 14 points / degree 5 on the
    tetrahedron -- solved from the moment equations at import) are smaller than the collapsed Gauss-Jacobi rules they replace."""
    import math
    for degree in range(1, 8):
        pts, wts = forms.simplex_rule(dim, degree)
        assert wts.min() > 0 and pts.min() > 0 and pts.sum(axis=1).max() < 1
        for p in np.ndindex(*(degree + 1,) * dim):
            if sum(p) <= degree:
                exact = math.prod(math.factorial(e) for e in p) / math.factorial(sum(p) + dim)
                assert abs((wts * np.prod(pts ** np.array(p), axis=1)).sum() - exact) < 2e-15
    assert len(forms.simplex_rule(2, 4)[1]) == 6 and len(forms.simplex_rule(3, 4)[1]) == 14 == len(forms.simplex_rule(3, 5)[1])
    assert len(forms.gauss_jacobi_simplex(3, 4)[1]) == 27


@pytest.mark.parametrize("degree,nq", [(1, 2), (2, 3), (3, 4), (2, 4)])
def test_qk_hex_element_matrix_rows_sum_to_the_mass_action(degree, nq):
    """The dense Q_k Helmholtz kernel (the oracle's definition for the tensor wrappers of every degree): constants are in the
    space, so A_e 1 = M_e 1 and 1^T A_e 1 = |cell|; symmetric."""
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import mesh as fmesh
    m = fmesh.make_extruded_hex_mesh(1, 1, degree, perturb=0.0)
    k = forms.helmholtz_hex_jacobian_kernel(degree, nq)
    nd = (degree + 1) ** 3
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    csr = oracle.build_sparsity(nd, nd, [(cm, cm, m.layers, m.cell_node_map.offset, m.cell_node_map.offset)])
    oracle.par_loop(k.code, k.name, 0, 1, [OMat(csr, INC, cm, cm, roffset=m.cell_node_map.offset, coffset=m.cell_node_map.offset),
                                           ODat(np.array(m.coordinates.data_ro_with_halos), READ, xm, offset=m.coord_map.offset)],
                    layers=(0, m.layers + 1))
    A = csr.toscipy().toarray()
    assert abs(A - A.T).max() < 1e-14
    assert abs(A.sum() - 1.0) < 1e-13


@pytest.mark.parametrize("dim", [2, 3])
def test_symmetric_rule_parameters_are_the_solution_of_the_moment_equations(dim):
    """The literals behind forms.symmetric_simplex_rule against their derivation (Gauss-Newton on the moment equations)."""
    assert_allclose(forms.solve_symmetric_rule(dim), forms._SYMMETRIC_RULES[dim][1], rtol=0, atol=5e-15)
