"""The variational problems of the reference's regression tests, stated with this repository's kernels, meshes and
assembly path, so that the thresholds THOSE tests assert (tests/golden/reference_thresholds.json, lifted from the reference
sources by tests/golden/make_thresholds.py) can be checked here:

* tests/firedrake/regression/test_helmholtz.py:23-53      -div grad u + u = f, L2 error against the analytic solution
* tests/firedrake/extrusion/test_helmholtz_scalar.py:8-33 the same in 3-D on extruded cells (here: Q4 hexahedra)
* tests/firedrake/regression/test_poisson_strong_bcs.py:20-64  Laplace with strong Dirichlet data, u = 42 y

``backend="gpu"``: every form is assembled by HIP wrapper kernels through op2.par_loop and the system is solved by the
device CG of tests/device_cg.py.  ``backend="oracle"``: the same parloops run through the CPU oracle (same kernel text)
and scipy solves -- this pins the element tensors of the shared C text to the reference-held criteria on machines without
a GPU; the GPU-vs-oracle parity tests then tie the HIP path to it."""
import numpy as np

from firedrake_amd import forms, mesh as fmesh, op2


def _run(backend, kernel, iterset, *args):
    """op2.par_loop on the device, or the oracle writing its results back into the carriers."""
    if backend == "gpu":
        op2.par_loop(kernel, iterset, *args)
        return
    from helpers import oracle_run
    outs = oracle_run(kernel, iterset, *args)
    for a, o in zip(args, outs):
        if a.access == op2.READ:
            continue
        if hasattr(a.data, "sparsity"):
            a.data._oracle_csr = o
        else:
            a.data.data_with_halos[...] = o


def _matrix(backend, mat):
    if backend == "gpu":
        return mat.toscipy().tocsr()
    return mat._oracle_csr.toscipy().tocsr()


def _solve(backend, mat, b, x0=None, rtol=1e-13):
    """x with A x = b; returns a numpy vector"""
    if backend == "gpu":
        from device_cg import cg
        x = op2.Dat(b.dataset, x0)
        it, res = cg(mat, b, x, rtol=rtol)
        assert res <= 10 * rtol, (it, res)
        return np.array(x.data_ro)
    import scipy.sparse as ssp
    import scipy.sparse.linalg as sla
    A = _matrix(backend, mat)
    bb = np.array(b.data_ro)
    x, info = sla.cg(A, bb, x0=x0, rtol=rtol, atol=0.0, maxiter=20000, M=ssp.diags(1.0 / A.diagonal()))
    assert info == 0, info
    return x


def helmholtz_simplex(backend, dim, degree, n):
    """test_helmholtz.py:23-53 on UnitSquareMesh(n, n) / the 3-D analogue of test_helmholtz_scalar.py:8-33 on
    UnitCubeMesh(n): returns the L2 error sqrt(assemble((sol - expect)^2 dx))."""
    m = fmesh.UnitSquareMesh(n, n, degrees=(degree,)) if dim == 2 else fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(4, 4, 4))
    V, X = m.space(degree), m.coord_space
    cm, xm = V.cell_node_map, X.cell_node_map
    p = V.node_points
    if dim == 2:
        expect = np.cos(2 * np.pi * p[:, 0]) * np.cos(2 * np.pi * p[:, 1])
        fv = (1 + 8 * np.pi ** 2) * expect
    else:
        expect = np.cos(2 * np.pi * p[:, 0]) * np.cos(2 * np.pi * p[:, 1]) * np.cos(2 * np.pi * p[:, 2])
        fv = (1 + 12 * np.pi ** 2) * expect
    f, b = V.dat(1, fv), V.dat(1)
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    A, M = op2.Mat(sp), op2.Mat(sp)
    _run(backend, forms.helmholtz_kernel(dim, degree), m.cell_set, A(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm))
    _run(backend, forms.mass_kernel(dim, degree), m.cell_set, M(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm))
    _run(backend, forms.rhs_kernel(dim, degree), m.cell_set, b(op2.INC, cm), m.coordinates(op2.READ, xm), f(op2.READ, cm))
    sol = _solve(backend, A, b)
    e = sol - expect
    return float(np.sqrt(e @ (_matrix(backend, M) @ e)))


def helmholtz_hex(backend, degree, n):
    """The extruded-hexahedra Helmholtz problem (test_helmholtz_scalar.py:8-33: ExtrudedMesh(UnitSquareMesh(n, n,
    quadrilateral=True), n), CG_degree x CG_degree, f and the exact solution interpolated into the space): the operator through
    the fp64-MFMA matrix wrapper, right-hand side and error norm through the sum-factorised MASS action (degree + 1 Gauss
    points per axis: exact for the mass form on an affine cell)."""
    m = fmesh.make_extruded_hex_mesh(n, n, degree, perturb=0.0)
    cm, xm = m.cell_node_map, m.coord_map
    p = m.node_points
    expect = np.cos(2 * np.pi * p[:, 0]) * np.cos(2 * np.pi * p[:, 1]) * np.cos(2 * np.pi * p[:, 2])
    f = op2.Dat(m.node_set, (1 + 12 * np.pi ** 2) * expect)
    b = op2.Dat(m.node_set)
    sp = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, None)])
    A = op2.Mat(sp)
    _run(backend, forms.helmholtz_hex_jacobian_kernel(degree), m.cell_set, A(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm))
    kmass = forms.helmholtz_hex_action_kernel(degree, None, f"mass_q{degree}_hex_action", 0.0, 1.0)
    _run(backend, kmass, m.cell_set, b(op2.INC, cm), m.coordinates(op2.READ, xm), f(op2.READ, cm))
    sol = _solve(backend, A, b)
    e = op2.Dat(m.node_set, sol - expect)
    Me = op2.Dat(m.node_set)
    _run(backend, kmass, m.cell_set, Me(op2.INC, cm), m.coordinates(op2.READ, xm), e(op2.READ, cm))
    return float(np.sqrt(np.array(e.data_ro) @ np.array(Me.data_ro)))


def helmholtz_q4_hex(backend, n):
    """Config C3's element on that problem."""
    return helmholtz_hex(backend, 4, n)


def poisson_strong_bcs(backend, degree, r=2, newton=False):
    """test_poisson_strong_bcs.py:20-64: a = inner(grad u, grad v) dx on UnitSquareMesh(2^r, 2^r), u = 0 on y = 0 (side 3)
    and u = 42 on y = 1 (side 4), natural conditions elsewhere; exact solution 42 y.  ``newton``: the nonlinear form
    ``solve(a == 0, u)`` (run_test) -- one Newton step from the BC-satisfying initial guess; else run_test_linear.  Both
    go through the residual / Jacobian pair with BC rows zeroed and BC rows and columns masked (assemble.py:1243-1267,
    2075-2108), i.e. the benchmark's own PoissonProblem."""
    n = 2 ** r
    m = fmesh.UnitSquareMesh(n, n, degrees=(degree,))
    V = m.space(degree)
    p = V.node_points
    lo, hi = np.nonzero(p[:, 1] < 1e-12)[0], np.nonzero(p[:, 1] > 1 - 1e-12)[0]
    bc = np.concatenate([lo, hi]).astype(np.int32)
    prob = forms.PoissonProblem(m, degree, bcs=True, bc_nodes=bc)
    u0 = np.zeros(len(p))
    u0[hi] = 42.0                                         # bc.apply(u): the initial guess satisfies the conditions
    prob.u.data[...] = u0
    prob.f.data[...] = 0.0
    cm, xm = V.cell_node_map, m.coord_space.cell_node_map
    if backend == "gpu":
        prob.assemble_residual()
        prob.assemble_jacobian()
        J = prob.jacobian()[0]
    else:
        _run(backend, prob.kres, m.cell_set, prob.r(op2.INC, cm), m.coordinates(op2.READ, xm), prob.u(op2.READ, cm), prob.f(op2.READ, cm))
        prob.r.data[bc] = 0.0
        J, loop = prob.jacobian()
        lg = loop.arguments[0].lgmaps
        _run(backend, prob.kjac, m.cell_set, J(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm))
        csr = J._oracle_csr
        rp, ci = csr.rowptr, csr.colidx
        for row in bc:                                    # assemble.py:1501-1507: unit diagonal on the BC rows
            k = rp[row] + np.searchsorted(ci[rp[row]:rp[row + 1]], row)
            csr.values[k] = 1.0
    rhs = op2.Dat(V.node_set, -np.array(prob.r.data_ro))
    du = _solve(backend, J, rhs)
    sol = u0 + du
    M = op2.Mat(J.sparsity)
    _run(backend, forms.mass_kernel(2, degree), m.cell_set, M(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm))
    e = sol - 42.0 * p[:, 1]
    return float(np.sqrt(abs(e @ (_matrix(backend, M) @ e))))
