"""GPU parity of the benchmark forms (configs C1/C2/C5 shapes at oracle-friendly sizes): the HIP
path (assemble-shaped front end, staged wrappers, device CSR, BC masking) against the oracle."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

import oracle
from oracle import ODat, OMat, READ, INC
from firedrake_amd import forms, mesh as fmesh, op2
from firedrake_amd.configuration import configuration

pytestmark = pytest.mark.gpu


def _oracle_problem(prob, with_bcs):
    m, deg = prob.mesh, prob.degree
    V, X = prob.V, m.coord_space
    cm, xm = V.cell_node_map.values_with_halo, X.cell_node_map.values_with_halo
    nn = V.node_set.total_size
    coords = np.array(m.coordinates.data_ro_with_halos)
    r = np.zeros(nn)
    oracle.par_loop(prob.kres.code, prob.kres.name, 0, m.cell_set.size,
                    [ODat(r, INC, cm), ODat(coords, READ, xm), ODat(np.array(prob.u.data_ro_with_halos), READ, cm),
                     ODat(np.array(prob.f.data_ro_with_halos), READ, cm)])
    csr = oracle.build_sparsity(nn, nn, [(cm, cm)])
    lg = None
    if with_bcs:
        lg = np.arange(nn, dtype=np.int32)
        lg[prob.bc_nodes] = -1
        r[prob.bc_nodes] = 0.0
    oracle.par_loop(prob.kjac.code, prob.kjac.name, 0, m.cell_set.size,
                    [OMat(csr, INC, cm, cm, row_lgmap=lg, col_lgmap=lg), ODat(coords, READ, xm)])
    A = csr.toscipy().tolil()
    if with_bcs:
        for b in prob.bc_nodes:
            A[b, b] = 1.0
    return r, A.tocsr()


@pytest.mark.parametrize("dim,degree,n", [(2, 1, 64), (3, 1, 12), (3, 2, 6), (2, 2, 16)])
@pytest.mark.parametrize("bcs", [False, True])
@pytest.mark.parametrize("ocr", [1, 0])
def test_poisson_residual_and_jacobian(dim, degree, n, bcs, ocr, monkeypatch):
    monkeypatch.setitem(configuration, "mat_ocr", ocr)
    m = (fmesh.UnitSquareMesh(n, n, degrees=(degree,), perturb=0.1) if dim == 2
         else fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(4, 4, 2), perturb=0.1))
    prob = forms.PoissonProblem(m, degree, bcs=bcs)
    r = prob.assemble_residual()
    mat = prob.assemble_jacobian()
    ro, Ao = _oracle_problem(prob, bcs)
    # stated fp64 tolerance: 1e-12 relative to the largest entry
    assert_allclose(r.data_ro, ro, rtol=0, atol=1e-12 * max(1.0, np.abs(ro).max()))
    A = mat.toscipy()
    assert np.array_equal(A.indptr, Ao.indptr) and np.array_equal(A.indices, Ao.indices)
    assert_allclose(A.data, Ao.data, rtol=0, atol=1e-12 * np.abs(Ao.data).max())
    # second assembly into the same tensors gives the same answer (tensor reuse, assemble.py:1042-1047)
    r2 = np.array(prob.assemble_residual().data_ro)
    assert_allclose(r2, ro, rtol=0, atol=1e-12 * max(1.0, np.abs(ro).max()))
    if not bcs:
        # A*u - M f == F(u)  i.e. matrix-free action identity with the assembled operator
        y = prob.V.dat()
        mat.mult(prob.u, y)
        f0 = forms.PoissonProblem(m, degree, bcs=False)
        f0.u.assign(0.0)
        load = np.array(f0.assemble_residual().data_ro)
        assert_allclose(y.data_ro + load, ro, rtol=0, atol=1e-10 * max(1.0, np.abs(ro).max()))


def test_bc_column_masking_fused_and_accumulating():
    """Owner-computes-rows with BC lgmaps: (i) assembled from zero (pending Mat.zero() fused into the loop) -> columns
    masked in the kernel; (ii) a second loop WITHOUT Mat.zero() accumulates (MatSetValuesLocal ADD_VALUES,
    builder.py:573-625): exactly twice the dropped-column matrix, BC diagonal untouched by the loop."""
    m = fmesh.UnitCubeMesh(10, degrees=(1,), tile=(4, 4, 2), perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    _, Ao = _oracle_problem(prob, True)
    A1 = prob.assemble_jacobian().toscipy()
    assert_allclose(A1.data, Ao.data, rtol=0, atol=1e-12 * np.abs(Ao.data).max())
    mat, loop = prob.jacobian()
    loop()                                   # no zero(): accumulate a second copy of the masked operator
    A2 = mat.toscipy()
    bc = np.zeros(prob.V.node_set.total_size, dtype=bool)
    bc[prob.bc_nodes] = True
    rows = np.repeat(np.arange(A2.shape[0]), np.diff(A2.indptr))
    masked = bc[rows] | bc[A2.indices]
    expect = np.where(masked, A1.data, 2.0 * A1.data)     # masked entries: diagonal 1 / zeros stay as they were
    assert_allclose(A2.data, expect, rtol=0, atol=1e-12 * np.abs(Ao.data).max())


def test_c1_config_sizes_and_direct_mode(monkeypatch):
    """BASELINE.json configs[0]: Poisson CG1 on UnitSquareMesh(64,64), both wrapper shapes."""
    m = fmesh.UnitSquareMesh(64, 64)
    for mode in ("auto", "direct"):
        monkeypatch.setitem(configuration, "mode", mode)
        prob = forms.PoissonProblem(m, 1, bcs=True)
        r = prob.assemble_residual()
        mat = prob.assemble_jacobian()
        ro, Ao = _oracle_problem(prob, True)
        assert mat.sparsity.nz == 29057
        assert_allclose(r.data_ro, ro, rtol=0, atol=1e-12 * np.abs(ro).max())
        assert_allclose(mat.toscipy().data, Ao.data, rtol=0, atol=1e-12 * np.abs(Ao.data).max())


def test_medium_cube_properties():
    """Size-independent properties at a size the oracle would take too long for: K*1 = 0, symmetry,
    repeated assembly idempotent (zero + assemble)."""
    m = fmesh.UnitCubeMesh(48, degrees=(1,), perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=False)
    mat = prob.assemble_jacobian()
    A = mat.toscipy()
    amax = np.abs(A.data).max()
    assert np.abs(A @ np.ones(A.shape[0])).max() <= 1e-11 * amax
    assert abs(A - A.T).max() <= 1e-12 * amax
    A2 = prob.assemble_jacobian().toscipy()
    assert_allclose(A2.data, A.data, rtol=0, atol=1e-12 * amax)
    ones = prob.V.dat(1, np.ones(prob.V.node_set.total_size))
    prob.u.assign(1.0)
    prob.f.assign(0.0)
    r = prob.assemble_residual()
    assert np.abs(r.data_ro).max() <= 1e-11 * amax
    del ones


def test_c1_step_as_hipgraph():
    """Config C1 (launch-bound): the whole assembly step replayed from a hipGraph gives the same tensors."""
    from firedrake_amd.graph import CapturedStep
    m = fmesh.UnitSquareMesh(64, 64, perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    ro, Ao = _oracle_problem(prob, True)

    def step():
        prob.assemble_residual()
        prob.assemble_jacobian()

    g = CapturedStep(step)
    prob.r.zero()
    prob.jacobian()[0].zero()
    prob.jacobian()[0]._values_dev()            # flush the pending zero: the tensors really are cleared now
    for _ in range(3):
        g()
    # no sync, no invalidation by hand: a replay declares the host copies of what the step writes stale, and the download waits for it
    assert_allclose(prob.r.data_ro, ro, rtol=0, atol=1e-12 * np.abs(ro).max())
    assert_allclose(prob.jacobian()[0].toscipy().data, Ao.data, rtol=0, atol=1e-12 * np.abs(Ao.data).max())
    # the state rewritten ON THE HOST between two replays reaches the device (the replay uploads what the step reads when it is stale)
    u0 = np.array(prob.u.data_ro)
    prob.u.data[...] = 0.0
    prob.f.data[...] = 0.0
    g()
    assert np.abs(prob.r.data_ro).max() <= 1e-13 * np.abs(ro).max()
    prob.u.data[...] = u0
    del u0


def test_release_of_device_memory_inside_a_capture_is_parked():
    """A finaliser can run at any moment of a captured step (the garbage collector), and hipFree there would invalidate the capture
    ("operation failed due to a previous error during capture" at the next launch).  The library parks releases that arrive between
    fd_graph_begin and fd_graph_end and carries them out at the end; an ALLOCATION inside a capture is refused with a message, and
    the capture survives it; captures do not nest."""
    import ctypes
    from firedrake_amd import _lib
    from firedrake_amd.device import DeviceBuffer
    from firedrake_amd.graph import CapturedStep
    m = fmesh.UnitSquareMesh(16, 16, perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    ro, Ao = _oracle_problem(prob, True)
    state = {"garbage": None, "old": None}

    def step():
        prob.assemble_residual()
        if state["garbage"] is not None:         # (only the captured call finds something to drop)
            state["garbage"] = None              # four buffers released inside the capture (DeviceBuffer.__del__ -> fd_free)
            state["old"] = None                  # and a whole graph: exec, graph, stream (fd_graph_free)
            with pytest.raises(_lib.FDHipError, match="fd_malloc inside"):
                DeviceBuffer(64)
            h = ctypes.c_void_p()
            with pytest.raises(_lib.FDHipError, match="do not nest"):
                _lib.call("fd_graph_begin", ctypes.byref(h))
        prob.assemble_jacobian()

    step(); step()                               # warm: plans exist
    state["garbage"] = [DeviceBuffer(1 << 20) for _ in range(4)]
    state["old"] = CapturedStep(lambda: prob.assemble_residual())
    g = CapturedStep(step, warmup=0)
    assert state["garbage"] is None and state["old"] is None
    prob.r.zero()
    for _ in range(2):
        g()
    g.sync()
    assert_allclose(prob.r.data_ro, ro, rtol=0, atol=1e-12 * np.abs(ro).max())
    assert_allclose(prob.jacobian()[0].toscipy().data, Ao.data, rtol=0, atol=1e-12 * np.abs(Ao.data).max())
    DeviceBuffer(64)                             # (allocation works again after the capture)


def test_c2_full_size_properties():
    """BASELINE.json configs[1] at full size (59.6 M tets, 10.08 M DoFs, nnz 150 M): size-independent properties of the
    assembled tensors instead of the oracle -- constants in the null space, symmetry (x'Ay = y'Ax), residual of a
    constant state with f = 0 vanishes, F(u) = A u - M f linearity, second assembly idempotent."""
    n = 215
    m = fmesh.UnitCubeMesh(n, degrees=(1,), perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=False)
    V = prob.V
    assert V.node_set.size == 10077696 and m.cell_set.size == 59630250
    mat = prob.assemble_jacobian()
    assert mat.sparsity.nz == 150048286                      # SURVEY.md 8 sizes
    nn = V.node_set.size
    ones = V.dat(1, np.ones(nn))
    y = V.dat()
    mat.mult(ones, y)
    rp, ci, v = mat.csr()
    amax = np.abs(v).max()
    assert np.abs(y.data_ro).max() <= 1e-10 * amax
    rng = np.random.default_rng(0)
    a, b = rng.standard_normal(nn), rng.standard_normal(nn)
    da, db, ta, tb = V.dat(1, a), V.dat(1, b), V.dat(), V.dat()
    mat.mult(da, ta)
    mat.mult(db, tb)
    lhs, rhs = float(b @ ta.data_ro), float(a @ tb.data_ro)
    assert abs(lhs - rhs) <= 1e-10 * max(abs(lhs), abs(rhs))
    # residual: F(u) with f = 0 equals A u
    prob.f.assign(0.0)
    r = np.array(prob.assemble_residual().data_ro)
    mat.mult(prob.u, y)
    assert np.abs(r - y.data_ro).max() <= 1e-10 * max(1.0, np.abs(r).max())
    prob.u.assign(3.25)
    r0 = prob.assemble_residual().data_ro
    assert np.abs(r0).max() <= 1e-10 * amax
    v2 = prob.assemble_jacobian().csr()[2]
    assert np.abs(v2 - v).max() <= 1e-12 * amax


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic"])
def test_plan_ordered_copies_follow_their_dats(numbering, monkeypatch):
    """Parloop._plan_copy: a READ Dat that did not change since the previous call is kept in plan order and streamed by the
    staging phase; the copy must be dropped the moment the Dat changes -- a new state (Newton update), new forcing, a moved
    mesh -- and rebuilt when the new state repeats.  Residual and Jacobian against the oracle after every change."""
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(7, degrees=(1,), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    rng = np.random.default_rng(3)

    def check(tag):
        ro, Ao = _oracle_problem(prob, True)
        r = np.array(prob.assemble_residual().data_ro)
        A = prob.assemble_jacobian().toscipy()
        assert np.abs(r - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max()), tag
        assert np.abs(A.data - Ao.data).max() <= 1e-12 * np.abs(Ao.data).max(), tag

    def copies(loop):
        geos = list(loop._prepared["parts"].values())
        return sum(1 for g in geos for e in g.get("plan_copies", {}).values() if e["buf"] is not None)

    for rep in range(3):
        check(f"static {rep}")                                     # from the second repetition on every READ Dat is streamed
    assert copies(prob.res_loop) >= 3 and copies(prob.jacobian()[1]) >= 1
    prob.u.data[...] = rng.standard_normal(prob.u.data.shape)       # Newton update: u's copy is stale
    check("new u")
    check("new u, again")
    prob.f.data[...] = rng.standard_normal(prob.f.data.shape)
    check("new f")
    xyz = m.coordinates.data
    xyz[...] = xyz * 1.25 + 0.1                                     # a moved mesh: geometry copies of BOTH loops are stale
    check("moved mesh")
    check("moved mesh, again")


def test_two_parloops_on_two_streams_and_the_cached_diagonal_places():
    """``device.Stream``: the Jacobian of a Newton step forked onto a side stream while the residual runs on the launching one (two
    loops that write different tensors and only read what they share), joined by an event -- no host synchronisation in between --
    gives the tensors of the serial order; and ``Mat.set_diagonal_rows`` (places of the BC diagonal entries searched once, stored
    through at every assembly) equals ``set_local_diagonal_entries`` of the same rows."""
    from firedrake_amd import _lib
    from firedrake_amd.device import Stream
    m = fmesh.UnitCubeMesh((14, 14, 14), degrees=(1,), perturb=0.1, numbering="lexicographic")
    prob = forms.PoissonProblem(m, 1, bcs=True)
    r0 = np.array(prob.assemble_residual().data_ro)
    v0 = prob.assemble_jacobian().csr()[2].copy()
    side = Stream()
    for _ in range(3):
        prob.u.dat_version += 1
        with side.fork():
            prob.assemble_jacobian()
        prob.assemble_residual()
        side.join()
    _lib.call("fd_device_sync")
    assert_allclose(np.array(prob.r.data_ro), r0, rtol=0, atol=1e-13 * np.abs(r0).max())
    mat = prob.jacobian()[0]
    assert_allclose(mat.csr()[2], v0, rtol=0, atol=1e-13 * np.abs(v0).max())
    # the same fork inside a captured step: two branches of one hipGraph (the fork hands the CAPTURING stream back when it ends)
    from firedrake_amd.graph import CapturedStep

    def step():
        with side.fork():
            prob.assemble_jacobian()
        prob.assemble_residual()
        side.join()

    g = CapturedStep(step)
    prob.r.zero()
    mat.zero()
    mat._values_dev()                           # flush the pending zero: the tensors really are cleared now
    for _ in range(3):
        g()
    g.sync()
    assert_allclose(np.array(prob.r.data_ro), r0, rtol=0, atol=1e-13 * np.abs(r0).max())
    assert_allclose(mat.csr()[2], v0, rtol=0, atol=1e-13 * np.abs(v0).max())
    prob.assemble_residual()                    # eager launches after the capture go to the null stream again
    _lib.call("fd_device_sync")
    assert_allclose(np.array(prob.r.data_ro), r0, rtol=0, atol=1e-13 * np.abs(r0).max())
    # the diagonal fix-up through the API that takes host rows: same places, other value
    rp, ci, vb = mat.csr()
    vb = vb.copy()
    mat.set_local_diagonal_entries(prob.bc_nodes, 3.5)
    v1 = mat.csr()[2]
    diag = np.array([rp[b] + np.searchsorted(ci[rp[b]:rp[b + 1]], b) for b in prob.bc_nodes])
    assert np.all(v1[diag] == 3.5)
    rest = np.ones(len(v1), dtype=bool)
    rest[diag] = False
    assert np.array_equal(v1[rest], vb[rest])
    mat.set_local_diagonal_entries(prob.bc_nodes[::2], 1.0)          # another list: its own places
    v2 = mat.csr()[2]
    assert np.all(v2[diag[::2]] == 1.0) and np.all(v2[diag[1::2]] == 3.5)


@pytest.mark.parametrize("degree,n", [(1, 10), (2, 5)])
def test_output_tensor_of_a_cached_parloop_is_swapped(degree, n):
    """Firedrake's assemblers build their parloops once and swap the output tensor per call -- ``parloop.arguments[0].data = data``
    (firedrake/assemble.py:1073-1077).  Nothing a Parloop caches may hang on the tensor it was built with: the swapped-in Dat / Mat
    receives the result, the old one is left alone (whole-entity and row-sliced matrix loops, staged vector loops)."""
    m = fmesh.UnitCubeMesh(n, degrees=(degree,), perturb=0.1)
    prob = forms.PoissonProblem(m, degree, bcs=True)
    loop = prob.res_loop
    r1 = prob.r
    for _ in range(2):                                   # (twice: plans, plan copies and argument getters exist)
        r1.zero()
        loop()
    ref = np.array(r1.data_ro)
    assert np.abs(ref).max() > 0
    r2 = prob.V.dat(1, None, "r2")
    loop.arguments[0].data = r2
    r1.data[...] = -7.0
    r2.zero()
    loop()
    assert_allclose(r2.data_ro, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    assert np.all(r1.data_ro == -7.0)
    loop.arguments[0].data = r1
    r1.zero()
    loop()
    assert_allclose(r1.data_ro, ref, rtol=0, atol=1e-13 * np.abs(ref).max())
    mat, jloop = prob.jacobian()
    for _ in range(2):
        mat.zero()
        jloop()
    v1 = mat.csr()[2].copy()
    assert np.abs(v1).max() > 0
    mat2 = op2.Mat(mat.sparsity)
    jloop.arguments[0].data = mat2
    mat2.zero()
    jloop()
    assert_allclose(mat2.csr()[2], v1, rtol=0, atol=1e-13 * np.abs(v1).max())
    assert np.array_equal(mat.csr()[2], v1)              # the first matrix is left alone
    jloop()                                              # ADD_VALUES on top of the swapped-in matrix
    assert_allclose(mat2.csr()[2], 2.0 * v1, rtol=0, atol=1e-13 * np.abs(v1).max())
    jloop.arguments[0].data = mat
    mat.zero()
    jloop()
    assert_allclose(mat.csr()[2], v1, rtol=0, atol=1e-13 * np.abs(v1).max())
