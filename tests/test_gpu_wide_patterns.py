"""GPU: sparsity patterns of more than 2^31 entries -- row starts are ``fd_nnz_t`` (64-bit; PETSc's IntType for nnz,
pyop2/datatypes.py:6-10, the 64-bit-index CI variant .github/workflows/core.yml:276), everything inside a row or a block stays 32-bit.
The Q4 Helmholtz matrix of BASELINE.json configs[2] at n = 64 holds 3.6e9 entries (44 GB of indices and values): assembled on the
fp64 matrix cores and checked where it crosses 2^31 -- sampled rows against the oracle's dense element kernel, global properties
through the device SpMV.  Skipped on devices without the memory."""
import ctypes

import numpy as np
import pytest

from firedrake_amd import _lib, forms, mesh as fmesh, op2

pytestmark = pytest.mark.gpu


def _free_bytes():
    name = ctypes.create_string_buffer(256)
    cus, mem, wave = ctypes.c_int(), ctypes.c_size_t(), ctypes.c_int()
    _lib.call("fd_device_info", -1, name, 256, ctypes.byref(cus), ctypes.byref(mem), ctypes.byref(wave))
    return int(mem.value)


def test_q4_matrix_beyond_2_31_entries():
    from test_forms_identities import _element_tensor
    if _free_bytes() < 200e9:
        pytest.skip("needs ~160 GB of device memory while the pattern is built")
    n = 64
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=True)
    mat = prob.assemble_jacobian()
    assert prob.jac_loop._prepare()["cw"].src.mode == "tp_matrix"
    sp = mat.sparsity
    assert sp.nz > 2 ** 31 and m.node_set.size == 257 ** 3
    rp = sp.rowptr
    assert rp.dtype == np.int64 and int(rp[-1]) == sp.nz and (np.diff(rp) > 0).all()
    cm, xm = m.cell_node_map.values_with_halo, m.coord_map.values_with_halo
    coords = np.array(m.coordinates.data_ro_with_halos)
    bc = np.zeros(m.node_set.total_size, dtype=bool)
    bc[prob.bc_nodes] = True
    interior = [(a * 5 + b) * 5 + c for a in (1, 2, 3) for b in (1, 2, 3) for c in (1, 2, 3)]
    rng = np.random.default_rng(5)
    # cells all over the mesh -- the rows of the later columns start beyond 2^31
    cols_ = np.concatenate([rng.integers(0, m.base_set.size, 40), [0, m.base_set.size - 1, m.base_set.size - 2, m.base_set.size // 2 + 700]])
    lays = np.concatenate([rng.integers(0, n, 40), [0, n - 1, 5, n - 2]])
    worst, beyond = 0.0, 0
    for col, lay in zip(cols_, lays):
        nodes = cm[col] + 4 * lay
        Ae = _element_tensor(prob.kjac, 125, coords[xm[col] + lay]).reshape(125, 125)
        scale = np.abs(Ae).max()
        for i in interior[::3]:
            r = nodes[i]
            n0, n1 = int(rp[r]), int(rp[r + 1])
            beyond += n0 > 2 ** 31
            cidx, vals = np.empty(n1 - n0, dtype=np.int32), np.empty(n1 - n0)
            _lib.call("fd_memcpy_d2h", cidx.ctypes.data, sp._colidx.ptr + 4 * n0, cidx.nbytes, None)
            _lib.call("fd_memcpy_d2h", vals.ctypes.data, mat._values_dev().ptr + 8 * n0, vals.nbytes, None)
            assert np.array_equal(np.sort(nodes), cidx)
            worst = max(worst, np.abs(vals[np.searchsorted(cidx, nodes)] - np.where(bc[nodes], 0.0, Ae[i])).max() / scale)
    assert beyond > 50 and worst <= 1e-11, (beyond, worst)
    # global properties through the device SpMV (64-bit row starts): without BCs 1'A1 = |Omega| and x'Ay = y'Ax
    del prob, mat, sp
    prob = forms.HelmholtzQ4Problem(m)
    mat = prob.assemble_jacobian()
    nn = m.node_set.size
    a, b = rng.standard_normal(nn), rng.standard_normal(nn)
    d1, da, db = op2.Dat(m.node_set, np.ones(nn)), op2.Dat(m.node_set, a), op2.Dat(m.node_set, b)
    t1, ta, tb = op2.Dat(m.node_set), op2.Dat(m.node_set), op2.Dat(m.node_set)
    mat.mult(d1, t1); mat.mult(da, ta); mat.mult(db, tb)
    assert abs(np.ones(nn) @ np.array(t1.data_ro) - 1.0) < 1e-9
    assert abs(b @ np.array(ta.data_ro) - a @ np.array(tb.data_ro)) <= 1e-9 * (np.abs(a) @ np.abs(np.array(tb.data_ro)))
    # the operator action of the same form agrees with the assembled matrix
    y = np.array(prob.assemble_action().data_ro)
    du = op2.Dat(m.node_set, np.array(prob.u.data_ro))
    mat.mult(du, t1)
    assert np.abs(y - np.array(t1.data_ro)).max() <= 1e-10 * np.abs(y).max()


def test_cg2_cube_of_configs4_on_one_gpu():
    """BASELINE.json configs[4] as written -- Poisson CG2 on UnitCubeMesh(215, 215, 215): 80.06 M DoFs, 2.29e9 nonzeros -- on ONE
    device: the row-sliced owner-computes-rows Jacobian with its run-coded flush (64-bit run displacements) over a pattern that crosses
    2^31.  The oracle cannot hold that matrix (its CSR is int32, and a CPU pass over it takes an hour), so: sampled rows -- the last
    ones start beyond 2^31 -- against the sum of the oracle's element tensors of the cells around the row's node, the pattern of every
    sampled row against the union of those cells' nodes, and global properties of the stiffness matrix through the device SpMV."""
    from test_forms_identities import _element_tensor
    if _free_bytes() < 200e9:
        pytest.skip("needs ~120 GB of device memory")
    m = fmesh.UnitCubeMesh(215, degrees=(2,), perturb=0.1, numbering="lexicographic")
    prob = forms.PoissonProblem(m, 2, bcs=True)
    mat = prob.assemble_jacobian()
    assert prob.jacobian()[1]._prepare()["cw"].src.mode.startswith("ocrs")
    sp = mat.sparsity
    nn = prob.V.node_set.size
    assert nn == 431 ** 3 and sp.nz == 2292210461 and sp.nz > 2 ** 31
    rp = sp.rowptr
    assert rp.dtype == np.int64 and int(rp[-1]) == sp.nz
    cm = np.asarray(prob.V.cell_node_map.values_with_halo)
    xm = np.asarray(m.coord_space.cell_node_map.values_with_halo)
    coords = np.array(m.coordinates.data_ro_with_halos)
    bc = np.zeros(nn, dtype=bool)
    bc[prob.bc_nodes] = True
    rng = np.random.default_rng(3)
    beyond = np.nonzero((rp[:-1] > 2 ** 31) & ~bc)[0]                       # (the last 6 % of the rows)
    assert len(beyond) > 1000000
    cand = np.concatenate([rng.choice(beyond, 10, replace=False), [beyond[0], beyond[-1]], rng.integers(0, nn, 30)])
    rows = [int(r) for r in cand if not bc[r]][:22]
    assert sum(int(rp[r]) > 2 ** 31 for r in rows) >= 12
    worst = 0.0
    for r in rows:
        cells = np.nonzero((cm == r).any(axis=1))[0]
        expect, scale = {}, 0.0
        for c in cells:
            Ae = np.asarray(_element_tensor(prob.kjac, 10, coords[xm[c]])).reshape(10, 10)
            scale = max(scale, np.abs(Ae).max())
            i = int(np.nonzero(cm[c] == r)[0][0])
            for j, g in enumerate(cm[c]):
                expect[int(g)] = expect.get(int(g), 0.0) + (0.0 if bc[g] else Ae[i, j])
        n0, n1 = int(rp[r]), int(rp[r + 1])
        cidx, vals = np.empty(n1 - n0, dtype=np.int32), np.empty(n1 - n0)
        _lib.call("fd_memcpy_d2h", cidx.ctypes.data, sp._colidx.ptr + 4 * n0, cidx.nbytes, None)
        _lib.call("fd_memcpy_d2h", vals.ctypes.data, mat._values_dev().ptr + 8 * n0, vals.nbytes, None)
        assert np.array_equal(cidx, np.array(sorted(expect), dtype=np.int32))
        worst = max(worst, np.abs(vals - np.array([expect[g] for g in sorted(expect)])).max() / scale)
    assert worst <= 1e-12, worst
    # a second assembly into the same tensor reproduces the sampled rows bit for bit (deterministic block-owned sums)
    r = rows[-1]
    n0, n1 = int(rp[r]), int(rp[r + 1])
    v1 = np.empty(n1 - n0)
    _lib.call("fd_memcpy_d2h", v1.ctypes.data, mat._values_dev().ptr + 8 * n0, v1.nbytes, None)
    prob.assemble_jacobian()
    v2 = np.empty(n1 - n0)
    _lib.call("fd_memcpy_d2h", v2.ctypes.data, mat._values_dev().ptr + 8 * n0, v2.nbytes, None)
    assert np.array_equal(v1, v2)
    # without BCs the stiffness matrix annihilates constants and is symmetric
    del prob, mat, sp
    prob = forms.PoissonProblem(m, 2, bcs=False)
    mat = prob.assemble_jacobian()
    a, b = rng.standard_normal(nn), rng.standard_normal(nn)
    ns = prob.V.node_set
    d1, da, db = op2.Dat(ns, np.ones(nn)), op2.Dat(ns, a), op2.Dat(ns, b)
    t1, ta, tb = op2.Dat(ns), op2.Dat(ns), op2.Dat(ns)
    mat.mult(d1, t1); mat.mult(da, ta); mat.mult(db, tb)
    ya, yb = np.array(ta.data_ro), np.array(tb.data_ro)
    assert np.abs(np.array(t1.data_ro)).max() <= 1e-10 * np.abs(ya).max()
    assert abs(b @ ya - a @ yb) <= 1e-10 * (np.abs(b) @ np.abs(ya))


def test_whole_entity_loops_on_wide_patterns_take_the_row_sliced_shape(monkeypatch):
    """A P1-like (whole-entity) matrix loop over a derived row order flushes through 32-bit places; a pattern of 2^31 entries or more
    used to end on the slow path.  It now moves to the row-sliced shape -- 64-bit row starts, run-coded flush --
    whatever the size of its element matrix.  Exercised by lowering the limit: the P1 Jacobian takes "ocrs" and matches the oracle."""
    from firedrake_amd import forms, mesh as fmesh, parloop
    from firedrake_amd.configuration import configuration
    from helpers import oracle_run
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    monkeypatch.setattr(parloop, "WIDE_PLACES", 1000)
    m = fmesh.UnitCubeMesh(10, degrees=(1,), perturb=0.1, numbering="lexicographic")
    prob = forms.PoissonProblem(m, 1, bcs=True)
    mat, pl = prob.jacobian()
    for _ in range(2):
        mat.zero()
        pl.compute()
    assert pl._prepared["cw"].src.mode.startswith("ocrs")
    geo = [g for key, g in pl._prepared["parts"].items() if key[0] == "ocr"][0]
    assert geo["cw"].src.mode.startswith("ocrs") and geo["groups"] is not None and len(geo["groups"]) == 2       # two pairs of the four rows
    mpa = pl.arguments[0]
    args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
    ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
    assert np.abs(mat.csr()[2] - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
