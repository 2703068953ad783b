"""GPU parity at larger sizes: staged (LDS plan) wrapper vs direct wrapper vs oracle on
seeded unstructured-looking inputs, plus the block-localisation plan against a numpy restatement."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import op2
from firedrake_amd.configuration import configuration
import golden_kernels as gk
from helpers import lane_slot_to_entity as _lane_slot_to_entity, oracle_run, plan_ref as _plan_ref, structured_tri_mesh

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arity,epb,n", [(3, 256, 5000), (4, 1024, 20000), (10, 512, 7001), (1, 64, 130), (8, 2048, 4096)])
def test_plan_matches_numpy(arity, epb, n):
    rng = np.random.default_rng(arity)
    ntgt = max(n // 2, 4)
    # locality like a mesh numbering: targets near the entity index, plus a few far ones
    base = (np.arange(n)[:, None] * ntgt // n + rng.integers(0, 40, size=(n, arity))) % ntgt
    it, to = op2.Set(n), op2.Set(ntgt)
    m = op2.Map(it, to, arity, base.astype(np.int32))
    for (s, e) in [(0, n), (17, n - 5)]:
        p = m.plan(s, e, epb)
        blk, lst, lm = p.download()
        rblk, rlst, rlm = _plan_ref(m.values_with_halo, s, e, epb)
        assert np.array_equal(blk, rblk) and np.array_equal(lst, rlst) and np.array_equal(lm, rlm)
        assert p.max_nd == np.diff(rblk).max() and p.nblocks == len(rblk) - 1


def test_plan_with_explicit_blocks():
    rng = np.random.default_rng(5)
    n, arity = 3000, 4
    it, to = op2.Set(n), op2.Set(900)
    m = op2.Map(it, to, arity, ((np.arange(n)[:, None] * 900 // n + rng.integers(0, 30, size=(n, arity))) % 900).astype(np.int32))
    cuts = np.sort(rng.choice(np.arange(1, n), size=17, replace=False))
    blocks = np.concatenate([[0], cuts, [n]]).astype(np.int32)
    p = m.plan(0, n, 0, blocks)
    blk, lst, lm = p.download()
    off = 0
    for b in range(len(blocks) - 1):
        u, inv = np.unique(m.values[blocks[b]:blocks[b + 1]].reshape(-1), return_inverse=True)
        assert np.array_equal(lst[blk[b]:blk[b + 1]], u)
        assert np.array_equal(lm[blocks[b]:blocks[b + 1]].reshape(-1), inv)
    assert p.nblocks == len(blocks) - 1


@pytest.mark.parametrize("T", [64, 256, 512])
def test_plan_lane_order_is_the_documented_permutation(T):
    rng = np.random.default_rng(T)
    n, arity = 5003, 4
    it, to = op2.Set(n), op2.Set(1500)
    m = op2.Map(it, to, arity, ((np.arange(n)[:, None] * 1500 // n + rng.integers(0, 30, size=(n, arity))) % 1500).astype(np.int32))
    cuts = np.sort(rng.choice(np.arange(1, n), size=9, replace=False))
    blocks = np.concatenate([[0], cuts, [cuts[-1]], [n]]).astype(np.int32)       # includes an empty block
    p0 = m.plan(0, n, 0, blocks)
    p1 = m.plan(0, n, 0, blocks, lane_threads=T)
    assert p1 is not p0
    blk0, lst0, lm0 = p0.download()
    blk1, lst1, lm1 = p1.download()
    assert np.array_equal(blk0, blk1) and np.array_equal(lst0, lst1)          # node lists unchanged
    for b in range(len(blocks) - 1):
        e0, e1 = blocks[b], blocks[b + 1]
        if e1 > e0:
            ent = _lane_slot_to_entity(e1 - e0, T)
            assert np.array_equal(lm1[e0:e1], lm0[e0:e1][ent])


@pytest.mark.parametrize("order", ["stencil", "natural"])
def test_ocr_instance_orders_are_permutations(order, monkeypatch):
    """Every instance order of an owner-computes-rows plan lists, per row block, exactly the entities that touch
    the block's rows (once each); only their order differs."""
    from firedrake_amd import mesh as fmesh
    from firedrake_amd.op2types import OcrPlan
    monkeypatch.setitem(configuration, "ocr_order", order)
    m = fmesh.UnitCubeMesh(6, degrees=(1,), tile=(4, 2, 2), perturb=0.1)
    V = m.space(1)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    sp._build()
    nrows = V.node_set.size
    rb = np.unique(np.concatenate([np.arange(0, nrows, 37), [nrows]])).astype(np.int32)
    op = OcrPlan(sp, cm, cm, {0: cm}, 0, m.cell_set.size, rb, lane_threads=128)
    ent = np.empty(op.ninst, dtype=np.int32)
    from firedrake_amd import _lib
    _lib.call("fd_memcpy_d2h", ent.ctypes.data, op.inst_ent, ent.nbytes, None)
    vals = cm.values_with_halo
    for b in range(len(rb) - 1):
        got = np.sort(ent[op.inst_off_host[b]:op.inst_off_host[b + 1]])
        touch = ((vals[:m.cell_set.size] >= rb[b]) & (vals[:m.cell_set.size] < rb[b + 1])).any(axis=1)
        assert np.array_equal(got, np.nonzero(touch)[0])


def test_ocr_plan_matches_numpy_restatement(monkeypatch):
    """The numpy restatement the host-sim of the owner-computes-rows wrapper runs on (helpers.ocr_plan_ref) equals the
    device's plan in natural order: instance offsets, instance lists and the per-instance row-offset table."""
    from firedrake_amd import _lib, mesh as fmesh
    from firedrake_amd.op2types import OcrPlan
    from helpers import ocr_plan_ref
    monkeypatch.setitem(configuration, "ocr_order", "natural")
    m = fmesh.UnitCubeMesh(5, degrees=(1,), tile=(4, 2, 2), perturb=0.1)
    V = m.space(1)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    sp._build()
    nrows = V.node_set.size
    rb = np.unique(np.concatenate([np.arange(0, nrows, 29), [nrows]])).astype(np.int32)
    op = OcrPlan(sp, cm, cm, {0: cm}, 0, m.cell_set.size, rb, lane_threads=0)
    inst_off, inst_ent, kidx = ocr_plan_ref(cm.values_with_halo, cm.values_with_halo, m.cell_set.size, rb, sp.rowptr, sp.colidx)
    assert np.array_equal(op.inst_off_host, inst_off)
    ent = np.empty(op.ninst, dtype=np.int32)
    _lib.call("fd_memcpy_d2h", ent.ctypes.data, op.inst_ent, ent.nbytes, None)
    assert np.array_equal(ent, inst_ent)
    assert op.kbytes == 1
    assert np.array_equal(op.kidx.download(np.uint8, kidx.shape), kidx)


@pytest.mark.parametrize("nx,ny", [(7, 5), (64, 64), (200, 150)])
@pytest.mark.parametrize("shuffle", [False, True])
def test_p1_mass_and_rhs_all_paths(nx, ny, shuffle, monkeypatch):
    coords, cells = structured_tri_mesh(nx, ny, perturb=0.2)
    rng = np.random.default_rng(3)
    if shuffle:
        cells = cells[rng.permutation(len(cells))]
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x = op2.Dat(nodes ** 2, coords)
    f = op2.Dat(nodes, rng.standard_normal(len(coords)))
    krhs, kmass = op2.Kernel(gk.RHS_Q6, "rhs_q6"), op2.Kernel(gk.MASS_Q6, "mass_q6")
    ob = oracle_run(krhs, ele, op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))[0]
    sp = op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)])
    ocsr = oracle_run(kmass, ele, op2.Mat(sp)(op2.INC, (m, m)), x(op2.READ, m))[0]
    # sparsity identical to the oracle's restatement of sparsity.pyx
    assert np.array_equal(sp.rowptr, ocsr.rowptr) and np.array_equal(sp.colidx, ocsr.colidx)
    for mode, scatter, ocr in (("auto", "table", 1), ("auto", "table", 0), ("auto", "search", 0), ("direct", "table", 0), ("direct", "search", 0)):
        if True:
            monkeypatch.setitem(configuration, "mode", mode)
            monkeypatch.setitem(configuration, "mat_scatter", scatter)
            monkeypatch.setitem(configuration, "mat_ocr", ocr)
            b = op2.Dat(nodes)
            op2.par_loop(krhs, ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
            # tolerance: SURVEY.md Appendix D -- atomics reorder fp adds
            assert_allclose(b.data, ob, rtol=0, atol=1e-12 * max(1.0, np.abs(ob).max()))
            mat = op2.Mat(sp)
            op2.par_loop(kmass, ele, mat(op2.INC, (m, m)), x(op2.READ, m))
            _, _, v = mat.csr()
            assert_allclose(v, ocsr.values, rtol=0, atol=1e-12 * np.abs(ocsr.values).max())
    # identities the reference's regression tests use: sum(M) = |Omega| (test_assemble.py:60-73)
    assert_allclose(v.sum(), 1.0, rtol=1e-6)
    # A*x == action: M f vs rhs(f)  (test_matrix_free.py:97-123)
    y = op2.Dat(nodes)
    mat.mult(f, y)
    assert_allclose(y.data, ob, rtol=0, atol=1e-9 * np.abs(ob).max())


def test_repeated_launch_accumulates_and_versions():
    coords, cells = structured_tri_mesh(20, 20)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x, f, b = op2.Dat(nodes ** 2, coords), op2.Dat(nodes, np.ones(len(coords))), op2.Dat(nodes)
    pl = op2.LegacyParloop(op2.Kernel(gk.RHS_AFFINE, "rhs_affine"), ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    pl()
    once = b.data_ro.copy()
    pl()
    assert_allclose(b.data_ro, 2 * once, rtol=1e-13)
    f.data[:] = 2.0                     # host write must reach the device (dat_version staleness)
    b.zero()
    pl()
    assert_allclose(b.data_ro, 2 * once, rtol=1e-13)
    assert_allclose(b.data_ro.sum(), 2.0, rtol=1e-12)


def test_empty_and_tiny_iteration_sets():
    nodes, ele = op2.Set(4), op2.Set(0)
    m = op2.Map(ele, nodes, 3, np.zeros((0, 3), np.int32))
    b = op2.Dat(nodes)
    x = op2.Dat(nodes ** 2)
    op2.par_loop(op2.Kernel(gk.MASS_AFFINE.replace("mass_affine", "mass_e").replace("double A[9]", "double *A"), "mass_e"),
                 ele, b(op2.INC, m), x(op2.READ, m))
    assert np.all(b.data == 0)


@pytest.mark.parametrize("dtype,ctype", [(np.float32, "float"), (np.int32, "int"), (np.uint32, "unsigned int"), (np.float64, "double")])
@pytest.mark.parametrize("mode", ["auto", "direct"])
def test_staged_inc_other_dtypes_and_high_arity(dtype, ctype, mode, monkeypatch):
    """INC through a map for every stageable dtype, arity 27 (a Q2-hexahedron-sized map), vector cdim 2."""
    monkeypatch.setitem(configuration, "mode", mode)
    rng = np.random.default_rng(7)
    n, ntgt, ar = 3000, 800, 27
    vals = ((np.arange(n)[:, None] * ntgt // n + rng.integers(0, 60, size=(n, ar))) % ntgt).astype(np.int32)
    it, to = op2.Set(n), op2.Set(ntgt)
    m = op2.Map(it, to, ar, vals)
    src = op2.Dat(to ** 2, rng.integers(1, 5, size=(ntgt, 2)).astype(dtype), dtype)
    out = op2.Dat(to ** 2, dtype=dtype)
    k = op2.Kernel(f"static void acc27({ctype} *o, const {ctype} *s) {{ for (int i = 0; i < 27; ++i) {{ o[2*i] += s[2*i+1]; o[2*i+1] += ({ctype})(i % 3); }} }}",
                   f"acc27_{np.dtype(dtype).name}".replace("acc27_", "acc27"))
    k = op2.Kernel(k.code.replace("acc27(", f"acc27_{np.dtype(dtype).name}("), f"acc27_{np.dtype(dtype).name}")
    op2.par_loop(k, it, out(op2.INC, m), src(op2.READ, m))
    ref = oracle_run(k, it, op2.Dat(to ** 2, dtype=dtype)(op2.INC, m), src(op2.READ, m))[0]
    if np.issubdtype(dtype, np.integer):
        assert np.array_equal(out.data_ro, ref)                  # integer work is bit-exact
    else:
        assert_allclose(out.data_ro, ref, rtol=1e-5 if dtype == np.float32 else 1e-13)


def test_plan_with_empty_and_single_entity_blocks():
    rng = np.random.default_rng(11)
    n = 700
    it, to = op2.Set(n), op2.Set(300)
    m = op2.Map(it, to, 3, rng.integers(0, 300, size=(n, 3)).astype(np.int32))
    blocks = np.array([0, 0, 1, 1, 250, 250, 699, 700, 700], dtype=np.int32)     # empty, single-entity and large blocks
    p = m.plan(0, n, 0, blocks)
    blk, lst, lm = p.download()
    for b in range(len(blocks) - 1):
        rows = m.values[blocks[b]:blocks[b + 1]]
        u, inv = (np.unique(rows.reshape(-1), return_inverse=True) if len(rows) else (np.zeros(0, np.int32), np.zeros(0, np.int64)))
        assert np.array_equal(lst[blk[b]:blk[b + 1]], u)
        assert np.array_equal(lm[blocks[b]:blocks[b + 1]].reshape(-1), inv)


@pytest.mark.parametrize("numbering", ["lexicographic", "random"])
def test_locality_order_matches_numpy_and_drives_the_staged_wrapper(numbering, monkeypatch):
    """fd_entity_centroids + fd_kd_order against the numpy restatement (identical order and leaf boundaries), the first-touch
    row order derived from it with its ranks, and the un-hinted residual / Jacobian taking the "stagedo" / "ocrp" wrappers
    end to end."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import DeviceBuffer
    from helpers import locality_order_ref
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(9, degrees=(1,), perturb=0.1, numbering=numbering)
    V = m.space(1)
    cm = V.cell_node_map
    n = m.cell_set.size
    pos = np.array(m.coordinates.data_ro)
    target = 96
    from firedrake_amd.parloop import kd_order
    from helpers import kd_order_ref
    nnod = V.node_set.total_size
    nbuf, nstarts = kd_order(m.coordinates._dev_ptr(False), 3, nnod, 0, 17)
    nord_ref, nstarts_ref = kd_order_ref(pos, 17)
    assert np.array_equal(nstarts, nstarts_ref) and np.array_equal(nbuf.download(np.int32, (nnod,)), nord_ref)   # same keys, stable sorts
    assert len(nstarts) - 1 == -(-nnod // 17) and np.diff(nstarts).max() - np.diff(nstarts).min() <= 1                 # equal leaves
    order_ref, starts_ref = locality_order_ref(cm.values_with_halo, 0, n, pos, target=target)
    monkeypatch.setitem(configuration, "locality_tile_entities", target)
    pl = forms.PoissonProblem(m, 1, bcs=False).res_loop
    pl._prepare()
    lo = pl._locality_order(0, n)
    order, starts = lo.buf.download(np.int32, (n,)), lo.blocks
    assert sorted(order.tolist()) == list(range(n))
    assert np.array_equal(starts, starts_ref) and np.array_equal(order, order_ref)
    buf = lo.buf
    # a block touches its leaf of nodes plus one layer: far fewer nodes than a run of the caller's order of the same length
    touched = lambda o: sum(len(np.unique(cm.values_with_halo[o[a:b]])) for a, b in zip(starts[:-1], starts[1:]))      # noqa: E731
    if numbering == "random":
        assert touched(order) < 0.5 * touched(np.arange(n))
    # first-touch rows under that order
    nn = V.node_set.size
    pinv, plist, rank = DeviceBuffer(nn * 4), DeviceBuffer(nn * 4), DeviceBuffer(nn * 4)
    _lib.call("fd_first_touch_order", cm._dev_values(), 4, buf.ptr, n, nn, pinv.ptr, plist.ptr, rank.ptr, None)
    pl, pi, rk = plist.download(np.int32, (nn,)), pinv.download(np.int32, (nn,)), rank.download(np.int32, (nn,))
    assert np.array_equal(pi[pl], np.arange(nn)) and sorted(pl.tolist()) == list(range(nn))
    first = np.full(nn, np.iinfo(np.int64).max)
    flat = cm.values_with_halo[order].reshape(-1)
    first[flat[::-1]] = np.arange(len(flat) - 1, -1, -1) // 4
    assert (np.diff(first[pl]) >= 0).all()                          # rows sorted by the rank of the first cell touching them
    assert np.array_equal(rk, first[pl])                            # and the rank itself is returned per row position
    prob = forms.PoissonProblem(m, 1, bcs=True)
    r = prob.assemble_residual()
    mode = prob.res_loop._staged_geometry(0, n)["cw"].src.mode
    if numbering == "random":            # (a numbering with locality keeps the caller's order unless the derived one wins clearly)
        assert mode.startswith("stagedo")
    from test_gpu_forms import _oracle_problem
    ro, Ao = _oracle_problem(prob, True)
    assert np.abs(r.data_ro - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max())
    A = prob.assemble_jacobian().toscipy()
    geo = prob.jacobian()[1]._ocr_geometry()
    assert geo["cw"].src.mode.startswith("ocrp") and geo["row_order"] is not None
    assert np.abs(A.data - Ao.data).max() <= 1e-12 * np.abs(Ao.data).max()
    mat, loop = prob.jacobian()
    A2 = prob.assemble_jacobian().toscipy()                         # plans reused, pending zero fused
    assert np.abs(A2.data - Ao.data).max() <= 1e-12 * np.abs(Ao.data).max()


def test_ordered_ocr_plan_matches_numpy_restatement(monkeypatch):
    """fd_ocrplan_create_ordered (row blocks = ranges of row positions) against helpers.ocr_plan_ref(pinv=...) in natural
    instance order, with the row order built by fd_first_touch_order from a numpy-made entity order."""
    from firedrake_amd import _lib, mesh as fmesh
    from firedrake_amd.device import DeviceBuffer
    from firedrake_amd.op2types import OcrPlan, RowOrder
    from helpers import first_touch_ref, locality_order_ref, ocr_plan_ref
    monkeypatch.setitem(configuration, "ocr_order", "natural")
    m = fmesh.UnitCubeMesh(6, degrees=(1,), perturb=0.1, numbering="random")
    V = m.space(1)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    sp._build()
    n, nrows = m.cell_set.size, V.node_set.size
    order = locality_order_ref(cm.values_with_halo, 0, n, np.array(m.coordinates.data_ro), target=128)[0]
    obuf = DeviceBuffer.from_numpy(order)
    ro = RowOrder(cm, obuf, n, nrows, sp.rowptr)
    plist, pinv = first_touch_ref(cm.values_with_halo, order, nrows)
    assert np.array_equal(ro.plist.download(np.int32, (nrows,)), plist) and np.array_equal(ro.pinv.download(np.int32, (nrows,)), pinv)
    rb = np.unique(np.concatenate([np.arange(0, nrows, 31), [nrows]])).astype(np.int32)
    op = OcrPlan(sp, cm, cm, {0: cm}, 0, n, rb, lane_threads=0, row_order=ro)
    inst_off, inst_ent, kidx = ocr_plan_ref(cm.values_with_halo, cm.values_with_halo, n, rb, sp.rowptr, sp.colidx, pinv=pinv)
    assert np.array_equal(op.inst_off_host, inst_off)
    ent = np.empty(op.ninst, dtype=np.int32)
    _lib.call("fd_memcpy_d2h", ent.ctypes.data, op.inst_ent, ent.nbytes, None)
    assert np.array_equal(ent, inst_ent)
    assert np.array_equal(op.kidx.download(np.uint8, kidx.shape), kidx)


@pytest.mark.parametrize("chunk", [0, 997, 70000])
def test_sparsity_built_in_chunks_equals_the_oracle_pattern(chunk, monkeypatch):
    """fd_csr_from_maps emits / sorts / uniques the candidate entries in chunks of at most 2^30 keys and merges the unique keys
    in an accumulator (the 215^3 CG2 half-cube of BASELINE configs[4] has 3.0e9 candidates: more than a 32-bit item count).
    FDHIP_CSR_CHUNK shrinks the chunk so that a small mesh takes the many-chunk path, including the accumulator's own
    compaction; extruded pair, variable layers and a rectangular pair included."""
    from firedrake_amd import mesh as fmesh
    from helpers import oracle_pattern
    if chunk:
        monkeypatch.setenv("FDHIP_CSR_CHUNK", str(chunk))
    m = fmesh.UnitCubeMesh(5, degrees=(1, 2), perturb=0.1, numbering="random")
    V1, V2 = m.space(1), m.space(2)
    for rows, cols in ((V2, V2), (V1, V2)):
        sp = op2.Sparsity((rows.node_set ** 1, cols.node_set ** 1), [(rows.cell_node_map, cols.cell_node_map, None)])
        ref = oracle_pattern(sp)
        assert np.array_equal(sp.rowptr, ref.rowptr) and np.array_equal(sp.colidx, ref.colidx)
    if chunk == 997:
        return                   # (a chunk holds whole entity columns: one Q4 column emits 4 x 125 x 125 candidates)
    hm = fmesh.make_extruded_hex_mesh(3, 4, 4, perturb=0.1)
    sp = op2.Sparsity((hm.node_set ** 1, hm.node_set ** 1), [(hm.cell_node_map, hm.cell_node_map, None)])
    ref = oracle_pattern(sp)
    assert np.array_equal(sp.rowptr, ref.rowptr) and np.array_equal(sp.colidx, ref.colidx)
