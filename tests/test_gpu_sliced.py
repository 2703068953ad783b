"""GPU: the row-sliced owner-computes-rows path (codegen.generate_sliced_wrapper, fd_ocrplan_create_sliced) -- selected for
matrix loops whose row map has >= 8 entries (P2 tetrahedra: 10).  Plan arrays against the numpy restatement, assembled
matrices against the oracle's MatSetValuesLocal (builder.py:573-625) and against the unsliced wrapper, lgmap pairs swapped
between calls like the reference does per assemble (parloop.py:279-314)."""
import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import forms, mesh as fmesh, op2
from firedrake_amd.configuration import configuration
from helpers import oracle_run

pytestmark = pytest.mark.gpu


def _oracle_jac(prob, pl, mat):
    mpa = pl.arguments[0]
    args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
    return oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]


@pytest.mark.parametrize("ordered,interleave", [(False, 0), (True, 0), (False, 7)])
def test_sliced_plan_matches_numpy_restatement(ordered, interleave, monkeypatch):
    from firedrake_amd import _lib
    from firedrake_amd.device import DeviceBuffer
    from firedrake_amd.op2types import RowOrder, SlicedOcrPlan
    from helpers import first_touch_ref, locality_order_ref, ocrs_plan_ref
    monkeypatch.setitem(configuration, "ocrs_interleave", interleave)
    m = fmesh.UnitCubeMesh(4, degrees=(2,), perturb=0.1, numbering="random" if ordered else "tiled")
    V = m.space(2)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    sp._build()
    n, nrows = m.cell_set.size, V.node_set.size
    rp = np.asarray(sp.rowptr)
    ro, pinv, acc_node, acc_pos = None, None, rp, rp
    if ordered:
        order = locality_order_ref(m.coord_space.cell_node_map.values_with_halo, 0, n, np.array(m.coordinates.data_ro), target=64)[0]
        ro = RowOrder(cm, DeviceBuffer.from_numpy(order), n, nrows, rp)
        plist, pinv = first_touch_ref(cm.values_with_halo, order, nrows)
        acc_pos = np.concatenate([[0], np.cumsum(np.diff(rp)[:nrows][plist])]).astype(np.int32)
        acc_node = np.zeros(nrows, dtype=np.int32)
        acc_node[plist] = acc_pos[:-1]
    rb = np.unique(np.concatenate([np.arange(0, nrows, 37), [nrows]])).astype(np.int32)
    op = SlicedOcrPlan(sp, cm, cm, {}, 0, n, rb, row_order=ro)
    rng = np.random.default_rng(5)
    rlg = np.arange(V.node_set.total_size, dtype=np.int32)
    rlg[rng.choice(nrows, nrows // 7, replace=False)] = -1
    clg = np.arange(V.node_set.total_size, dtype=np.int32)
    clg[rng.choice(nrows, nrows // 5, replace=False)] = -1
    for lgs in ((None, None), (rlg, clg)):
        inst_off, ent, role, valid, slot, kk, _ = ocrs_plan_ref(np.asarray(cm.values_with_halo), np.asarray(cm.values_with_halo), 0, n, rb,
                                                            rp, np.asarray(sp.colidx), acc_node, acc_pos, pinv=pinv, rlg=lgs[0], clg=lgs[1], interleave=interleave)
        assert np.array_equal(op.inst_off_host, inst_off) and op.ninst == len(ent) and op.nreal == int(valid.sum())
        assert op.nreal == int(((np.asarray(cm.values_with_halo)[:n] < nrows)).sum())      # every (entity, owned row) pair exactly once

        def down(ptr, dt, shape):
            a = np.empty(shape, dtype=dt)
            _lib.call("fd_memcpy_d2h", a.ctypes.data, ptr, a.nbytes, None)
            return a
        assert np.array_equal(down(op.inst_ent, np.int32, (op.ninst,)), ent)
        assert np.array_equal(down(op.chunk_role, np.uint8, (op.ninst // 64,)), role)
        assert np.array_equal(down(op.valid, np.uint8, (op.ninst,)), valid)
        keep = []
        s_, k_, *_ = op.tables(lgs[0], lgs[1], lambda a: keep.append(DeviceBuffer.from_numpy(a)) or keep[-1].ptr)
        assert np.array_equal(s_.download(np.uint16, (op.ninst,)), slot)
        assert np.array_equal(k_.download(np.uint8, kk.shape), kk)


@pytest.mark.parametrize("ordered,interleave", [(False, 0), (True, 0), (True, 7)])
def test_paired_plan_matches_numpy_restatement(ordered, interleave, monkeypatch):
    """Two rows per instance (fd_ocrplan_create_paired, fd_ocrplan_pair_counts, two table rows per instance, fd_ocr_pack_records_rows)
    against the numpy restatements; every (entity, owned row) pair is served by exactly one live table row."""
    from firedrake_amd import _lib
    from firedrake_amd.codegen import sliced_record_layout
    from firedrake_amd.device import DeviceBuffer
    from firedrake_amd.op2types import RowOrder, SlicedOcrPlan
    from helpers import choose_groups_ref, first_touch_ref, locality_order_ref, ocrs_pair_counts_ref, ocrs_paired_plan_ref, plan_ref_blocks
    from hostsim import pack_records_ref
    monkeypatch.setitem(configuration, "ocrs_interleave", interleave)
    m = fmesh.UnitCubeMesh(4, degrees=(2,), perturb=0.1, numbering="random" if ordered else "tiled")
    V = m.space(2)
    cm, xm = V.cell_node_map, m.coord_space.cell_node_map
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    sp._build()
    n, nrows = m.cell_set.size, V.node_set.size
    rp = np.asarray(sp.rowptr)
    ro, pinv, acc_node, acc_pos = None, None, rp, rp
    if ordered:
        order = locality_order_ref(xm.values_with_halo, 0, n, np.array(m.coordinates.data_ro), target=64)[0]
        ro = RowOrder(cm, DeviceBuffer.from_numpy(order), n, nrows, rp)
        plist, pinv = first_touch_ref(cm.values_with_halo, order, nrows)
        acc_pos = np.concatenate([[0], np.cumsum(np.diff(rp)[:nrows][plist])]).astype(np.int32)
        acc_node = np.zeros(nrows, dtype=np.int32)
        acc_node[plist] = acc_pos[:-1]
    rb = np.unique(np.concatenate([np.arange(0, nrows, 37), [nrows]])).astype(np.int32)
    cmv = np.asarray(cm.values_with_halo)
    cnt = ocrs_pair_counts_ref(cmv, 0, n, rb, pinv=pinv)
    groups = SlicedOcrPlan.choose_groups(cm, 0, n, rb, ro)
    assert groups == choose_groups_ref(cnt) and len(groups) == 5 and all(b is not None for _, b in groups)
    op = SlicedOcrPlan(sp, cm, cm, {0: xm}, 0, n, rb, row_order=ro, groups=groups)
    rng = np.random.default_rng(5)
    rlg = np.arange(V.node_set.total_size, dtype=np.int32)
    rlg[rng.choice(nrows, nrows // 7, replace=False)] = -1
    clg = np.arange(V.node_set.total_size, dtype=np.int32)
    clg[rng.choice(nrows, nrows // 5, replace=False)] = -1

    def down(ptr, dt, shape):
        a = np.empty(shape, dtype=dt)
        _lib.call("fd_memcpy_d2h", a.ctypes.data, ptr, a.nbytes, None)
        return a

    for lgs in ((None, None), (rlg, clg)):
        inst_off, ent, role, valid, slot, kk = ocrs_paired_plan_ref(cmv, cmv, 0, n, rb, rp, np.asarray(sp.colidx), acc_node, acc_pos, groups,
                                                                   pinv=pinv, rlg=lgs[0], clg=lgs[1], interleave=interleave)
        assert np.array_equal(op.inst_off_host, inst_off) and op.ninst == len(ent) and op.nreal == int(valid.sum())
        assert op.nreal < int((cmv[:n] < nrows).sum())                  # fewer instances than (entity, row) pairs
        assert np.array_equal(down(op.inst_ent, np.int32, (op.ninst,)), ent)
        assert np.array_equal(down(op.chunk_role, np.uint8, (op.ninst // 64,)), role)
        assert np.array_equal(down(op.valid, np.uint8, (op.ninst,)), valid)
        keep = []
        s_, k_, *_ = op.tables(lgs[0], lgs[1], lambda a: keep.append(DeviceBuffer.from_numpy(a)) or keep[-1].ptr)
        assert np.array_equal(s_.download(np.uint16, slot.shape), slot)
        assert np.array_equal(k_.download(np.uint8, kk.shape), kk)
        if lgs[0] is None:
            # every (entity, owned row) pair has exactly ONE live table row
            live = slot != 0xffff
            rows_of = np.array([[(-1 if r is None else r) for r in g] for g in groups])[np.repeat(role, 64)]
            served = np.zeros(cmv[:n].shape, dtype=np.int32)
            for s2 in range(2):
                sel = live[:, s2]
                np.add.at(served, (ent[sel], rows_of[sel, s2]), 1)
            assert np.array_equal(served, (cmv[:n] < nrows).astype(np.int32))
        # records: the coordinate map's local rows + two rows of positions + two slots
        lm = plan_ref_blocks(np.asarray(xm.values_with_halo)[ent], inst_off)[2]
        nd = op.plans[0].max_nd
        lbits, kbits, sbits, words = sliced_record_layout([xm.arity], [nd], cm.arity, int(np.diff(rp).max()), op.max_nnz, rows=2)
        rec = op.records(lgs[0], lgs[1], lambda a: keep.append(DeviceBuffer.from_numpy(a)) or keep[-1].ptr, [0], lbits, kbits, sbits, words)
        ref = pack_records_ref([lm], lbits, kk.reshape(len(ent), -1), 2, cm.arity, kbits, False, words, extra=slot, ebits=sbits, sentinel=True)
        assert np.array_equal(rec.download(np.uint32, (op.ninst, words)), ref)


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic", "random"])
@pytest.mark.parametrize("pairs", [0, 1])
def test_p2_jacobian_paired_instances_against_oracle(numbering, pairs, monkeypatch):
    """The default of scalar row-sliced loops since round 6: two rows per instance ("_g" variants).  Same matrix as one row per
    instance and as the oracle, with BCs (dropped rows and columns), accumulation on top, and lgmaps swapped between calls."""
    monkeypatch.setitem(configuration, "locality_min_entities", 0)
    monkeypatch.setitem(configuration, "ocrs_pairs", pairs)
    m = fmesh.UnitCubeMesh(7, degrees=(2,), tile=(4, 4, 2), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 2, bcs=True)
    mat, pl = prob.jacobian()
    mat.zero()
    pl()
    geo = pl._ocr_geometry()
    assert ("_g" in geo["cw"].src.mode) == bool(pairs) and (geo["groups"] is not None) == bool(pairs)
    if pairs and numbering != "random":
        assert geo["ocr"].nreal < m.cell_set.size * 10 * 0.75          # most rows found a partner in their block
    ref = _oracle_jac(prob, pl, mat)
    assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    pl()
    assert_allclose(mat.csr()[2], 2.0 * ref.values, rtol=0, atol=2e-12 * np.abs(ref.values).max())
    mpa = pl.arguments[0]
    nn = prob.V.node_set.total_size
    lg = np.arange(nn, dtype=np.int32)
    lg[np.random.default_rng(3).choice(nn, nn // 6, replace=False)] = -1
    mpa.lgmaps = (lg, np.arange(nn, dtype=np.int32))
    mat.zero()
    pl()
    ref2 = _oracle_jac(prob, pl, mat)
    assert_allclose(mat.csr()[2], ref2.values, rtol=0, atol=1e-12 * np.abs(ref2.values).max())


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic", "random"])
@pytest.mark.parametrize("bcs", [False, True])
def test_p2_jacobian_sliced_against_oracle_and_unsliced(numbering, bcs, monkeypatch):
    monkeypatch.setitem(configuration, "locality_min_entities", 0)
    m = fmesh.UnitCubeMesh(7, degrees=(2,), tile=(4, 4, 2), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 2, bcs=bcs)
    mat, pl = prob.jacobian()
    pl._ensure_geometry()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs")
    geo = pl._ocr_geometry()
    assert geo["cw"].src.mode.startswith("ocrsp" if numbering != "tiled" else "ocrs")
    mat.zero()
    pl()
    ref = _oracle_jac(prob, pl, mat)
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    pl()                                                  # no zero(): ADD_VALUES on top of the first assembly
    assert_allclose(mat.csr()[2], 2.0 * ref.values, rtol=0, atol=2e-12 * np.abs(ref.values).max())
    # the unsliced owner-computes-rows wrapper on the same problem
    monkeypatch.setitem(configuration, "ocr_sliced", 0)
    prob2 = forms.PoissonProblem(m, 2, bcs=bcs)
    mat2, pl2 = prob2.jacobian()
    mat2.zero()
    pl2()
    assert not pl2._prepare()["cw"].src.mode.startswith("ocrs")
    assert_allclose(mat2.csr()[2], v, rtol=0, atol=1e-12 * np.abs(ref.values).max())


def test_lgmap_pairs_swap_between_calls():
    """One Parloop, three lgmap pairs in turn (the reference swaps the Mat's lgmaps per call): each pair gets its own slot /
    column-position tables, earlier pairs are reused, results match the oracle with that pair."""
    m = fmesh.UnitCubeMesh(5, degrees=(2,), tile=(4, 4, 2), perturb=0.1)
    prob = forms.PoissonProblem(m, 2, bcs=True)
    mat, pl = prob.jacobian()
    mpa = pl.arguments[0]
    nn = prob.V.node_set.total_size
    rng = np.random.default_rng(11)
    pairs = [mpa.lgmaps]
    for frac in (9, 4):
        lg = np.arange(nn, dtype=np.int32)
        lg[rng.choice(nn, nn // frac, replace=False)] = -1
        pairs.append((lg, np.arange(nn, dtype=np.int32)))
    for lgs in pairs + pairs[:1]:
        mpa.lgmaps = lgs
        mat.zero()
        pl()
        ref = _oracle_jac(prob, pl, mat)
        assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    assert len(pl._ocr_geometry()["ocr"]._tables) == 3


def test_sliced_with_direct_and_global_read_arguments():
    """A high-arity matrix loop whose kernel also reads a direct (cell-wise) Dat and a Global: both reach every
    instantiation unchanged."""
    m = fmesh.UnitCubeMesh(4, degrees=(2,), tile=(4, 4, 2), perturb=0.1)
    V = m.space(2)
    cm, xm = V.cell_node_map, m.coord_space.cell_node_map
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(cm, cm, None)])
    mat = op2.Mat(sp)
    w = op2.Dat(m.cell_set, np.random.default_rng(2).uniform(0.5, 1.5, m.cell_set.total_size), np.float64)
    g = op2.Global(2, [0.25, -1.5], np.float64)
    k = op2.Kernel("""
static void wk(double *A, const double *x, const double *w, const double *g)
{
  const double d = (x[3] - x[0]) * (x[7] - x[1]) + x[11] * g[0];
  for (int i = 0; i < 10; ++i)
    for (int j = 0; j < 10; ++j)
      A[i*10 + j] += w[0] * (d + g[1] * (i + 1)) * (1.0 + 0.125 * j) + (i == j ? x[2] : 0.0);
}""", "wk")
    pl = op2.LegacyParloop(k, m.cell_set, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), w(op2.READ), g(op2.READ))
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs")
    ref = oracle_run(k, m.cell_set, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), w(op2.READ), g(op2.READ))[0]
    assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


def test_long_rows_take_16_bit_column_positions(monkeypatch):
    """A hub node shared by 300 eight-node cells: its CSR row has 2101 entries, so the column positions no longer fit a
    byte ("ocrs_k16") and the hub's block is cut to that one long row."""
    monkeypatch.setitem(configuration, "ocr_sliced_min_arity", 8)
    ncell, ar = 300, 8
    nn = 1 + ncell * (ar - 1)
    nodes, cells = op2.Set(nn), op2.Set(ncell)
    mv = np.zeros((ncell, ar), dtype=np.int32)
    mv[:, 1:] = 1 + np.arange(ncell * (ar - 1)).reshape(ncell, ar - 1)
    rng = np.random.default_rng(4)
    mv = np.stack([rng.permutation(r) for r in mv])              # the hub sits at a different local index in every cell
    m = op2.Map(cells, nodes, ar, mv)
    pos = op2.Dat(nodes ** 2, rng.uniform(0, 1, (nn, 2)), np.float64)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.Kernel("""
static void hub(double *A, const double *x)
{
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 8; ++j)
      A[i*8 + j] += (1.0 + i) * x[2*j] - 0.5 * j * x[2*i + 1];
}""", "hub")
    pl = op2.LegacyParloop(k, cells, mat(op2.INC, (m, m)), pos(op2.READ, m))
    pl()
    geo = pl._ocr_geometry()
    assert geo["cw"].src.mode.startswith("ocrs") and "_k16" in geo["cw"].src.mode and geo["ocr"].max_nnz >= 2101
    ref = oracle_run(k, cells, mat(op2.INC, (m, m)), pos(op2.READ, m))[0]
    assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic"])
def test_vector_valued_blocks_against_oracle(numbering, monkeypatch):
    """MatSetValuesBlockedLocal (builder.py:573-625): vector P1 on tetrahedra -- 12x12 element matrices in 3x3 blocks -- takes
    the row-sliced wrapper (an instance owns the three scalar rows of one node), with block lgmaps, a second accumulating
    call, and against the direct wrapper's global atomics."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from mixed_cases import vector_p1_elasticity_kernel
    monkeypatch.setitem(configuration, "locality_min_entities", 0)
    mesh = fmesh.UnitCubeMesh(9, degrees=(1,), tile=(4, 4, 2), perturb=0.1, numbering=numbering)
    V = mesh.space(1)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 3, V.node_set ** 3), [(cm, cm, None)])
    lgv = np.arange(V.node_set.total_size, dtype=np.int32)
    lgv[np.random.default_rng(1).choice(V.node_set.size, V.node_set.size // 6, replace=False)] = -1
    k = vector_p1_elasticity_kernel(3)
    mat = op2.Mat(sp)
    pl = op2.LegacyParloop(k, mesh.cell_set, mat(op2.INC, (cm, cm), lgmaps=(lgv, lgv)), mesh.coordinates(op2.READ, cm))
    pl()
    geo = pl._ocr_geometry()
    assert geo["cw"].src.mode.startswith("ocrsp" if numbering != "tiled" else "ocrs") and geo["ocr"].block == 9
    ref = oracle_run(k, mesh.cell_set, mat(op2.INC, (cm, cm), lgmaps=(lgv, lgv)), mesh.coordinates(op2.READ, cm))[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    pl()
    assert_allclose(mat.csr()[2], 2.0 * ref.values, rtol=0, atol=2e-12 * np.abs(ref.values).max())
    monkeypatch.setitem(configuration, "mat_ocr", 0)
    mat2 = op2.Mat(sp)
    pl2 = op2.LegacyParloop(k, mesh.cell_set, mat2(op2.INC, (cm, cm), lgmaps=(lgv, lgv)), mesh.coordinates(op2.READ, cm))
    pl2()
    assert not pl2._prepare()["cw"].src.mode.startswith("ocr")
    assert_allclose(mat2.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


def test_reference_vector_matrix_golden_through_the_sliced_wrapper(monkeypatch):
    """The reference's own expected vector-valued mass matrix (tests/pyop2/test_matrices.py, lifted into
    tests/golden/pyop2_matrices.json) with the row-sliced wrapper forced onto its 6x6 element matrices."""
    import golden_kernels as gk
    monkeypatch.setitem(configuration, "ocr_sliced_min_arity", 1)
    nodes, ele = op2.Set(4), op2.Set(2)
    m = op2.Map(ele, nodes, 3, gk.ELEM_NODE)
    mat = op2.Mat(op2.Sparsity((nodes ** 2, nodes ** 2), [(m, m, None)]))
    x = op2.Dat(nodes ** 2, gk.COORDS)
    pl = op2.LegacyParloop(op2.Kernel(gk.MASS_VEC_AFFINE, "mass_vec_affine"), ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs")
    assert_allclose(mat.values, np.array(gk.GOLD["expected_vector_matrix"]), rtol=1e-6, atol=1e-8)


@pytest.mark.parametrize("region", [None, "bottom"])
@pytest.mark.parametrize("sliced,ordered", [(False, False), (True, False), (False, True), (True, True)])
def test_matrix_over_an_extruded_set_against_oracle_and_direct(region, sliced, ordered, monkeypatch):
    """Extruded matrix assembly (node = map + offset*layer, builder.py:94-124) through both owner-computes-rows wrappers over
    the derived (column, layer) map: Q1 Helmholtz on a perturbed hex column mesh (8 rows: row-sliced by default, whole-entity
    instances when the threshold is raised), against the oracle and the direct wrapper."""
    from mixed_cases import q1_hex_helmholtz_kernel
    monkeypatch.setitem(configuration, "ocr_sliced_min_arity", 8 if sliced else 10)
    # ordered: the backend-derived row order on the virtual space (meshes this small keep the caller's rows by default)
    monkeypatch.setitem(configuration, "locality_min_entities", 0 if ordered else 1 << 30)
    m = fmesh.make_extruded_hex_mesh(12, 9, degree=1)
    cm, xm = m.cell_node_map, m.coord_map
    sp = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, None)])
    k = q1_hex_helmholtz_kernel()
    kw = {"iteration_region": op2.ON_BOTTOM} if region else {}
    mat = op2.Mat(sp)
    pl = op2.LegacyParloop(k, m.cell_set, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), **kw)
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs") == sliced and pl._prepare()["cw"].src.mode.startswith("ocr")
    assert (pl._ocr_geometry()["row_order"] is not None) == ordered
    ref = oracle_run(k, m.cell_set, mat(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), **kw)[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    monkeypatch.setitem(configuration, "mat_ocr", 0)
    mat2 = op2.Mat(sp)
    pl2 = op2.LegacyParloop(k, m.cell_set, mat2(op2.INC, (cm, cm)), m.coordinates(op2.READ, xm), **kw)
    pl2()
    assert pl2._prepare()["cw"].src.mode == "direct"
    assert_allclose(mat2.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())


def test_per_dof_lgmaps_against_oracle():
    """``unroll_map`` (MatSetValuesLocal on dof indices, mat.py:700-716): component-wise Dirichlet conditions on a vector
    space travel as per-instance row / column bit masks; against the oracle, and the masks against the numpy restatement."""
    from helpers import ocrs_plan_ref
    from mixed_cases import vector_p1_elasticity_kernel
    mesh = fmesh.UnitCubeMesh(8, degrees=(1,), tile=(4, 4, 2), perturb=0.1)
    V = mesh.space(1)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 3, V.node_set ** 3), [(cm, cm, None)])
    mat = op2.Mat(sp)
    rng = np.random.default_rng(6)
    nn = V.node_set.total_size
    rlg, clg = np.arange(3 * nn, dtype=np.int32), np.arange(3 * nn, dtype=np.int32)
    fx, fz = rng.choice(nn, nn // 4, replace=False), rng.choice(nn, nn // 5, replace=False)
    rlg[3 * fx] = -1
    rlg[3 * fz + 2] = -1
    clg[3 * fx] = -1
    clg[3 * rng.choice(nn, nn // 7, replace=False) + 1] = -1
    k = vector_p1_elasticity_kernel(3)
    args = lambda: (mat(op2.INC, (cm, cm), lgmaps=(rlg, clg), unroll_map=True), mesh.coordinates(op2.READ, cm))
    pl = op2.LegacyParloop(k, mesh.cell_set, *args())
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs")
    ref = oracle_run(k, mesh.cell_set, *args())[0]
    assert_allclose(mat.csr()[2], ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
    op = pl._ocr_geometry()["ocr"]
    tabs = op.tables(rlg, clg, pl._lgmap, per_dof=True)
    nrp = np.asarray(sp._node_rowptr_host())
    nci = np.asarray(sp._node_colidx.download(np.int32, (sp._node_nnz,)))
    res = ocrs_plan_ref(np.asarray(cm.values_with_halo), np.asarray(cm.values_with_halo), 0, mesh.cell_set.size, op.row_blocks, nrp, nci, nrp, nrp,
                        rlg=rlg, clg=clg, interleave=configuration["ocrs_interleave"], per_dof=(3, 3))
    assert np.array_equal(tabs[0].download(np.uint16, (op.ninst,)), res[4])
    assert np.array_equal(tabs[3].download(np.uint8, (op.ninst,)), res[7])
    assert np.array_equal(tabs[4].download(np.uint64, (op.ninst,)), res[8])


@pytest.mark.parametrize("ar", [10, 4])
def test_negative_entries_in_matrix_maps(ar):
    """MatSetValuesLocal ignores negative indices (builder.py:573-625): a matrix map with -1 entries runs through the
    row-sliced wrapper (rows never instantiated, columns never positioned) when its element matrix is large enough, and is
    demoted to the direct wrapper otherwise (whole-entity instance tables and block plans do not take such maps)."""
    rng = np.random.default_rng(12)
    nn, ne = 900, 700
    mv = np.stack([rng.choice(nn, ar, replace=False) for _ in range(ne)]).astype(np.int32)
    mv[rng.random(mv.shape) < 0.15] = -1
    nodes, ele = op2.Set(nn), op2.Set(ne)
    m = op2.Map(ele, nodes, ar, mv)
    xs = op2.Dat(ele ** 2, rng.uniform(0, 1, (ne, 2)), np.float64)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.Kernel(f"""
static void neg{ar}(double *A, const double *w)
{{
  for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {ar}; ++j) A[i*{ar} + j] += w[0] * (i + 1) + w[1] * j;
}}""", f"neg{ar}")
    pl = op2.LegacyParloop(k, ele, mat(op2.INC, (m, m)), xs(op2.READ))
    pl()
    assert pl._prepare()["cw"].src.mode.startswith("ocrs" if ar == 10 else "direct")
    ref = oracle_run(k, ele, mat(op2.INC, (m, m)), xs(op2.READ))[0]
    rp, ci, v = mat.csr()
    assert np.array_equal(rp, ref.rowptr) and np.array_equal(ci, ref.colidx)
    assert_allclose(v, ref.values, rtol=0, atol=1e-12 * np.abs(ref.values).max())
