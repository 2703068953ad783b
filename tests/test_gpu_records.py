"""GPU: the compact index tables of the whole-entity owner-computes-rows wrapper (round 4) -- bit-packed instance records
(fd_ocr_pack_records + fd_ocr_node_diag) and the run-coded flush of a derived row order (fd_ocr_row_runs) -- against their
numpy restatements (tests/hostsim.py) and, end to end, against the oracle's MatSetValuesLocal (builder.py:573-625) with every
combination of the switches."""
import ctypes

import numpy as np
import pytest

from firedrake_amd import _lib, forms, mesh as fmesh, op2
from firedrake_amd.configuration import configuration
from firedrake_amd.device import DeviceBuffer
from helpers import oracle_run

pytestmark = pytest.mark.gpu


def _down(ptr, dt, shape):
    a = np.empty(shape, dtype=dt)
    if a.nbytes:
        _lib.call("fd_memcpy_d2h", a.ctypes.data, ptr, a.nbytes, None)
    return a


@pytest.mark.parametrize("diag", [False, True])
def test_pack_records_matches_numpy_restatement(diag):
    from hostsim import pack_records_ref
    rng = np.random.default_rng(3)
    ninst, nr, nc = 1000, 4, 4
    lm0 = rng.integers(0, 500, (ninst, 4)).astype(np.uint16)
    lm1 = rng.integers(0, 37, (ninst, 3)).astype(np.uint16)
    kidx = rng.integers(0, 27, (ninst, nr * nc)).astype(np.uint8)
    lbits, kbits = [9, 6], 5
    bits = 4 * 9 + 3 * 6 + (nr * nc - (nr if diag else 0)) * kbits
    words = -(-bits // 32)
    d0, d1, dk = DeviceBuffer.from_numpy(lm0), DeviceBuffer.from_numpy(lm1), DeviceBuffer.from_numpy(kidx)
    out = DeviceBuffer(ninst * words * 4)
    lm = (ctypes.c_void_p * 2)(d0.ptr, d1.ptr)
    ar = (ctypes.c_int32 * 2)(4, 3)
    lb = (ctypes.c_int32 * 2)(*lbits)
    _lib.call("fd_ocr_pack_records", ninst, 2, lm, ar, lb, dk.ptr, 1, nr, nc, kbits, int(diag), None, 0, 0, words, out.ptr, None)
    got = _down(out.ptr, np.uint32, (ninst, words))
    ref = pack_records_ref([lm0, lm1], lbits, kidx, nr, nc, kbits, diag, words)
    assert np.array_equal(got, ref)
    # an index that does not fit its field is an error, not a truncation
    with pytest.raises(_lib.FDHipError):
        _lib.call("fd_ocr_pack_records", ninst, 2, lm, ar, (ctypes.c_int32 * 2)(8, 6), dk.ptr, 1, nr, nc, kbits, int(diag), None, 0, 0,
                  -(-(4 * 8 + 3 * 6 + (nr * nc - (nr if diag else 0)) * kbits) // 32), out.ptr, None)
    if not diag:
        # row-sliced form: one row of nc positions + the slot as a last field; 0xff / 0xffff (dropped) become all ones of the field
        kk = rng.integers(0, 27, (ninst, 10)).astype(np.uint8)
        kk[rng.random((ninst, 10)) < 0.1] = 0xff
        slot = rng.integers(0, 4000, ninst).astype(np.uint16)
        slot[rng.random(ninst) < 0.05] = 0xffff
        dkk, dsl = DeviceBuffer.from_numpy(kk), DeviceBuffer.from_numpy(slot)
        w2 = -(-(4 * 9 + 10 * 5 + 12) // 32)
        out2 = DeviceBuffer(ninst * w2 * 4)
        _lib.call("fd_ocr_pack_records", ninst, 1, (ctypes.c_void_p * 1)(d0.ptr), (ctypes.c_int32 * 1)(4), (ctypes.c_int32 * 1)(9), dkk.ptr, 1, 1, 10,
                  5, 0, dsl.ptr, 12, 1, w2, out2.ptr, None)
        ref2 = pack_records_ref([lm0], [9], kk, 1, 10, 5, False, w2, extra=slot, ebits=12, sentinel=True)
        assert np.array_equal(_down(out2.ptr, np.uint32, (ninst, w2)), ref2)


def test_row_runs_match_numpy_restatement():
    from hostsim import row_runs_ref
    rng = np.random.default_rng(11)
    npos = 5000
    rowlen = rng.integers(1, 30, npos)
    # a row order made of runs of consecutive rows (what a k-d leaf of a lexicographic numbering looks like)
    starts = np.sort(rng.choice(npos, 400, replace=False))
    starts[0] = 0
    pieces = np.split(np.arange(npos), starts[1:])
    plist = np.concatenate([pieces[i] for i in rng.permutation(len(pieces))])
    rp = np.concatenate([[0], np.cumsum(rowlen)]).astype(_lib.NNZ_DTYPE)             # (row starts: fd_nnz_t)
    prowptr = np.concatenate([[0], np.cumsum(rowlen[plist])]).astype(_lib.NNZ_DTYPE)
    gstart = rp[plist].astype(_lib.NNZ_DTYPE)
    rb = np.unique(np.concatenate([np.sort(rng.choice(npos, 60, replace=False)), [0, npos]])).astype(np.int32)
    nb = len(rb) - 1
    dp, dg, dr = DeviceBuffer.from_numpy(prowptr), DeviceBuffer.from_numpy(gstart), DeviceBuffer.from_numpy(rb)
    grun, brun, rdelta = DeviceBuffer(int(prowptr[-1])), DeviceBuffer((nb + 1) * 4), DeviceBuffer(npos * _lib.NNZ_BYTES)
    nruns, mx = ctypes.c_int32(), ctypes.c_int32()
    _lib.call("fd_ocr_row_runs", npos, dp.ptr, dg.ptr, dr.ptr, nb, grun.ptr, brun.ptr, rdelta.ptr, ctypes.byref(nruns), ctypes.byref(mx), None)
    g_ref, b_ref, d_ref, mx_ref = row_runs_ref(prowptr, gstart, rb)
    assert mx.value == mx_ref and nruns.value == b_ref[-1]
    assert np.array_equal(_down(brun.ptr, np.int32, (nb + 1,)), b_ref)
    assert np.array_equal(_down(rdelta.ptr, _lib.NNZ_DTYPE, (nruns.value,)), d_ref[:nruns.value])
    assert np.array_equal(_down(grun.ptr, np.uint8, (int(prowptr[-1]),)), g_ref[:int(prowptr[-1])])
    # the decoded places are exactly the per-entry table of the plain "ocrp" flush
    gpos = np.concatenate([np.arange(gstart[p], gstart[p] + rowlen[plist[p]]) for p in range(npos)])
    blk = np.repeat(np.arange(nb), np.diff(prowptr[rb]))
    assert np.array_equal(np.arange(int(prowptr[-1])) + d_ref[b_ref[blk] + g_ref[:int(prowptr[-1])]], gpos)


def test_masked_entry_positions_match_numpy_restatement():
    from hostsim import row_entry_positions_masked_ref
    rng = np.random.default_rng(2)
    npos, ncol = 3000, 3500
    rowlen = rng.integers(1, 25, npos)
    plist = rng.permutation(npos)
    rp = np.concatenate([[0], np.cumsum(rowlen)]).astype(_lib.NNZ_DTYPE)
    colidx = rng.integers(0, ncol, int(rp[-1])).astype(np.int32)
    prowptr = np.concatenate([[0], np.cumsum(rowlen[plist])]).astype(_lib.NNZ_DTYPE)
    gstart = rp[plist].astype(_lib.NNZ_DTYPE)
    clg = np.arange(ncol, dtype=np.int32)
    clg[rng.choice(ncol, ncol // 6, replace=False)] = -1
    bufs = [DeviceBuffer.from_numpy(a) for a in (prowptr, gstart, colidx, clg)]
    out, plain = DeviceBuffer(int(rp[-1]) * 4), DeviceBuffer(int(rp[-1]) * 4)
    _lib.call("fd_row_entry_positions", npos, bufs[0].ptr, bufs[1].ptr, plain.ptr, None)
    _lib.call("fd_row_entry_positions_masked", npos, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, bufs[3].ptr, out.ptr, None)
    gp = _down(plain.ptr, np.int32, (int(rp[-1]),))
    assert np.array_equal(gp, np.concatenate([np.arange(gstart[p], gstart[p] + rowlen[plist[p]]) for p in range(npos)]))
    got = _down(out.ptr, np.int32, (int(rp[-1]),))
    assert np.array_equal(got, row_entry_positions_masked_ref(gp, colidx, clg)) and (got < -1).sum() > 0


@pytest.mark.parametrize("flush_colmask", [0, 1])
def test_p1_jacobian_column_mask_in_the_flush_matches_oracle(flush_colmask, monkeypatch):
    """"ocrpm": BC columns masked by the flush's place table.  Fresh assembly (the value array holds garbage: a pending zero() is no
    memset), accumulation on top (masked entries untouched), and a swap of the lgmaps between calls (a new masked table)."""
    monkeypatch.setitem(configuration, "ocr_flush_colmask", flush_colmask)
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(12, degrees=(1,), perturb=0.1, numbering="lexicographic")
    prob = forms.PoissonProblem(m, 1, bcs=True)
    mat, pl = prob.jacobian()
    mpa = pl.arguments[0]

    def oracle():
        args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
        return oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0].values

    mat._values_dev().upload(np.full(mat.sparsity.nz, 7.0))          # garbage under a pending zero()
    mat.zero()
    pl.compute()
    geo = [g for key, g in pl._prepared["parts"].items() if key[0] == "ocr"][0]
    assert geo["cw"].src.mode.startswith("ocrpm") == bool(flush_colmask)
    ref = oracle()
    assert np.abs(mat.csr()[2] - ref).max() <= 1e-12 * np.abs(ref).max()
    pl.compute()                                                       # ADD_VALUES on top
    assert np.abs(mat.csr()[2] - 2.0 * ref).max() <= 2e-12 * np.abs(ref).max()
    nn = prob.V.node_set.total_size
    lg = np.arange(nn, dtype=np.int32)
    lg[np.random.default_rng(4).choice(nn, nn // 5, replace=False)] = -1
    mpa.lgmaps = (lg, lg.copy())
    mat.zero()
    pl.compute()
    ref2 = oracle()
    assert np.abs(mat.csr()[2] - ref2).max() <= 1e-12 * np.abs(ref2).max()


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic", "random"])
@pytest.mark.parametrize("records,diag", [(0, 0), (1, 0), (1, 1)])
def test_p1_jacobian_with_compact_tables_matches_oracle(numbering, records, diag, monkeypatch):
    monkeypatch.setitem(configuration, "ocr_records", records)
    monkeypatch.setitem(configuration, "ocr_records_diag", diag)
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(12, degrees=(1,), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    mat, pl = prob.jacobian()
    for _ in range(2):                                   # second call: plan-ordered copies, cached tables
        mat.zero()
        pl.compute()
    geo = [g for key, g in pl._prepared["parts"].items() if key[0] == "ocr"][0]
    assert bool(geo["rec"]) == bool(records) and (geo["rec"][2] if records else False) == bool(diag)
    if numbering == "tiled":
        assert geo["cw"].src.mode.startswith("ocr_") or geo["cw"].src.mode == "ocr"
    elif numbering == "lexicographic":
        assert geo["cw"].src.mode.startswith("ocrpm_" if records else "ocrpm")      # (BCs: column mask in the flush's place table)
    mpa = pl.arguments[0]
    args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
    ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
    _, _, v = mat.csr()
    assert np.abs(v - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("numbering", ["tiled", "lexicographic"])
@pytest.mark.parametrize("records,runs", [(0, 0), (1, 0), (1, 1)])
def test_p2_jacobian_with_compact_tables_matches_oracle(numbering, records, runs, monkeypatch):
    """The row-sliced wrapper (CG2) with one record per instance and the run-coded flush of a derived row order, boundary rows and
    columns dropped (all-ones fields), against the oracle."""
    monkeypatch.setitem(configuration, "ocr_records", records)
    monkeypatch.setitem(configuration, "ocrs_run_flush", runs)
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(8, degrees=(2,), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 2, bcs=True)
    mat, pl = prob.jacobian()
    for _ in range(2):
        mat.zero()
        pl.compute()
    geo = [g for key, g in pl._prepared["parts"].items() if key[0] == "ocr"][0]
    mode = geo["cw"].src.mode
    assert mode.startswith("ocrs") and bool(geo["rec"]) == bool(records) and ("_q" in mode) == bool(records)
    if numbering == "lexicographic":
        assert mode.startswith("ocrspr") == bool(runs)
    mpa = pl.arguments[0]
    args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
    ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
    _, _, v = mat.csr()
    assert np.abs(v - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("numbering", ["lexicographic", "tiled"])
def test_fixed_point_accumulators_give_the_oracle_matrix_bit_reproducibly(numbering, monkeypatch):
    """Checked fixed-point accumulation (codegen mode "_fx", FDHIP_OCR_FIXED_POINT=1: opt-in for whole-entity owner-computes-rows loops): the first
    launch of a plan has no scales and runs fp64 blocks, each of which leaves the scale record of its next launch; from the second
    launch on the element matrices are reduced as 64-bit integers.  The P1 Jacobian is the oracle's to 1e-12 max|A| either way, and
    -- the sums being exact integers -- the same BITS whatever the order of the instances inside the blocks."""
    monkeypatch.setitem(configuration, "ocr_fixed_point", 1)          # opt-in since round 6 (normwise guarantee only)
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(14, degrees=(1,), perturb=0.1, numbering=numbering)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    mat, pl = prob.jacobian()
    mpa = pl.arguments[0]
    args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
    ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
    vmax = np.abs(ref.values).max()
    got = []
    for order in ("stencil", "natural"):                   # two different instance orders inside the blocks
        monkeypatch.setitem(configuration, "ocr_order", order)
        p2 = forms.PoissonProblem(m, 1, bcs=True)
        mat2, pl2 = p2.jacobian()
        mat2.zero()
        pl2.compute()                                      # no scales yet: fp64 blocks
        (st,) = pl2.fixed_point_state()
        assert st["unscaled_blocks"] == st["blocks"] and st["fallback_blocks"] == 0 and st["scaled_blocks"] >= st["blocks"] - 2
        assert np.abs(mat2.csr()[2] - ref.values).max() <= 1e-12 * vmax
        geo = [g for key, g in pl2._prepared["parts"].items() if key[0] == "ocr"][0]
        assert geo["cw"].src.mode.endswith("_fx")
        mat2.zero()
        pl2.compute()                                      # fixed-point blocks
        (st2,) = pl2.fixed_point_state()
        assert st2["fallback_blocks"] == 0 and st2["unscaled_blocks"] == 2 * st["blocks"] - st["scaled_blocks"]
        assert st2["limit_exponents"] == st["limit_exponents"]
        v = mat2.csr()[2]
        assert np.abs(v - ref.values).max() <= 1e-12 * vmax
        pl2.compute()                                      # accumulates on top
        assert np.abs(mat2.csr()[2] - 2.0 * ref.values).max() <= 2e-12 * vmax
        got.append(v)
    assert np.array_equal(got[0], got[1])


def test_fixed_point_blocks_fall_back_to_fp64_when_the_contributions_leave_the_window_of_their_scale(monkeypatch):
    """A block's scale comes from its PREVIOUS launch.  When its element matrices outgrow it (here: the mesh is stretched 64-fold
    between two assemblies, the P1 stiffness entries grow with it) or shrink so far that the scale has become too coarse (4096-fold)
    the block redoes its rows with fp64 atomics inside the same launch: the matrix is right, the fallbacks are counted, and the next
    launch runs fixed-point again at the new scales.  Non-finite contributions: the blocks that meet them fall back, NaN lands
    where fp64 puts it, the rest of the matrix is untouched."""
    monkeypatch.setitem(configuration, "ocr_fixed_point", 1)
    monkeypatch.setitem(configuration, "locality_min_entities", 64)
    m = fmesh.UnitCubeMesh(12, degrees=(1,), perturb=0.1, numbering="lexicographic")
    prob = forms.PoissonProblem(m, 1, bcs=True)
    mat, pl = prob.jacobian()
    mpa = pl.arguments[0]

    def reference():
        args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
        return oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0].values

    def assemble():
        mat.zero()
        pl.compute()
        return mat.csr()[2], pl.fixed_point_state()[0]

    ref = reference()
    assemble()
    v, st = assemble()
    assert st["fallback_blocks"] == 0 and np.abs(v - ref).max() <= 1e-12 * np.abs(ref).max()
    fell, lim = 0, st["limit_exponents"]
    for factor, shift in ((64.0, 6), (1.0 / 4096.0, -12)):
        m.coordinates.data[...] *= factor                   # (bumps dat_version: plan-ordered copies are dropped)
        ref = reference()
        v, st = assemble()                                  # launched at the old scales: every block with contributions falls back
        assert st["fallback_blocks"] - fell >= st["scaled_blocks"] - 2
        fell = st["fallback_blocks"]
        assert np.abs(v - ref).max() <= 1e-12 * np.abs(ref).max()
        assert st["limit_exponents"] == (lim[0] + shift, lim[1] + shift)        # ... and is re-scaled for the next launch
        lim = st["limit_exponents"]
        v, st = assemble()
        assert st["fallback_blocks"] == fell and st["limit_exponents"] == lim
        assert np.abs(v - ref).max() <= 1e-12 * np.abs(ref).max()
    x = m.coordinates.data
    x[5, 0] = np.nan
    v, st = assemble()
    assert st["fallback_blocks"] > fell and np.isnan(v).any() and np.isfinite(v).sum() > 0.5 * len(v)
