"""Test helpers: run the same parloop through the oracle (CPU restatement) for comparison."""
import numpy as np

import oracle
from firedrake_amd import op2
from firedrake_amd.parloop import DatLegacyArg, GlobalLegacyArg, MatLegacyArg


_REGION = {None: oracle.ALL, op2.ALL: oracle.ALL, op2.ON_BOTTOM: oracle.ON_BOTTOM, op2.ON_TOP: oracle.ON_TOP,
           op2.ON_INTERIOR_FACETS: oracle.ON_INTERIOR_FACETS}


def oracle_pattern(sp):
    """The CSR pattern of Sparsity ``sp`` built by the oracle (one contribution per map pair and iteration region)."""
    rds, cds = sp.dsets
    pairs = []
    for r, c, regions in sp._pairs:
        it = r.iterset
        if it._extruded:
            for reg in regions:
                nl = it.layers - 1 if it.constant_layers else np.asarray(it.layers_array, dtype=np.int32)
                pairs.append((r.values_with_halo, c.values_with_halo, nl, r.offset, c.offset,
                              r.offset_quotient, c.offset_quotient, bool(it._extruded_periodic), _REGION[reg]))
        else:
            pairs.append((r.values_with_halo, c.values_with_halo))
    return oracle.build_sparsity(rds.set.total_size, cds.set.total_size, pairs, rbs=rds.cdim, cbs=cds.cdim,
                                 set_diag=sp._has_diagonal)


def _odat(a, access, iterset):
    view = None
    if isinstance(a.data, op2.DatView):          # the oracle gets the parent's rows and the component to look at
        parent = a.data._parent
        data = np.array(parent.data_ro_with_halos, copy=True)
        view = int(np.ravel_multi_index(a.data.index, parent.dim))
    else:
        data = np.array(a.data.data_ro_with_halos, copy=True)
    m = a.map_
    perm = mv = off = quot = None
    if m is not None:
        if isinstance(m, op2.PermutedMap):
            perm = list(m.permutation)
        mv = m._base().values_with_halo
        off = m.offset if iterset._extruded else None
        quot = m.offset_quotient if iterset._extruded else None
    return oracle.ODat(data, int(access), mv, offset=off, perm=perm, offset_quotient=quot, view_index=view), data


def _omat(mat, maps, lgmaps, access, iterset, unroll=False):
    csr = oracle_pattern(mat.sparsity)
    rm, cm = maps
    lg = lgmaps or (None, None)
    ext = iterset._extruded
    return oracle.OMat(csr, int(access), rm._base().values_with_halo, cm._base().values_with_halo,
                       roffset=rm.offset if ext else None, coffset=cm.offset if ext else None,
                       row_lgmap=None if lg[0] is None else np.ascontiguousarray(lg[0], dtype=np.int32),
                       col_lgmap=None if lg[1] is None else np.ascontiguousarray(lg[1], dtype=np.int32),
                       unroll=unroll,
                       roffset_quotient=rm.offset_quotient if ext else None,
                       coffset_quotient=cm.offset_quotient if ext else None,
                       rperm=list(rm.permutation) if isinstance(rm, op2.PermutedMap) else None,
                       cperm=list(cm.permutation) if isinstance(cm, op2.PermutedMap) else None), csr


def oracle_run(kernel, iterset, *args, iteration_region=None, pass_layer_arg=False):
    """Execute ``op2.par_loop(kernel, iterset, *args)`` semantics on the CPU oracle using COPIES of the
    carriers' host data.  Returns a list with the post-loop array (or OracleCSR) per argument; mixed arguments
    give a list of arrays (MixedDat) or a list of lists of OracleCSRs (MixedMat)."""
    from firedrake_amd import parloop as _pl
    oargs, outs = [], []
    for a in args:
        if isinstance(a, getattr(_pl, "MixedDatLegacyArg", ())):
            parts = [_odat(p, a.access, iterset) for p in a.split]
            oargs.append(oracle.OMixedDat([p for p, _ in parts], int(a.access)))
            outs.append([d for _, d in parts])
        elif isinstance(a, getattr(_pl, "MixedMatLegacyArg", ())):
            blocks = [[_omat(b.data, b.maps, b.lgmaps, a.access, iterset) for b in row] for row in a.split]
            oargs.append(oracle.OMixedMat([[m for m, _ in row] for row in blocks], int(a.access)))
            outs.append([[c for _, c in row] for row in blocks])
        elif isinstance(a, DatLegacyArg):
            od, data = _odat(a, a.access, iterset)
            oargs.append(od)
            outs.append(data)
        elif isinstance(a, GlobalLegacyArg):
            data = np.array(a.data.data_ro, copy=True)
            oargs.append(oracle.OGlobal(data, int(a.access)))
            outs.append(data)
        elif isinstance(a, MatLegacyArg):
            om, csr = _omat(a.data, a.maps, a.lgmaps, a.access, iterset, unroll=getattr(a, "unroll_map", False))
            oargs.append(om)
            outs.append(csr)
    subset = iterset.indices if isinstance(iterset, op2.Subset) else None
    layers = None
    if iterset._extruded:
        la = iterset.layers_array
        layers = tuple(int(x) for x in la[0]) if iterset.constant_layers else np.asarray(la, dtype=np.int32)
    oracle.par_loop(kernel.code, kernel.name, 0, iterset.size, oargs, subset=subset, layers=layers,
                    iteration_region=_REGION[iteration_region], pass_layer_arg=pass_layer_arg,
                    periodic=bool(iterset._extruded and iterset._extruded_periodic),
                    init_with_zero=bool(getattr(kernel, "requires_zeroed_output_arguments", False)))
    return outs


def plan_ref(mapv, start, end, epb):
    """numpy restatement of the block-localisation plan (include/fdhip.h: fd_plan_create): per block of ``epb``
    entities the sorted distinct nodes and, per entity, the positions of its nodes in that list."""
    blk, lst, lm = [0], [], np.zeros((end - start, mapv.shape[1]), dtype=np.uint16)
    for b0 in range(start, end, epb):
        b1 = min(end, b0 + epb)
        u, inv = np.unique(mapv[b0:b1].reshape(-1), return_inverse=True)
        lst.append(u)
        lm[b0 - start:b1 - start] = inv.reshape(b1 - b0, -1)
        blk.append(blk[-1] + len(u))
    return np.array(blk, np.int32), np.concatenate(lst).astype(np.int32) if lst else np.zeros(0, np.int32), lm


def plan_ref_blocks(mapv, blocks):
    """plan_ref over explicit block boundaries ``blocks`` (slot offsets, first = 0, last = len(mapv))."""
    blk, lst, lm = [0], [], np.zeros((len(mapv), mapv.shape[1]), dtype=np.uint16)
    for b in range(len(blocks) - 1):
        b0, b1 = int(blocks[b]), int(blocks[b + 1])
        if b1 == b0:                                          # a block without instances (chained plans have them)
            blk.append(blk[-1])
            continue
        u, inv = np.unique(mapv[b0:b1].reshape(-1), return_inverse=True)
        lst.append(u)
        lm[b0:b1] = inv.reshape(b1 - b0, -1)
        blk.append(blk[-1] + len(u))
    return (np.array(blk, np.int32), np.concatenate(lst).astype(np.int32) if lst else np.zeros(0, np.int32), lm)


def first_touch_ref(mapv, order, nnodes):
    """numpy restatement of fd_first_touch_order: rows [0, nnodes) sorted by (rank of the first entity of ``order`` touching
    them, node id); untouched rows last.  Returns (plist, pinv)."""
    rank = np.full(nnodes, np.iinfo(np.uint32).max, dtype=np.uint64)
    rows = np.asarray(mapv)[np.asarray(order)]
    for r in range(len(rows) - 1, -1, -1):
        for g in rows[r]:
            if 0 <= g < nnodes:
                rank[g] = r
    plist = np.lexsort((np.arange(nnodes), rank)).astype(np.int32)
    pinv = np.empty(nnodes, dtype=np.int32)
    pinv[plist] = np.arange(nnodes, dtype=np.int32)
    return plist, pinv


def ocr_plan_ref(rmapv, cmapv, nent, row_blocks, rowptr, colidx, pinv=None):
    """numpy restatement of the owner-computes-rows plan (include/fdhip.h: fd_ocrplan_create in natural order +
    fd_csr_elem_row_offsets): per block of row nodes the entities that touch one of its rows (*instances*, in entity
    order), and per instance the position of every (i, j) entry inside its CSR row."""
    inst_off, inst_ent = [0], []
    rows_of = rmapv[:nent]
    if pinv is not None:                 # blocks are ranges of row POSITIONS (fd_ocrplan_create_ordered)
        rows_of = np.where((rows_of >= 0) & (rows_of < len(pinv)), np.asarray(pinv)[np.clip(rows_of, 0, len(pinv) - 1)], -1)
    for b in range(len(row_blocks) - 1):
        hit = ((rows_of >= row_blocks[b]) & (rows_of < row_blocks[b + 1])).any(axis=1)
        inst_ent.append(np.nonzero(hit)[0])
        inst_off.append(inst_off[-1] + int(hit.sum()))
    inst_ent = np.concatenate(inst_ent).astype(np.int32) if inst_ent else np.zeros(0, np.int32)
    ar, ac = rmapv.shape[1], cmapv.shape[1]
    kidx = np.zeros((len(inst_ent), ar * ac), dtype=np.uint8)
    for s_, e in enumerate(inst_ent):
        for i in range(ar):
            r = rmapv[e, i]
            row = colidx[rowptr[r]:rowptr[r + 1]]
            for j in range(ac):
                pos = int(np.searchsorted(row, cmapv[e, j]))
                assert pos < len(row) and row[pos] == cmapv[e, j] and pos < 255
                kidx[s_, i * ac + j] = pos
    return np.array(inst_off, np.int32), inst_ent, kidx


def ocrs_pair_counts_ref(rmapv, start, end, row_blocks, pinv=None):
    """numpy restatement of fd_ocrplan_pair_counts: cnt[a, b] (a < b) = entities whose local rows a and b fall into one row block."""
    rows = np.asarray(rmapv)[start:end]
    pos = rows
    if pinv is not None:
        pos = np.where((rows >= 0) & (rows < len(pinv)), np.asarray(pinv)[np.clip(rows, 0, len(pinv) - 1)], -1)
    rb = np.asarray(row_blocks)
    blk = np.where((pos >= rb[0]) & (pos < rb[-1]), np.searchsorted(rb, pos, side="right") - 1, -1)
    ar = rows.shape[1]
    cnt = np.zeros((ar, ar), dtype=np.int64)
    for a in range(ar):
        for b in range(a + 1, ar):
            cnt[a, b] = int(((blk[:, a] >= 0) & (blk[:, a] == blk[:, b])).sum())
    return cnt


def choose_groups_ref(cnt):
    """numpy restatement of SlicedOcrPlan.choose_groups: greedy matching on the co-ownership counts, ties to the lower rows."""
    ar = cnt.shape[0]
    pairs = sorted(((int(cnt[a, b]), -a, -b) for a in range(ar) for b in range(a + 1, ar)), reverse=True)
    free, groups = set(range(ar)), []
    for _, na, nb in pairs:
        if -na in free and -nb in free:
            groups.append((-na, -nb))
            free -= {-na, -nb}
    return tuple(sorted(groups + [(a, None) for a in sorted(free)]))


def ocrs_paired_plan_ref(rmapv, cmapv, start, end, row_blocks, rowptr, colidx, acc_by_node, acc_by_pos, groups, pinv=None, rlg=None, clg=None,
                         interleave=0):
    """numpy restatement of the PAIRED row-sliced plan (include/fdhip.h: fd_ocrplan_create_paired + fd_ocrplan_sliced_tables): instances
    (entity, group of one or two local rows) for every group with a row inside the row block, per block grouped by group (entity
    order inside a group), every group padded to a multiple of 64 slots with copies of its last entity; two table rows per instance --
    a row of the group that is absent, dropped, or owned by ANOTHER block reads slot 0xffff.  Returns (padded inst_off, inst_ent,
    chunk_role = group index, valid, slot (n, 2), kk (n, 2, ac))."""
    ar, ac = rmapv.shape[1], cmapv.shape[1]
    rows = np.asarray(rmapv)[start:end]
    pos = rows
    if pinv is not None:
        pos = np.where((rows >= 0) & (rows < len(pinv)), np.asarray(pinv)[np.clip(rows, 0, len(pinv) - 1)], -1)
    inst_off, ent, role, valid = [0], [], [], []
    for b in range(len(row_blocks) - 1):
        n = 0
        inb = (pos >= row_blocks[b]) & (pos < row_blocks[b + 1])
        for g, (ra, rb_) in enumerate(groups):
            ina = inb[:, ra]
            inb2 = np.zeros_like(ina) if rb_ is None else inb[:, rb_]
            # sorted by ownership class (both rows / the first only / the second only), entity order inside a class; the stride
            # permutation stays inside a class
            parts = []
            for sel in (ina & inb2, ina & ~inb2, ~ina & inb2):
                ec = start + np.nonzero(sel)[0]
                if interleave > 1 and len(ec) > 1:
                    P = next(q for q in range(interleave, interleave + len(ec) + 2) if np.gcd(q, len(ec)) == 1)
                    ec = ec[(np.arange(len(ec)) * P) % len(ec)]
                parts.append((ec, start + np.nonzero(sel)[0]))
            es = np.concatenate([pp[0] for pp in parts])
            if len(es) == 0:
                continue
            last = [pp[1][-1] for pp in parts if len(pp[1])][-1]       # the padding repeats the last entity of the sorted segment
            pad = -len(es) % 64
            ent += [es, np.full(pad, last)]
            valid += [np.ones(len(es), np.uint8), np.zeros(pad, np.uint8)]
            role += [np.full((len(es) + pad) // 64, g, np.uint8)]
            n += len(es) + pad
        inst_off.append(inst_off[-1] + n)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    ent, role, valid = cat(ent, np.int32), cat(role, np.uint8), cat(valid, np.uint8)
    slot = np.full((len(ent), 2), 0xffff, dtype=np.uint16)
    kk = np.full((len(ent), 2, ac), 0xff, dtype=np.uint8)
    blk = np.searchsorted(np.asarray(inst_off), np.arange(len(ent)), side="right") - 1
    for t, e in enumerate(ent):
        for s_, lr in enumerate(groups[role[t // 64]]):
            if lr is None or not valid[t]:
                continue
            r = rmapv[e, lr]
            if r < 0 or (rlg is not None and rlg[r] < 0):
                continue
            p_ = r if pinv is None else (pinv[r] if r < len(pinv) else -1)
            if not (row_blocks[blk[t]] <= p_ < row_blocks[blk[t] + 1]):
                continue
            slot[t, s_] = acc_by_node[r] - acc_by_pos[row_blocks[blk[t]]]
            row = colidx[rowptr[r]:rowptr[r + 1]]
            for j in range(ac):
                c = cmapv[e, j]
                if c < 0 or (clg is not None and clg[c] < 0):
                    continue
                q = int(np.searchsorted(row, c))
                assert q < len(row) and row[q] == c and q < 255
                kk[t, s_, j] = q
    return np.array(inst_off, np.int32), ent, role, valid, slot, kk


def ocrs_plan_ref(rmapv, cmapv, start, end, row_blocks, rowptr, colidx, acc_by_node, acc_by_pos, pinv=None, rlg=None, clg=None,
                  interleave=0, per_dof=None):
    """numpy restatement of the row-sliced plan (include/fdhip.h: fd_ocrplan_create_sliced + fd_ocrplan_sliced_tables):
    instances (entity, local row i) for every row-map entry inside a row block, per block grouped by i (entity order inside
    a group), every group padded to a multiple of 64 slots with copies of its last entity.  Returns (padded inst_off,
    inst_ent, chunk_role, valid, slot, kk, rowlen).  ``per_dof=(rbs, cbs)``: the lgmaps are per DOF (``unroll``); the result
    gains the row masks (uint8) and column masks (uint64)."""
    ar, ac = rmapv.shape[1], cmapv.shape[1]
    rows = np.asarray(rmapv)[start:end]
    pos = rows
    if pinv is not None:
        pos = np.where((rows >= 0) & (rows < len(pinv)), np.asarray(pinv)[np.clip(rows, 0, len(pinv) - 1)], -1)
    inst_off, ent, role, valid = [0], [], [], []
    for b in range(len(row_blocks) - 1):
        n = 0
        for i in range(ar):
            es = start + np.nonzero((pos[:, i] >= row_blocks[b]) & (pos[:, i] < row_blocks[b + 1]))[0]
            if len(es) == 0:
                continue
            last = es[-1]                                   # the padding repeats the last entity of the group in entity order
            if interleave > 1 and len(es) > 1:
                P = next(q for q in range(interleave, interleave + len(es) + 2) if np.gcd(q, len(es)) == 1)
                es = es[(np.arange(len(es)) * P) % len(es)]
            pad = -len(es) % 64
            ent += [es, np.full(pad, last)]
            valid += [np.ones(len(es), np.uint8), np.zeros(pad, np.uint8)]
            role += [np.full((len(es) + pad) // 64, i, np.uint8)]
            n += len(es) + pad
        inst_off.append(inst_off[-1] + n)
    cat = lambda xs, dt: np.concatenate(xs).astype(dt) if xs else np.zeros(0, dt)
    ent, role, valid = cat(ent, np.int32), cat(role, np.uint8), cat(valid, np.uint8)
    slot = np.full(len(ent), 0xffff, dtype=np.uint16)
    kk = np.full((len(ent), ac), 0xff, dtype=np.uint8)
    rowlen = np.zeros(len(ent), dtype=np.uint16)
    rmask, cmask = np.zeros(len(ent), dtype=np.uint8), np.zeros(len(ent), dtype=np.uint64)
    blk = np.searchsorted(np.asarray(inst_off), np.arange(len(ent)), side="right") - 1
    for t, e in enumerate(ent):
        r = rmapv[e, role[t // 64]]
        if r >= 0:
            rowlen[t] = rowptr[r + 1] - rowptr[r]
        if per_dof is not None:
            rbs, cbs = per_dof
            rm = sum(1 << p for p in range(rbs) if r >= 0 and (rlg is None or rlg[r * rbs + p] >= 0))
            if not valid[t] or r < 0 or rm == 0:
                continue
            rmask[t] = rm
            cmask[t] = np.uint64(sum(1 << (j * cbs + q) for j in range(ac) for q in range(cbs)
                                     if cmapv[e, j] >= 0 and (clg is None or clg[cmapv[e, j] * cbs + q] >= 0)))
        elif not valid[t] or r < 0 or (rlg is not None and rlg[r] < 0):
            continue
        slot[t] = acc_by_node[r] - acc_by_pos[row_blocks[blk[t]]]
        row = colidx[rowptr[r]:rowptr[r + 1]]
        for j in range(ac):
            c = cmapv[e, j]
            if c < 0 or (per_dof is None and clg is not None and clg[c] < 0):
                continue
            q = int(np.searchsorted(row, c))
            assert q < len(row) and row[q] == c and q < 255
            kk[t, j] = q
    if per_dof is not None:
        return np.array(inst_off, np.int32), ent, role, valid, slot, kk, rowlen, rmask, cmask
    return np.array(inst_off, np.int32), ent, role, valid, slot, kk, rowlen


def lane_slot_to_entity(n, T):
    """numpy restatement of the lane order (include/fdhip.h: fd_plan_set_lane_order): slot k*T + t holds the k-th
    entity of lane t's contiguous run; the first n % T runs are one longer."""
    q, rem = divmod(n, T)
    ent = np.full(n, -1, dtype=np.int64)
    for t in range(T):
        cnt = q + 1 if t < rem else q
        first = t * q + min(t, rem)
        for k in range(cnt):
            ent[k * T + t] = first + k
    assert (ent >= 0).all() and len(set(ent.tolist())) == n
    return ent


def structured_tri_mesh(nx, ny, seed=0, perturb=0.0):
    """P1 triangles on a grid: (coords (nv,2), cells (nc,3) int32)."""
    xs, ys = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="xy")
    coords = np.stack([xs.ravel(), ys.ravel()], axis=1)
    if perturb:
        rng = np.random.default_rng(seed)
        interior = ((coords > 1e-12) & (coords < 1 - 1e-12)).all(axis=1)      # keep |Omega| = 1
        coords[interior] += perturb * rng.uniform(-1, 1, size=(interior.sum(), 2)) / max(nx, ny)
    cells = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            b, c, d = a + 1, a + nx + 1, a + nx + 2
            cells += [(a, b, c), (b, d, c)]
    return coords, np.asarray(cells, dtype=np.int32)


def kd_order_ref(pts, leaf_size, base=0):
    """numpy restatement of fd_kd_order: k-d partition of the points into ceil(n / leaf_size) leaves of equal population --
    every segment holding more than one leaf is sorted (stably) along the longest axis of its bounding box, position
    quantised to 32 bits, and cut at the boundary between its two groups of leaves.  Returns (order, leaf starts)."""
    pts = np.asarray(pts, dtype=np.float64)
    n, pdim = pts.shape
    if n == 0:
        return np.zeros(0, np.int32), np.zeros(1, np.int64)
    idx = np.arange(n)
    segs = [(0, n, -(-n // leaf_size))]
    while any(l > 1 for _, _, l in segs):
        nxt, keys = [], np.zeros(n, dtype=np.uint64)
        for s, (st, c, l) in enumerate(segs):
            sl = slice(st, st + c)
            keys[sl] = np.uint64(s) << np.uint64(32)
            if l <= 1:
                nxt.append((st, c, l))
                continue
            if c:
                P = pts[idx[sl]]
                lo, hi = P.min(axis=0), P.max(axis=0)
                w = hi - lo
                ax = int(np.argmax(w))                      # first of the longest axes, like the device loop
                if w[ax] > 0:
                    u = np.clip((P[:, ax] - lo[ax]) / w[ax], 0.0, 1.0)
                    keys[sl] |= (u * 4294967295.0).astype(np.uint64)
            ll = l // 2
            cl = (c * ll + l // 2) // l
            nxt += [(st, cl, ll), (st + cl, c - cl, l - ll)]
        idx = idx[np.argsort(keys, kind="stable")]
        segs = nxt
    starts = np.array([st for st, _, _ in segs] + [n], dtype=np.int64)
    for a, b in zip(starts[:-1], starts[1:]):             # inside a leaf: index order
        idx[a:b] = np.sort(idx[a:b])
    return (base + idx).astype(np.int32), starts


def cut_blocks_ref(counts, target):
    """Parloop.LocalityOrder.cut restated: empty leaves dropped, leaves above 1.5 x target split into equal parts."""
    sizes = []
    for c in [int(c) for c in counts if c > 0]:
        if c > (3 * target) // 2:
            parts = -(-c // target)
            q, r = divmod(c, parts)
            sizes += [q + 1] * r + [q] * (parts - r)
        else:
            sizes.append(c)
    return np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)


def locality_order_ref(mapv, start, end, pos, target=1400):
    """numpy restatement of the entity order of un-hinted loops (Parloop._locality_order): the nodes of the position field in
    k-d leaves (fd_kd_order) sized for ~``target`` entities around each, every entity in the group of the lowest leaf among its
    nodes, groups in leaf order and entities in their own order inside a group (fd_group_entities).  Returns (order, block
    boundaries)."""
    rows = np.asarray(mapv[start:end])
    n, nnodes = len(rows), len(pos)
    leaf_nodes = max(int(round(target * nnodes / max(n, 1))), 1)
    norder, nstarts = kd_order_ref(pos, leaf_nodes)
    label = np.empty(nnodes, dtype=np.int64)
    label[norder] = np.repeat(np.arange(len(nstarts) - 1), np.diff(nstarts))
    ok = (rows >= 0) & (rows < nnodes)
    key = np.where(ok, label[np.clip(rows, 0, nnodes - 1)], len(nstarts) - 2).min(axis=1)
    order = (start + np.argsort(key, kind="stable")).astype(np.int32)
    return order, cut_blocks_ref(np.bincount(key, minlength=len(nstarts) - 1), target)
