"""tools/lds_sim.py -- offline model of the LDS traffic of the owner-computes-rows P1 Jacobian wrapper.

The kernel is bound by the LDS pipe and a third of its LDS cycles are bank conflicts (profiles/r3z_pmc_summary.txt).  This
script rebuilds, with numpy on the host, the plan the device builds for a hinted (tiled) cube mesh -- row blocks, instances,
local node ids, row offsets, the stencil order (fd_plan.hip: ocr_stencil_keys) and the bank-aware packing (ocr_pack_k) -- and
counts, per 16-lane conflict window (tools/microbench_lds.hip: 64-bit LDS operations are processed 16 lanes at a time;
ds_read_b64 resolves 32 eight-byte banks with equal addresses broadcast, ds_add_f64 only 16 and equal addresses serialise),
the passes every LDS instruction of the main loop needs.  It is a design aid for the instance scheduler: schedules can be
compared here before any of them is written as a device kernel.

    python tools/lds_sim.py [n] [variant ...]        n = cubes per axis (default 24); variants: natural stencil packed ...
    python tools/lds_sim.py n layouts                 accumulator-row layouts other than the CSR order (profiles/r3w_lds_sim.txt)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

NTHR = 512          # lanes per workgroup of the whole-entity owner-computes-rows wrapper
WINDOW = 16
CHUNK = 128


def build_plan(n):
    """Row blocks, instances (block, cell) in natural order and per-instance tables of the hinted C2 mesh."""
    from firedrake_amd import mesh as fmesh
    import scipy.sparse as sp
    m = fmesh.UnitCubeMesh(n, numbering="tiled")
    sd = m.spaces[1]
    cm = sd.cell_node_map
    rmap = np.asarray(cm.values_with_halo, dtype=np.int64)
    nn = int(rmap.max()) + 1
    ncell = rmap.shape[0]
    rb = np.asarray(cm.preferred_node_blocks, dtype=np.int64)
    # CSR pattern (sorted columns)
    r = np.repeat(rmap, 4, axis=1).reshape(-1)
    c = np.tile(rmap, (1, 4)).reshape(-1)
    A = sp.coo_matrix((np.ones(len(r), dtype=np.int8), (r, c)), shape=(nn, nn)).tocsr()
    A.sum_duplicates()
    A.sort_indices()
    rowptr, colidx = A.indptr.astype(np.int64), A.indices.astype(np.int64)
    keys_csr = np.repeat(np.arange(nn), np.diff(rowptr)) * nn + colidx
    # instances: (block, cell) for every block owning one of the cell's rows
    nblk = np.searchsorted(rb, rmap, side="right") - 1                   # block of every node of every cell
    pairs = np.unique(np.stack([nblk.reshape(-1), np.repeat(np.arange(ncell), 4)], axis=1), axis=0)
    iblk, ient = pairs[:, 0], pairs[:, 1]
    rows = rmap[ient]                                                    # (ninst, 4)
    own = nblk[ient] == iblk[:, None]
    # position of (row i, column j) inside row i
    key = (rows[:, :, None] * nn + rows[:, None, :]).reshape(len(ient), 16)
    pos = np.searchsorted(keys_csr, key) - rowptr[rows][:, :, None].repeat(4, axis=2).reshape(len(ient), 16)
    base = rowptr[rows] - rowptr[rb[iblk]][:, None]                      # accumulator offset of the row inside its block
    addr = np.where(np.repeat(own, 4, axis=1), np.repeat(base, 4, axis=1) + pos, -1)      # (ninst, 16) LDS double index or -1
    # local node ids: rank of the node among the distinct nodes of the block's instances
    bn = np.unique(np.stack([np.repeat(iblk, 4), rows.reshape(-1)], axis=1), axis=0)
    bstart = np.searchsorted(bn[:, 0], np.arange(len(rb) - 1))
    kk = np.repeat(iblk, 4) * (nn + 1) + rows.reshape(-1)
    lm = (np.searchsorted(bn[:, 0] * (nn + 1) + bn[:, 1], kk) - bstart[np.repeat(iblk, 4)]).reshape(len(ient), 4)
    return {"iblk": iblk, "ient": ient, "rows": rows, "own": own, "addr": addr, "lm": lm, "rb": rb, "n0": rb[iblk], "pos": pos,
            "rowlen": np.diff(rowptr), "csr_base": rowptr[:-1] - rowptr[rb[np.searchsorted(rb, np.arange(nn), side="right") - 1]]}


def stencil_keys(p):
    """fd_plan.hip: ocr_stencil_keys (hinted plans: row position = row)."""
    rows, own, n0 = p["rows"], p["own"], p["n0"]
    first = np.where(own, rows, np.iinfo(np.int64).max).copy()
    # FIRST owned row in vertex order (not the smallest)
    idx = np.argmax(own, axis=1)
    first = rows[np.arange(len(rows)), idx]
    h = np.full(len(rows), 2166136261, dtype=np.uint64)
    M = np.uint64(0xffffffff)
    for i in range(4):
        d = ((rows[:, i] - first) & 0xffffffff).astype(np.uint64)
        h = ((h ^ own[:, i].astype(np.uint64)) * np.uint64(16777619)) & M
        h = ((h ^ (d & np.uint64(0xffff))) * np.uint64(16777619)) & M
        h = ((h ^ (d >> np.uint64(16))) * np.uint64(16777619)) & M
    h ^= h >> np.uint64(16)
    mask = (own * (1 << np.arange(4))).sum(axis=1).astype(np.uint64)
    return ((p["iblk"].astype(np.uint64) << np.uint64(40)) | ((mask ^ np.uint64(15)) << np.uint64(32))
            | ((h & np.uint64(0xffff)) << np.uint64(16)) | ((first - n0).astype(np.uint64) & np.uint64(0xffff)))


def block_ranges(iblk):
    cuts = np.nonzero(np.diff(iblk))[0] + 1
    return np.concatenate([[0], cuts]), np.concatenate([cuts, [len(iblk)]])


def pack_greedy(p, order, chunk=CHUNK, window=WINDOW, gather_weight=3):
    """fd_plan.hip: ocr_pack_k restated: per chunk of ``chunk`` consecutive instances of a block, fill the windows one slot at a time with
    the unplaced candidate that adds the fewest bank collisions, ties to the earliest."""
    addr, lm, iblk = p["addr"][order], p["lm"][order], p["iblk"][order]
    out = order.copy()
    b0s, b1s = block_ranges(iblk)
    sig_a = np.where(addr >= 0, addr & 15, 255)          # (n, 16)
    sig_g = lm & 31                                      # (n, 4)
    for b0, b1 in zip(b0s, b1s):
        for o in range(b0, b1, chunk):
            n = min(chunk, b1 - o)
            if n <= window:
                continue
            sa, sg, ga = sig_a[o:o + n], sig_g[o:o + n], lm[o:o + n]
            placed = np.zeros(n, dtype=bool)
            sel = np.empty(n, dtype=np.int64)
            for s in range(n):
                if s % window == 0:
                    amask = np.zeros((16, 16), dtype=bool)       # [entry][bank]
                    gown = np.full((4, 32), -1, dtype=np.int64)
                held = gown[np.arange(4)[None, :], sg]           # (n, 4)
                cost = gather_weight * ((held >= 0) & (held != ga)).sum(axis=1)
                hit = amask[np.arange(16)[None, :], np.minimum(sa, 15)] & (sa != 255)
                cost = cost + hit.sum(axis=1)
                cost[placed] = 1 << 30
                c = int(np.argmin(cost))
                placed[c] = True
                sel[s] = c
                ok = sa[c] != 255
                amask[np.arange(16)[ok], sa[c][ok]] = True
                free = gown[np.arange(4), sg[c]] < 0
                gown[np.arange(4)[free], sg[c][free]] = ga[c][free]
            out[o:o + n] = order[o:o + n][sel]
    return out


def simulate(p, order, nthr=NTHR, window=WINDOW):
    """LDS passes of the main loop under the instance order ``order`` (slot s of a block -> lane s % nthr of trip s // nthr)."""
    addr, lm, own, iblk = p["addr"][order], p["lm"][order], p["own"][order], p["iblk"][order]
    b0s, b1s = block_ranges(iblk)
    slot = np.arange(len(order)) - np.repeat(b0s, b1s - b0s)
    # global window id: (block, trip, window in the workgroup); wave id: (block, trip, wave)
    wpb = nthr // window
    win = (np.repeat(np.arange(len(b0s)), b1s - b0s) * 1024 + slot // nthr) * wpb + (slot % nthr) // window
    wave = (np.repeat(np.arange(len(b0s)), b1s - b0s) * 1024 + slot // nthr) * (nthr // 64) + (slot % nthr) // 64
    _, win = np.unique(win, return_inverse=True)
    _, wave = np.unique(wave, return_inverse=True)
    nwin = win.max() + 1
    res = {}
    # gathers: per vertex i, distinct addresses per bank (32 banks), three coordinate reads each
    gp = gb = 0
    for i in range(4):
        a = lm[:, i]
        u = np.unique(np.stack([win, a], axis=1), axis=0)               # distinct (window, address)
        cnt = np.zeros((nwin, 32), dtype=np.int32)
        np.add.at(cnt, (u[:, 0], u[:, 1] & 31), 1)
        gp += 3 * cnt.max(axis=1).sum()
        gb += 3 * nwin
    res["gather_passes"], res["gather_min"] = int(gp), int(gb)
    # atomics: entry (i, j) issues for a wavefront when one of its lanes owns row i; per window the passes are the largest
    # number of lanes on one of the 16 banks
    ap = ab = ai = 0
    nwave = wave.max() + 1
    for i in range(4):
        wave_on = np.zeros(nwave, dtype=bool)
        wave_on[wave[own[:, i]]] = True
        ai += 4 * int(wave_on.sum())
        for j in range(4):
            a = addr[:, i * 4 + j]
            on = a >= 0
            cnt = np.zeros((nwin, 16), dtype=np.int32)
            np.add.at(cnt, (win[on], a[on] & 15), 1)
            mx = cnt.max(axis=1)
            ap += int(mx.sum())
            ab += int((mx > 0).sum())
    res["atomic_passes"], res["atomic_min"], res["atomic_wave_instructions"] = ap, ab, ai
    res["instances"] = len(order)
    # cycles at the measured conflict-free rates (profiles/r1i_microbench_lds.txt): ds_read_b64 11.8 lanes/clk, ds_add_f64 8.4
    res["model_cycles"] = gp * 16 / 11.8 + ap * 16 / 8.4
    res["model_cycles_min"] = gb * 16 / 11.8 + ab * 16 / 8.4
    return res


def natural(p):
    return np.arange(len(p["iblk"]))


def stencil(p):
    return np.argsort(stencil_keys(p), kind="stable")


VARIANTS = {
    "natural": lambda p: natural(p),
    "stencil": lambda p: stencil(p),
    "packed": lambda p: pack_greedy(p, stencil(p)),
    "packed256": lambda p: pack_greedy(p, stencil(p), chunk=256),
    "packed512": lambda p: pack_greedy(p, stencil(p), chunk=512),
    "packed_g1": lambda p: pack_greedy(p, stencil(p), gather_weight=1),
    "packed_nat": lambda p: pack_greedy(p, natural(p)),
}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    names = sys.argv[2:] or ["natural", "stencil", "packed"]
    t0 = time.time()
    p = build_plan(n)
    print(f"n={n}: {len(p['rb']) - 1} row blocks, {len(p['iblk'])} instances ({time.time() - t0:.1f} s)")
    for name in names:
        t0 = time.time()
        order = VARIANTS[name](p)
        assert np.array_equal(np.sort(order), np.arange(len(order)))
        r = simulate(p, order)
        print(f"{name:12s} gathers {r['gather_passes'] / r['gather_min']:.3f}x of {r['gather_min']}  atomics {r['atomic_passes'] / r['atomic_min']:.3f}x of "
              f"{r['atomic_min']}  wave-atomics/instance-wave {r['atomic_wave_instructions'] / (r['instances'] / 64):.2f}  "
              f"model cycles {r['model_cycles'] / 1e6:.2f} M (floor {r['model_cycles_min'] / 1e6:.2f} M)  [{time.time() - t0:.1f} s]")



# ---- experiments with the LDS LAYOUT of the accumulator rows (possible wherever the flush goes through a per-entry table) -------------
def relayout(p, base_of_row):
    """Plan with the accumulator row of node r at LDS offset base_of_row[r] (relative to its block) instead of its CSR offset."""
    q = dict(p)
    rows, own, pos = p["rows"], p["own"], p["pos"]
    q["addr"] = np.where(np.repeat(own, 4, axis=1), np.repeat(base_of_row[rows], 4, axis=1) + pos, -1)
    return q


def greedy_residues(p, order, nres=16, window=WINDOW, nthr=NTHR):
    """Algorithm G: walk the schedule ``order``; give every accumulator row, when it is first met, the residue (LDS offset mod 16) that
    collides least with the rows already placed in the windows it takes part in; then lay the rows of a block out in order of first
    appearance, each at the next offset with its residue (padding < 16 doubles).  Returns base_of_row and the padded size per block."""
    rows, own, pos, iblk = p["rows"][order], p["own"][order], p["pos"][order], p["iblk"][order]
    nn = int(p["rows"].max()) + 1
    b0s, b1s = block_ranges(iblk)
    res = np.full(nn, -1, dtype=np.int64)
    base = np.zeros(nn, dtype=np.int64)
    rowlen = p["rowlen"]
    sizes = []
    for b0, b1 in zip(b0s, b1s):
        n = b1 - b0
        r_, o_, p_ = rows[b0:b1], own[b0:b1], pos[b0:b1].reshape(n, 4, 4)
        slot = np.arange(n)
        win = (slot // nthr) * (nthr // window) + (slot % nthr) // window
        # accesses of every owned row: (window, entry q = i*4+j, column position)
        acc = {}
        for s in range(n):
            for i in range(4):
                if o_[s, i]:
                    acc.setdefault(int(r_[s, i]), []).append((win[s], i, p_[s, i]))
        used = {}                                            # (window, entry) -> set of banks taken
        seq = []
        for s in range(n):
            for i in range(4):
                r = int(r_[s, i])
                if o_[s, i] and res[r] < 0:
                    cost = np.zeros(nres, dtype=np.int64)
                    for (w, ii, cp) in acc[r]:
                        for j in range(4):
                            taken = used.get((w, ii * 4 + j))
                            if taken:
                                for rho in range(nres):
                                    if (rho + cp[j]) % nres in taken:
                                        cost[rho] += 1
                    # prefer the residue reachable with the least padding among the cheapest
                    cur = sum(x[1] for x in seq) if seq else 0
                    pad = (np.arange(nres) - cur) % nres
                    rho = int(np.lexsort((pad, cost))[0])
                    res[r] = rho
                    seq.append((r, rowlen[r] + int(pad[rho])))
                    base[r] = cur + int(pad[rho])
                    for (w, ii, cp) in acc[r]:
                        for j in range(4):
                            used.setdefault((w, ii * 4 + j), set()).add((rho + cp[j]) % nres)
        sizes.append(sum(x[1] for x in seq))
    return base, np.array(sizes)


def layout_experiments(p):
    """The layouts of profiles/r3w_lds_sim.txt on the hinted plan (interior tiles are 8 x 8 x 4 nodes in lexicographic order):
    the rows every kind of fully owned cell can anchor at first; the exact linear residue -(x + 7 y + 49 z) mod 16 reached by
    padding; the greedy residue assignment."""
    rb, nn = p["rb"], len(p["rowlen"])
    blk = np.searchsorted(rb, np.arange(nn), side="right") - 1
    loc = np.arange(nn) - rb[blk]
    x, y, z = loc % 8, (loc // 8) % 8, loc // 64
    full = (rb[blk + 1] - rb[blk]) == 256
    far = ((x == 7) | (y == 7) | (z == 3)) & full
    order = stencil(p)

    def show(name, q):
        for sched, o in (("stencil", order), ("packed", pack_greedy(q, order))):
            r = simulate(q, o)
            print(f"{name:28s} {sched:8s} atomics {r['atomic_passes'] / r['atomic_min']:.3f}x  gathers {r['gather_passes'] / r['gather_min']:.3f}x  "
                  f"model cycles {r['model_cycles'] / 1e6:.3f} M")

    show("CSR layout", p)
    base = np.zeros(nn, dtype=np.int64)
    lin = np.zeros(nn, dtype=np.int64)
    padded = unpadded = 0
    for b in range(len(rb) - 1):
        r = np.arange(rb[b], rb[b + 1])
        o = r[np.argsort(far[r].astype(int), kind="stable")]
        base[o] = np.concatenate([[0], np.cumsum(p["rowlen"][o])[:-1]])
        if not full[r[0]]:
            lin[o] = base[o]
            continue
        want = (-(x[o] + 7 * y[o] + 49 * z[o])) % 16
        cur = 0
        for k, node in enumerate(o):
            cur += (want[k] - cur) % 16
            lin[node] = cur
            cur += p["rowlen"][node]
        padded += cur
        unpadded += p["rowlen"][o].sum()
    show("anchor rows first", relayout(p, base))
    print(f"(linear residues: {padded / max(unpadded, 1):.3f}x the LDS of the full tiles)")
    show("linear residues, padded", relayout(p, lin))
    gbase, sizes = greedy_residues(p, order)
    print(f"(greedy residues: {sizes.sum() / p['rowlen'][np.unique(p['rows'][p['own']])].sum():.3f}x the LDS)")
    show("greedy residues, padded", relayout(p, gbase))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[2] == "layouts":
        layout_experiments(build_plan(int(sys.argv[1])))
    else:
        main()
