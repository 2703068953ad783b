"""Synthetic structured meshes + function-space data in the layouts Firedrake produces.

The reference obtains cell->node maps, core|owned|ghost entity classes and halo lists from
PETSc DMPlex through firedrake/cython/dmcommon.pyx (get_cell_nodes :1481-1597,
mark_entity_classes :2236-2323, plex_renumbering :2599-2729, create_halo_exchange_sf
:3891-3964).  DMPlex is out of scope; for the benchmark configurations this
module generates the *same layouts* directly:

  * ``UnitSquareMesh`` triangles (utility_meshes.py:640-700, 802), ``UnitCubeMesh`` 6-tet Kuhn
    split per cube (utility_meshes.py:1466-1495) -- same vertex-offset tables;
  * entities ordered core | owned | ghost (pyop2/types/set.py:32-55), only [0,size) executed;
  * nodes numbered in cell-traversal order inside each class (dmcommon.pyx:2688-2712) -- the
    traversal here walks the grid tile by tile so consecutive cells share nodes in all three
    directions (the reference gets its locality from an RCM cell order, mesh.py:1214-1228);
  * CG1 / CG2 scalar and vector spaces that share one Map object (functionspacedata.py:497-520);
  * z-slab partition across ranks with send/receive node lists per neighbour for the halo
    (firedrake/halo.py:87-172).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import numpy as np

from . import op2

# vertex offsets (di, dj, dk) of cube corners v0..v7 and the Kuhn split (utility_meshes.py:1477-1495)
_CORNER = np.array([(0, 0, 0), (1, 0, 0), (0, 1, 0), (1, 1, 0), (0, 0, 1), (1, 0, 1), (0, 1, 1), (1, 1, 1)], dtype=np.int32)
_TETS = np.array([(0, 1, 3, 7), (0, 1, 7, 5), (0, 5, 7, 4), (0, 3, 2, 7), (0, 6, 4, 7), (0, 2, 6, 7)], dtype=np.int32)
# UFC/FIAT edge -> vertex pairs of the reference tetrahedron (entity_dofs order of CG2)
_TET_EDGES = np.array([(2, 3), (1, 3), (1, 2), (0, 3), (0, 2), (0, 1)], dtype=np.int32)
_TRI_EDGES = np.array([(1, 2), (0, 2), (0, 1)], dtype=np.int32)


@dataclass
class HaloLists:
    """Per-neighbour node index lists (local numbering).  ``send[r]`` are owned nodes that rank r
    holds as ghosts, ``recv[r]`` are our ghosts owned by r, in matching order."""
    rank: int = 0
    nranks: int = 1
    send: Dict[int, np.ndarray] = field(default_factory=dict)
    recv: Dict[int, np.ndarray] = field(default_factory=dict)


@dataclass
class FunctionSpaceData:
    """What firedrake/functionspacedata.py caches per (mesh, element): node Set, cell-node Map."""
    degree: int
    node_set: op2.Set
    cell_node_map: op2.Map
    node_points: np.ndarray          # (total_nodes, gdim) physical location of every node
    halo: Optional[HaloLists] = None
    boundary_nodes: Optional[np.ndarray] = None     # local indices of nodes on the domain boundary
    global_dofs: int = 0

    def dat(self, dim=1, data=None, name=None):
        return op2.Dat(self.node_set ** dim if dim != 1 else self.node_set, data, np.float64, name)


@dataclass
class Mesh:
    gdim: int
    cell_set: op2.Set
    coordinates: op2.Dat             # vector CG1 Dat
    coord_space: FunctionSpaceData
    spaces: Dict[int, FunctionSpaceData]
    ncells_global: int
    shape: Tuple[int, ...]
    partition: Tuple[int, ...] = (1, 1, 1)

    def space(self, degree):
        return self.spaces[degree]


def _tile_keys(ix, iy, iz, nx, ny, nz, tile):
    """Sort key that walks a (nx,ny,nz) grid tile by tile, x fastest inside a tile."""
    tx, ty, tz = tile
    Tx, Ty = -(-nx // tx), -(-ny // ty)
    t = ((iz // tz).astype(np.int64) * Ty + iy // ty) * Tx + ix // tx
    loc = ((iz % tz) * ty + iy % ty) * tx + ix % tx
    return t * (tx * ty * tz) + loc


def _split_blocks(blocks, arity, max_entries=32768):
    """Halve tiles whose map rows exceed the plan builder's per-block capacity."""
    blocks = np.asarray(blocks, dtype=np.int64)
    while True:
        d = np.diff(blocks)
        big = np.nonzero(d * arity > max_entries)[0]
        if len(big) == 0:
            return blocks.astype(np.int32)
        mids = blocks[big] + d[big] // 2
        blocks = np.sort(np.concatenate([blocks, mids]))


NUMBERINGS = ("tiled", "lexicographic", "random")


def partition_grid(nranks, shape=None):
    """Process grid (px, py, pz) of a box partition of the cube.  Default: z-slabs (1, 1, nranks) -- two neighbours per
    rank, one message each way; ``shape="blocks"`` factorises nranks into the most cubic grid (8 -> 2 x 2 x 2: smaller
    surface, up to 26 neighbours).  SURVEY.md 8e names both."""
    if shape is None or shape == "slabs":
        return (1, 1, int(nranks))
    if shape == "blocks":
        best = None
        for px in range(1, nranks + 1):
            if nranks % px:
                continue
            for py in range(1, nranks // px + 1):
                if (nranks // px) % py:
                    continue
                pz = nranks // px // py
                cand = (max(px, py, pz) - min(px, py, pz), px + py + pz, (px, py, pz))
                if px <= py <= pz and (best is None or cand < best):
                    best = cand
        return best[2]
    g = tuple(int(v) for v in shape)
    if len(g) != 3 or g[0] * g[1] * g[2] != nranks:
        raise ValueError(f"partition {shape!r} does not multiply to {nranks} ranks")
    return g


def partition_overheads(n, nranks, partition=None, degree=1):
    """What a box partition of the n^3 cube costs, without building any mesh (same rules as UnitCubeMesh): per rank the cubes it
    owns and the ghost cubes of the low-side layers it computes again for the owner-computes-rows Jacobian, the lattice nodes
    (CG_degree) it sends / receives per exchange and its neighbours.  Returns the worst rank's figures -- the ones the step time
    of a bulk-synchronous assembly follows -- and the totals: ``{"grid", "redundant_cell_fraction" (max over ranks of ghost / owned
    cubes), "redundant_cells_total_fraction", "max_halo_nodes", "max_neighbours"}``."""
    nx, ny, nz = (int(n),) * 3 if np.isscalar(n) else tuple(int(v) for v in n)
    ndim = (nx, ny, nz)
    pgrid = partition_grid(nranks, partition)
    p = int(degree)
    worst, own_tot, ghost_tot, halo_max, neigh_max = 0.0, 0, 0, 0, 0
    for rank in range(nranks):
        rc = (rank % pgrid[0], (rank // pgrid[0]) % pgrid[1], rank // (pgrid[0] * pgrid[1]))
        own = tuple(((ndim[d] * rc[d]) // pgrid[d], (ndim[d] * (rc[d] + 1)) // pgrid[d]) for d in range(3))
        nown = int(np.prod([b - a for a, b in own]))
        nloc = int(np.prod([b - a + (1 if rc[d] > 0 else 0) for d, (a, b) in enumerate(own)]))
        own_tot, ghost_tot = own_tot + nown, ghost_tot + (nloc - nown)
        worst = max(worst, (nloc - nown) / nown)
        # nodes: owned box [p lo, p hi) (+ the domain's last plane on the last rank of an axis), local box = one more plane on the
        # high side and the ghost layer's planes on the low side
        oown = [p * (b - a) + (1 if rc[d] == pgrid[d] - 1 else 0) for d, (a, b) in enumerate(own)]
        oloc = [p * (b - a + (1 if rc[d] > 0 else 0)) + 1 for d, (a, b) in enumerate(own)]
        halo_max = max(halo_max, int(np.prod(oloc)) - int(np.prod(oown)))
        neigh = 1
        for d in range(3):
            neigh *= 1 + (1 if rc[d] > 0 else 0) + (1 if rc[d] < pgrid[d] - 1 else 0)
        neigh_max = max(neigh_max, neigh - 1)
    return {"grid": pgrid, "redundant_cell_fraction": worst, "redundant_cells_total_fraction": ghost_tot / own_tot,
            "max_halo_nodes": halo_max, "max_neighbours": neigh_max}


def UnitCubeMesh(n, degrees=(1,), tile=(8, 8, 4), rank=0, nranks=1, perturb=0.0, ghost_cells=True, numbering="tiled",
                 seed=0, partition=None):
    """Kuhn-split tetrahedral unit cube, box-partitioned over ``nranks`` ranks (``partition``: "slabs" = z-slabs, the
    default; "blocks"; or an explicit (px, py, pz) grid).  ``tile`` = cubes per traversal tile.

    ``numbering`` (SURVEY.md 8d asks for a locality-dependence variant of every measurement):
      * ``"tiled"``          cells walk the grid tile by tile, nodes are numbered tile by tile, and the Maps carry the
                             tile boundaries as producer hints (``preferred_blocks`` / ``preferred_node_blocks``);
      * ``"lexicographic"``  cells in plain x-fastest order inside each class, nodes numbered in order of first
                             appearance while walking the cells' closures -- the rule of dmcommon.pyx:2688-2712 applied
                             to an un-tiled cell order -- and NO hints: what a DMPlex-produced mesh looks like to the backend;
      * ``"random"``         cells and nodes randomly permuted inside each class (seeded), no hints: the worst case.

    Partition layout (firedrake/mesh.py:1131-1179 marks the same classes on a DMPlex): a rank owns the cubes of its box
    and the lattice nodes on [lo, hi) of every axis (the domain's last plane goes to the last rank of the axis); cubes
    touching a plane owned by the next rank are "owned" (non-core); with ``ghost_cells`` the local mesh also carries one
    ghost cube layer on the LOW side of every partitioned axis -- the cells that touch the lowest owned node planes --
    so every rank can assemble complete matrix rows for the nodes it owns (owner-computes-rows, SURVEY.md 8e option 1)
    instead of shipping off-process rows the way MatAssemblyBegin/End does (pyop2/types/mat.py:940-954).  Nothing is
    needed on the high side: those node planes are ghosts read by the rank's own last cube layer.
    """
    if numbering not in NUMBERINGS:
        raise ValueError(f"numbering must be one of {NUMBERINGS}")
    nx, ny, nz = (int(n),) * 3 if np.isscalar(n) else tuple(int(v) for v in n)
    ndim = (nx, ny, nz)
    pgrid = partition_grid(nranks, partition)
    if any(pg > nd for pg, nd in zip(pgrid, ndim)):
        raise ValueError(f"partition {pgrid} has more ranks than cubes along an axis of {ndim}")
    rc = (rank % pgrid[0], (rank // pgrid[0]) % pgrid[1], rank // (pgrid[0] * pgrid[1]))       # (rx, ry, rz)

    def cube_range(q):
        """own cubes [lo, hi) of every axis for the rank at grid position q"""
        return tuple(((ndim[d] * q[d]) // pgrid[d], (ndim[d] * (q[d] + 1)) // pgrid[d]) for d in range(3))

    own = cube_range(rc)
    last = tuple(rc[d] == pgrid[d] - 1 for d in range(3))
    # local cubes: the own box + one ghost layer on the low side of every axis that has a lower neighbour
    glo = tuple(own[d][0] - 1 if (rc[d] > 0 and ghost_cells) else own[d][0] for d in range(3))
    ghi = tuple(own[d][1] for d in range(3))
    ldim = tuple(ghi[d] - glo[d] for d in range(3))
    if numbering != "tiled":
        tile = ldim                  # one tile = plain lexicographic traversal
    kk, jj, ii = np.meshgrid(np.arange(glo[2], ghi[2], dtype=np.int32), np.arange(glo[1], ghi[1], dtype=np.int32),
                             np.arange(glo[0], ghi[0], dtype=np.int32), indexing="ij")
    ii, jj, kk = ii.ravel(), jj.ravel(), kk.ravel()
    cijk = (ii, jj, kk)
    # class of each cube: 0 core, 1 owned (touches a node plane owned by the next rank of an axis), 2 ghost
    ccls = np.zeros(ii.shape, dtype=np.int8)
    for d in range(3):
        if not last[d]:
            ccls[cijk[d] == own[d][1] - 1] = 1
    for d in range(3):
        ccls[cijk[d] < own[d][0]] = 2
    key = _tile_keys(ii - glo[0], jj - glo[1], kk - glo[2], ldim[0], ldim[1], ldim[2], tile) + ccls.astype(np.int64) * (1 << 50)
    if numbering == "random":
        key = np.random.default_rng(seed + 1000 * rank).permutation(len(ii)).astype(np.int64) + ccls.astype(np.int64) * (1 << 50)
    order = np.argsort(key, kind="stable")
    ii, jj, kk, ccls = ii[order], jj[order], kk[order], ccls[order]
    ncube = len(ii)
    sizes_c = tuple(int(6 * (ccls <= c).sum()) for c in (0, 1, 2))
    cell_set = op2.Set(sizes_c, "cells")
    # traversal tiles = natural plan blocks: boundaries (in cells) where the tile (or the class) changes
    tkey = key[order] // (tile[0] * tile[1] * tile[2])
    cuts = np.nonzero(np.diff(tkey))[0] + 1
    cell_blocks = (6 * np.concatenate([[0], cuts, [ncube]])).astype(np.int32)

    def node_boxes(p, q):
        """(owned, local) lattice-node boxes [lo, hi) per axis of the rank at grid position q, spacing 1/(p*n)"""
        o = cube_range(q)
        owned = tuple((p * o[d][0], p * o[d][1] if q[d] < pgrid[d] - 1 else p * ndim[d] + 1) for d in range(3))
        local = tuple((p * (o[d][0] - 1 if (q[d] > 0 and ghost_cells) else o[d][0]), p * o[d][1] + 1) for d in range(3))
        return owned, local

    def lattice_space(p):
        """CG_p nodes live on the lattice of spacing 1/(p*n); CG2 edge nodes are vertex sums."""
        (oown, oloc) = node_boxes(p, rc)
        lo = tuple(b[0] for b in oloc)
        L = tuple(b[1] - b[0] for b in oloc)           # local lattice planes per axis
        # index of a lattice point inside the local box is affine in its coordinates, so it is computed once per CUBE CORNER and
        # carried to the cells through the tetrahedron table: a CG2 vertex node sits at twice the vertex, an edge node at the sum
        # of its two vertices (coordinates in units of 1/(p n))
        def box_index(x, y, z):
            """index of global lattice point (x, y, z) inside the local lattice box"""
            return ((np.asarray(z, dtype=np.int64) - lo[2]) * L[1] + (y - lo[1])) * L[0] + (x - lo[0])
        lin = ((kk.astype(np.int64)[:, None] + _CORNER[:, 2][None, :]) * L[1] + (jj[:, None] + _CORNER[:, 1][None, :])) * L[0] \
            + (ii[:, None] + _CORNER[:, 0][None, :])                          # (ncube, 8): z L1 L0 + y L0 + x of every corner
        shift = (lo[2] * L[1] + lo[1]) * L[0] + lo[0]
        vlin = lin[:, _TETS]                                                    # (ncube, 6, 4)
        if p == 1:
            box = vlin - shift
        else:
            box = np.concatenate([2 * vlin, vlin[..., _TET_EDGES[:, 0]] + vlin[..., _TET_EDGES[:, 1]]], axis=-1) - shift
        arity = box.shape[-1]
        box = box.reshape(ncube * 6, arity)
        # ---- number the lattice nodes of the box: class, then tile traversal order
        zz, yy, xx = np.meshgrid(np.arange(oloc[2][0], oloc[2][1], dtype=np.int32), np.arange(oloc[1][0], oloc[1][1], dtype=np.int32),
                                 np.arange(oloc[0][0], oloc[0][1], dtype=np.int32), indexing="ij")
        xx, yy, zz = xx.ravel(), yy.ravel(), zz.ravel()
        xyz_ = (xx, yy, zz)
        ncls = np.zeros(xx.shape, dtype=np.int8)
        for d in range(3):
            if not last[d]:
                ncls[xyz_[d] >= p * (own[d][1] - 1)] = 1       # owned, but read by cells that also read ghosts
        for d in range(3):
            ncls[(xyz_[d] < oown[d][0]) | (xyz_[d] >= oown[d][1])] = 2
        tl = tuple(p * t for t in tile)
        lp = tuple(p * v for v in ldim)                          # lattice cells per axis of the local box
        rel = tuple(xyz_[d] - lo[d] for d in range(3))
        nkey = _tile_keys(np.minimum(rel[0], lp[0] - 1), np.minimum(rel[1], lp[1] - 1), np.minimum(rel[2], lp[2] - 1),
                          lp[0], lp[1], lp[2], tl)
        # tie-break inside a tile by the true coordinates so keys are unique
        nkey = (nkey + ncls.astype(np.int64) * (1 << 50)) * 8 + (rel[2] // lp[2]) * 4 + (rel[1] // lp[1]) * 2 + rel[0] // lp[0]
        if numbering == "lexicographic":
            # first appearance in the cell traversal (vertices of a cell before its edge nodes, as the closure walk
            # of dmcommon.pyx:2688-2712 meets them); lattice points no local cell touches keep the grid order, last
            flat = box.reshape(-1)
            ptype = np.int32 if len(flat) < (1 << 31) - 1 else np.int64
            first = np.full(len(xx), np.iinfo(ptype).max, dtype=ptype)
            # first occurrence of every node: assign positions in REVERSE order, the last write (= earliest position) stays
            first[flat[::-1]] = np.arange(len(flat) - 1, -1, -1, dtype=ptype)
            nkey = ncls.astype(np.int64) * (1 << 50) + np.where(first == np.iinfo(ptype).max, 1 << 49, first.astype(np.int64))
            nkey = nkey * 2                        # keep "key // 8 // tile volume" below meaningful only for "tiled"
        elif numbering == "random":
            nkey = ncls.astype(np.int64) * (1 << 50) + np.random.default_rng(seed + 7 + 1000 * rank + p).permutation(len(xx))
        norder = np.argsort(nkey, kind="stable")
        newnum = np.empty(len(norder), dtype=np.int32)
        newnum[norder] = np.arange(len(norder), dtype=np.int32)
        sizes_n = tuple(int((ncls <= c).sum()) for c in (0, 1, 2))
        cmap = newnum[box]
        pts = np.stack([xx[norder] * (1.0 / (p * nx)), yy[norder] * (1.0 / (p * ny)), zz[norder] * (1.0 / (p * nz))], axis=1)
        bnd = np.nonzero((xx[norder] == 0) | (xx[norder] == p * nx) | (yy[norder] == 0) | (yy[norder] == p * ny)
                         | (zz[norder] == 0) | (zz[norder] == p * nz))[0].astype(np.int32)
        halo = HaloLists(rank, nranks)
        if nranks > 1:
            def box_nodes(a, b):
                """local numbers of the lattice points in the intersection of boxes a and b, in global (z, y, x) order --
                the same order on both sides of the exchange"""
                r = [(max(a[d][0], b[d][0]), min(a[d][1], b[d][1])) for d in range(3)]
                if any(hi <= lo_ for lo_, hi in r):
                    return None
                z, y, x = np.meshgrid(*[np.arange(r[d][0], r[d][1], dtype=np.int64) for d in (2, 1, 0)], indexing="ij")
                return newnum[box_index(x.ravel(), y.ravel(), z.ravel())]
            for qz in range(max(rc[2] - 1, 0), min(rc[2] + 2, pgrid[2])):
                for qy in range(max(rc[1] - 1, 0), min(rc[1] + 2, pgrid[1])):
                    for qx in range(max(rc[0] - 1, 0), min(rc[0] + 2, pgrid[0])):
                        q = (qx, qy, qz)
                        if q == rc:
                            continue
                        qrank = (qz * pgrid[1] + qy) * pgrid[0] + qx
                        qown, qloc = node_boxes(p, q)
                        snd = box_nodes(oown, qloc)        # owned here, present (as ghosts) on q
                        rcv = box_nodes(qown, oloc)        # owned by q, ghosts here
                        if snd is not None:
                            halo.send[qrank] = snd
                        if rcv is not None:
                            halo.recv[qrank] = rcv
        node_set = op2.Set(sizes_n, f"cg{p}_nodes")
        m = op2.Map(cell_set, node_set, arity, cmap, f"cell_cg{p}")
        m._has_negative = False          # (built from lattice arithmetic: spares the Mat loops a scan of the table, Parloop._reject_negative_mat_maps)
        if numbering == "tiled":
            m.preferred_blocks = _split_blocks(cell_blocks, arity)
            # node ranges of the traversal tiles (row blocks for owner-computes-rows matrix assembly)
            ntile = (nkey[norder] // 8) // (tl[0] * tl[1] * tl[2])
            m.preferred_node_blocks = np.concatenate([[0], np.nonzero(np.diff(ntile))[0] + 1, [len(norder)]]).astype(np.int32)
        return FunctionSpaceData(p, node_set, m, pts, halo, bnd, (p * nx + 1) * (p * ny + 1) * (p * nz + 1))

    spaces = {}
    for p in sorted(set(degrees) | {1}):
        spaces[p] = lattice_space(p)
    cs = spaces[1]
    xyz = cs.node_points.copy()
    if perturb:
        hgrid = 1.0 / nx
        inner = ((xyz > 1e-12) & (xyz < 1 - 1e-12)).all(axis=1)
        d = perturb * hgrid * np.stack([np.sin(2 * np.pi * xyz[:, 1]) * np.sin(2 * np.pi * xyz[:, 2]),
                                        np.sin(2 * np.pi * xyz[:, 0]) * np.sin(2 * np.pi * xyz[:, 2]),
                                        np.sin(2 * np.pi * xyz[:, 0]) * np.sin(2 * np.pi * xyz[:, 1])], axis=1)
        xyz[inner] += d[inner]
    coords = op2.Dat(cs.node_set ** 3, xyz, np.float64, "coordinates")
    from .halo import attach_halo
    for sp in spaces.values():
        attach_halo(sp)
    mesh = Mesh(3, cell_set, coords, cs, spaces, 6 * nx * ny * nz, (nx, ny, nz))
    mesh.partition = pgrid
    return mesh


def UnitSquareMesh(nx, ny, degrees=(1,), tile=(16, 16), perturb=0.0):
    """Triangulated unit square, 2 triangles per square, "left" diagonal (utility_meshes.py:661-662)."""
    jj, ii = np.meshgrid(np.arange(ny, dtype=np.int32), np.arange(nx, dtype=np.int32), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    key = _tile_keys(ii, jj, np.zeros_like(ii), nx, ny, 1, (tile[0], tile[1], 1))
    order = np.argsort(key, kind="stable")
    ii, jj = ii[order], jj[order]
    nsq = len(ii)
    cell_set = op2.Set(2 * nsq, "cells")
    tkey = key[order] // (tile[0] * tile[1])
    cuts = np.nonzero(np.diff(tkey))[0] + 1
    cell_blocks = (2 * np.concatenate([[0], cuts, [nsq]])).astype(np.int32)
    # corners a=(i,j) b=(i+1,j) c=(i,j+1) d=(i+1,j+1); "left" diagonal joins b-c
    tri = np.array([[(0, 0), (1, 0), (0, 1)], [(1, 0), (1, 1), (0, 1)]], dtype=np.int32)   # (2, 3, 2)
    vx = ii[:, None, None] + tri[None, :, :, 0]
    vy = jj[:, None, None] + tri[None, :, :, 1]

    def lattice_space(p):
        Lx, Ly = p * nx + 1, p * ny + 1
        if p == 1:
            nxs, nys = vx, vy
        else:
            ex = vx[..., _TRI_EDGES[:, 0]] + vx[..., _TRI_EDGES[:, 1]]
            ey = vy[..., _TRI_EDGES[:, 0]] + vy[..., _TRI_EDGES[:, 1]]
            nxs, nys = np.concatenate([2 * vx, ex], -1), np.concatenate([2 * vy, ey], -1)
        arity = nxs.shape[-1]
        box = (nys.astype(np.int64) * Lx + nxs).reshape(nsq * 2, arity)
        yy, xx = np.meshgrid(np.arange(Ly, dtype=np.int32), np.arange(Lx, dtype=np.int32), indexing="ij")
        xx, yy = xx.ravel(), yy.ravel()
        nkey = _tile_keys(np.minimum(xx, p * nx - 1), np.minimum(yy, p * ny - 1), np.zeros_like(xx), p * nx, p * ny, 1,
                          (p * tile[0], p * tile[1], 1)) * 4 + (yy // (p * ny)) * 2 + xx // (p * nx)
        norder = np.argsort(nkey, kind="stable")
        newnum = np.empty(len(norder), dtype=np.int32)
        newnum[norder] = np.arange(len(norder), dtype=np.int32)
        pts = np.stack([xx[norder] / (p * nx), yy[norder] / (p * ny)], axis=1)
        bnd = np.nonzero((xx[norder] == 0) | (xx[norder] == p * nx) | (yy[norder] == 0) | (yy[norder] == p * ny))[0].astype(np.int32)
        node_set = op2.Set(len(norder), f"cg{p}_nodes")
        m = op2.Map(cell_set, node_set, arity, newnum[box], f"cell_cg{p}")
        m._has_negative = False          # (built from lattice arithmetic: spares the Mat loops a scan of the table, Parloop._reject_negative_mat_maps)
        m.preferred_blocks = _split_blocks(cell_blocks, arity)
        return FunctionSpaceData(p, node_set, m, pts, HaloLists(), bnd, Lx * Ly)

    spaces = {p: lattice_space(p) for p in sorted(set(degrees) | {1})}
    cs = spaces[1]
    xy = cs.node_points.copy()
    if perturb:
        inner = ((xy > 1e-12) & (xy < 1 - 1e-12)).all(axis=1)
        d = perturb / nx * np.stack([np.sin(2 * np.pi * xy[:, 1]), np.sin(2 * np.pi * xy[:, 0])], axis=1)
        xy[inner] += d[inner]
    coords = op2.Dat(cs.node_set ** 2, xy, np.float64, "coordinates")
    return Mesh(2, cell_set, coords, cs, spaces, 2 * nx * ny, (nx, ny))


# ------------------------------------------------------------------------------------------
# extruded hexahedra: ExtrudedMesh(UnitSquareMesh(n, n, quadrilateral=True), layers)  (config C3)
# ------------------------------------------------------------------------------------------
@dataclass
class ExtrudedHexMesh:
    """Column-structured hex mesh with Q_k = CG_k (x) CG_k (x) CG_k nodes (firedrake/mesh.py:3466,
    extrusion_utils.py:342-367).  Maps hold the BOTTOM cell of each column; node = map + offset*layer
    (pyop2/codegen/builder.py:94-124) with offset = k for every entry of Q_k and 1 for the Q1 coordinates.
    Nodes of one vertical line are contiguous (firedrake numbers column DoFs consecutively, mesh.py:1932-1950).
    Local DoF order inside a cell: (a, b, c) -> (a*(k+1) + b)*(k+1) + c, c along the extrusion direction."""
    n: int
    layers: int            # cell layers
    degree: int
    base_set: op2.Set
    cell_set: op2.ExtrudedSet
    node_set: op2.Set
    cell_node_map: op2.Map
    coord_node_set: op2.Set
    coord_map: op2.Map
    coordinates: op2.Dat
    node_points: np.ndarray

    @property
    def ncells(self):
        return self.base_set.size * self.layers


def _gll_nodes(k):
    """Gauss-Lobatto-Legendre points on [0, 1] (k+1 of them): the nodes of CG_k on an interval."""
    if k == 1:
        return np.array([0.0, 1.0])
    from numpy.polynomial import legendre as leg
    c = np.zeros(k + 1)
    c[k] = 1.0
    return 0.5 * (np.concatenate([[-1.0], np.sort(leg.legroots(leg.legder(c))), [1.0]]) + 1.0)


def make_extruded_hex_mesh(n, layers=None, degree=4, tile=(4, 4), perturb=0.1):
    layers = layers or n
    k = degree
    jj, ii = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    order = np.argsort(_tile_keys(ii, jj, np.zeros_like(ii), n, n, 1, (tile[0], tile[1], 1)), kind="stable")
    ii, jj = ii[order], jj[order]
    ncol = len(ii)
    base = op2.Set(ncol, "base_cells")
    ext = op2.ExtrudedSet(base, layers=layers + 1)

    def space(p):
        Lb = p * n + 1                      # base lattice points per axis
        nz = p * layers + 1                 # nodes per vertical line
        yy, xx = np.meshgrid(np.arange(Lb, dtype=np.int32), np.arange(Lb, dtype=np.int32), indexing="ij")
        xx, yy = xx.ravel(), yy.ravel()
        key = _tile_keys(np.minimum(xx, p * n - 1), np.minimum(yy, p * n - 1), np.zeros_like(xx), p * n, p * n, 1,
                         (p * tile[0], p * tile[1], 1)) * 4 + (yy // (p * n)) * 2 + xx // (p * n)
        border = np.argsort(key, kind="stable")
        bnum = np.empty(len(border), dtype=np.int64)
        bnum[border] = np.arange(len(border))
        a, b, c = np.meshgrid(np.arange(p + 1), np.arange(p + 1), np.arange(p + 1), indexing="ij")
        a, b, c = a.ravel(), b.ravel(), c.ravel()
        bx = p * ii[:, None] + a[None, :]
        by = p * jj[:, None] + b[None, :]
        cmap = (bnum[by.astype(np.int64) * Lb + bx] * nz + c[None, :]).astype(np.int32)
        nset = op2.Set(Lb * Lb * nz, f"q{p}_nodes")
        m = op2.Map(ext, nset, (p + 1) ** 3, cmap, f"cell_q{p}", offset=[p] * (p + 1) ** 3)
        # physical points of the nodes: CG_p on an interval has its nodes at the GLL points of each cell
        gll = _gll_nodes(p)

        def lat2x(lat, ncell):
            cell = np.minimum(lat // p, ncell - 1)
            return (cell + gll[lat - cell * p]) / ncell
        bxy = np.stack([lat2x(xx[border], n), lat2x(yy[border], n)], axis=1)
        pts = np.empty((Lb * Lb * nz, 3))
        pts[:, :2] = np.repeat(bxy, nz, axis=0)
        pts[:, 2] = np.tile(lat2x(np.arange(nz), layers), Lb * Lb)
        return nset, m, pts

    nset, cmap, pts = space(k)
    cset, xmap, xpts = space(1)
    xyz = xpts.copy()
    if perturb:
        h = 1.0 / n
        inner = ((xyz > 1e-12) & (xyz < 1 - 1e-12)).all(axis=1)
        d = perturb * h * np.stack([np.sin(2 * np.pi * xyz[:, 1]) * np.sin(2 * np.pi * xyz[:, 2]),
                                    np.sin(2 * np.pi * xyz[:, 0]) * np.sin(2 * np.pi * xyz[:, 2]),
                                    np.sin(2 * np.pi * xyz[:, 0]) * np.sin(2 * np.pi * xyz[:, 1])], axis=1)
        xyz[inner] += d[inner]
    coords = op2.Dat(cset ** 3, xyz, np.float64, "coordinates")
    return ExtrudedHexMesh(n, layers, k, base, ext, nset, cmap, cset, xmap, coords, pts)


# ------------------------------------------------------------------------------------------
# quadrilateral mesh with facet sets: UnitSquareMesh(n, n, quadrilateral=True)  (config C4, DG advection)
# ------------------------------------------------------------------------------------------
@dataclass
class QuadMesh:
    """Cells, interior/exterior facet sets and the maps the DG-advection demo needs.

    * DQ1: 4 DoFs per cell, dof = 4*cell + (a*2 + b)  (discontinuous: no sharing);
    * Q1 (vector CG1: coordinates and velocity): vertex nodes, cell vertex order (a, b) -> a*2 + b;
    * interior_facet maps have arity 2*ndof: [cell0 dofs..., cell1 dofs...] ('+' then '-',
      firedrake/functionspaceimpl.py:814-829, dmcommon.pyx:1660-1674);
    * ``local_facet_dat``: uint32 (nfacets, 2) / (nfacets, 1): local facet number inside each adjacent
      cell (firedrake/mesh.py:196-201); reference-quad facets: 0: xi=0, 1: xi=1, 2: eta=0, 3: eta=1.
    """
    n: int
    cell_set: op2.Set
    int_facet_set: op2.Set
    ext_facet_set: op2.Set
    dq_set: op2.Set
    q1_set: op2.Set
    cell_dq: op2.Map
    cell_q1: op2.Map
    int_dq: op2.Map
    int_q1: op2.Map
    ext_dq: op2.Map
    ext_q1: op2.Map
    int_local_facet: op2.Dat
    ext_local_facet: op2.Dat
    coordinates: op2.Dat
    dq_points: np.ndarray


def make_quad_mesh(n, tile=(16, 16), perturb=0.0):
    jj, ii = np.meshgrid(np.arange(n, dtype=np.int32), np.arange(n, dtype=np.int32), indexing="ij")
    ii, jj = ii.ravel(), jj.ravel()
    ckey = _tile_keys(ii, jj, np.zeros_like(ii), n, n, 1, (tile[0], tile[1], 1))
    order = np.argsort(ckey, kind="stable")
    ii, jj = ii[order], jj[order]
    ncell = n * n
    cellnum = np.empty(ncell, dtype=np.int64)
    cellnum[jj.astype(np.int64) * n + ii] = np.arange(ncell)
    cells, dq, q1n = op2.Set(ncell, "cells"), op2.Set(4 * ncell, "dq1_nodes"), op2.Set((n + 1) ** 2, "q1_nodes")
    # vertices numbered in tile order as well
    yy, xx = np.meshgrid(np.arange(n + 1, dtype=np.int32), np.arange(n + 1, dtype=np.int32), indexing="ij")
    xx, yy = xx.ravel(), yy.ravel()
    vkey = _tile_keys(np.minimum(xx, n - 1), np.minimum(yy, n - 1), np.zeros_like(xx), n, n, 1, (tile[0], tile[1], 1)) * 4 + (yy // n) * 2 + xx // n
    vorder = np.argsort(vkey, kind="stable")
    vnum = np.empty(len(vorder), dtype=np.int64)
    vnum[vorder] = np.arange(len(vorder))
    ab = np.array([(0, 0), (0, 1), (1, 0), (1, 1)], dtype=np.int32)       # vertex a*2+b -> (a, b)
    cq1 = vnum[(jj[:, None] + ab[None, :, 1]).astype(np.int64) * (n + 1) + ii[:, None] + ab[None, :, 0]].astype(np.int32)
    cdq = (4 * np.arange(ncell, dtype=np.int64)[:, None] + np.arange(4)[None, :]).astype(np.int32)
    xy = np.stack([xx[vorder] / n, yy[vorder] / n], axis=1).astype(np.float64)
    if perturb:
        inner = ((xy > 1e-12) & (xy < 1 - 1e-12)).all(axis=1)
        d = perturb / n * np.stack([np.sin(2 * np.pi * xy[:, 1]), np.sin(2 * np.pi * xy[:, 0])], axis=1)
        xy[inner] += d[inner]
    # interior facets: vertical (between (i,j) and (i+1,j)): + facet 1, - facet 0; horizontal: + facet 3, - facet 2
    iv, jv = np.meshgrid(np.arange(n - 1), np.arange(n), indexing="xy")
    c0v, c1v = cellnum[jv.ravel() * n + iv.ravel()], cellnum[jv.ravel() * n + iv.ravel() + 1]
    ih, jh = np.meshgrid(np.arange(n), np.arange(n - 1), indexing="xy")
    c0h, c1h = cellnum[jh.ravel() * n + ih.ravel()], cellnum[(jh.ravel() + 1) * n + ih.ravel()]
    c0 = np.concatenate([c0v, c0h]); c1 = np.concatenate([c1v, c1h])
    lf = np.concatenate([np.tile([1, 0], (len(c0v), 1)), np.tile([3, 2], (len(c0h), 1))]).astype(np.uint32)
    forder = np.argsort(np.minimum(c0, c1), kind="stable")            # facets follow the cell traversal
    c0, c1, lf = c0[forder], c1[forder], lf[forder]
    ifs = op2.Set(len(c0), "interior_facets")
    # exterior facets
    ec, ef = [], []
    for (sel_i, sel_j, f) in ((0, None, 0), (n - 1, None, 1), (None, 0, 2), (None, n - 1, 3)):
        for t in range(n):
            i, j = (sel_i, t) if sel_i is not None else (t, sel_j)
            ec.append(cellnum[j * n + i]); ef.append(f)
    ec, ef = np.array(ec), np.array(ef, dtype=np.uint32)
    eorder = np.argsort(ec, kind="stable")
    ec, ef = ec[eorder], ef[eorder]
    efs = op2.Set(len(ec), "exterior_facets")
    dq_pts = xy[cq1].reshape(-1, 2)
    return QuadMesh(n, cells, ifs, efs, dq, q1n,
                    op2.Map(cells, dq, 4, cdq, "cell_dq1"), op2.Map(cells, q1n, 4, cq1, "cell_q1"),
                    op2.Map(ifs, dq, 8, np.concatenate([cdq[c0], cdq[c1]], axis=1), "ifacet_dq1"),
                    op2.Map(ifs, q1n, 8, np.concatenate([cq1[c0], cq1[c1]], axis=1), "ifacet_q1"),
                    op2.Map(efs, dq, 4, cdq[ec], "efacet_dq1"), op2.Map(efs, q1n, 4, cq1[ec], "efacet_q1"),
                    op2.Dat(ifs ** 2, lf, np.uint32, "interior_local_facet"),
                    op2.Dat(efs ** 1, ef.reshape(-1, 1), np.uint32, "exterior_local_facet"),
                    op2.Dat(q1n ** 2, xy, np.float64, "coordinates"), dq_pts)
