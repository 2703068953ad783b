"""Parloop execution: argument marshalling, halo protocol, launches.

Mirror of pyop2/parloop.py:167-540 (Parloop) and :545-743 (legacy ``par_loop`` API).
``Parloop.__call__`` keeps the reference's protocol (parloop.py:243-260):

    halo begin -> compute(core part) -> halo end -> compute(owned part)
               -> global reductions -> reverse halo exchange for INC/MIN/MAX Dats

with the wrapper launched asynchronously on the device, so the RCCL exchange posted in
``global_to_local_begin`` overlaps the core-entity kernel exactly where the reference
overlaps MPI with ``_compute(core_part)``.
"""
from __future__ import annotations

import contextlib
from dataclasses import dataclass
from typing import Any, Optional, Tuple

import numpy as np

from . import _lib
from .configuration import configuration
from .device import DeviceBuffer
from .kernel import (CStringLocalKernel, DatKernelArg, GlobalKernel, GlobalKernelArg, MapKernelArg, MatKernelArg,
                     MixedDatKernelArg, MixedMatKernelArg, PermutedMapKernelArg)
from .op2types import (INC, MAX, MIN, READ, WRITE, Access, Dat, ExtrudedSet, Global, Map, MapValueError,
                       Mat, MixedDat, MixedMap, PermutedMap, Set, SetTypeError, Subset)

_NO_EVENT = contextlib.nullcontext()          # (tracing off: no generator-based context manager per launch)
_CHECK_ARGLISTS = False                       # tests: build every argument list both ways and compare (tests/conftest.py, FD_TEST_ARGLISTS)


def _same_arglists(layout, out, ref):
    """tests (``_CHECK_ARGLISTS``): the pre-dispatched argument list equals the branch-per-entry one.  ``plan_copy`` entries are
    left out: asking twice in one call is itself a state change there (a Dat seen unchanged a second time gets its copy)."""
    a = [int(v) if v is not None else 0 for d, v in zip(layout, out) if d[0] != "plan_copy"]
    b = [int(v) if v is not None else 0 for d, v in zip(layout, ref) if d[0] != "plan_copy"]
    assert len(out) == len(ref) == len(layout) and a == b, (layout, out, ref)


def _restore_versions(vers):
    for d, v in vers:
        try:
            d.dat_version = v
        except AttributeError:            # (a view or a mixed carrier: its version is its parts')
            pass


def small_loop_leaf(n: int, leaf: int, floor: int) -> int:
    """Leaf size (entities or rows per block of a derived order) for a loop over ``n`` of them: the configured ``leaf``, unless that
    leaves the device short of ``configuration["small_loop_blocks"]`` blocks (4 per CU) -- then ``n // blocks``, but not below
    ``floor`` (a block of fewer entities than lanes idles most of its workgroup) and never above ``leaf`` (the LDS budget it stands
    for).  0 blocks = off.  tools/size_sweep.py, profiles/r6s3_size_sweep*.txt."""
    sb = int(configuration["small_loop_blocks"])
    if sb <= 0:
        return int(leaf)
    return int(min(leaf, max(floor, n // sb)))


# ---- parloop arguments (pyop2/parloop.py:36-165) ----------------------------------------------
@dataclass
class GlobalParloopArg:
    data: Global

    @property
    def maps(self):
        return ()


@dataclass
class DatParloopArg:
    data: Dat
    map_: Optional[Map] = None

    def __post_init__(self):
        if self.map_ is not None and configuration["type_check"]:
            m = self.map_
            if m.iterset.total_size > 0 and len(m.values_with_halo) == 0:
                raise MapValueError(f"{m} is not initialized")

    @property
    def maps(self):
        return () if self.map_ is None else (self.map_,)


@dataclass
class MatParloopArg:
    data: Mat
    maps: Tuple[Map, Map]
    lgmaps: Optional[Any] = None      # (row_lgmap, col_lgmap) int32 arrays, -1 = dropped (parloop.py:279-302)


@dataclass
class MixedDatParloopArg:            # pyop2/parloop.py:97-118
    data: MixedDat
    map_: MixedMap

    @property
    def maps(self):
        return tuple(m for m in self.map_.split if m is not None)

    @property
    def split(self):
        return [DatParloopArg(d, m) for d, m in zip(self.data.split, self.map_.split)]


@dataclass
class MixedMatParloopArg:            # pyop2/parloop.py:137-152
    data: Mat
    maps: Tuple[MixedMap, MixedMap]
    lgmaps: Optional[Any] = None      # one (row_lgmap, col_lgmap) pair per block, row-major (parloop.py:286-291)

    @property
    def split(self):
        rmaps, cmaps = self.maps
        out = []
        for i, rm in enumerate(rmaps):
            for j, cm in enumerate(cmaps):
                lg = None if self.lgmaps is None else self.lgmaps[i * len(cmaps) + j]
                out.append(MatParloopArg(self.data[i, j], (rm, cm), lg))
        return out


# ---- legacy args (pyop2/parloop.py:545-660): what dat(access, map) returns -------------------
@dataclass
class DatLegacyArg:
    data: Dat
    map_: Optional[Map]
    access: Access

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def global_kernel_arg(self):
        # a DatView passes its parent's row shape and the component it shows (parloop.py:595-601)
        return DatKernelArg(self.data.dataset.dim, _map_kernel_arg(self.map_), getattr(self.data, "index", None))

    @property
    def parloop_arg(self):
        return DatParloopArg(self.data, self.map_)


@dataclass
class GlobalLegacyArg:
    data: Global
    access: Access

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def global_kernel_arg(self):
        return GlobalKernelArg(self.data.dim)

    @property
    def parloop_arg(self):
        return GlobalParloopArg(self.data)


@dataclass
class MatLegacyArg:
    data: Mat
    maps: Tuple[Map, Map]
    access: Access
    lgmaps: Optional[Any] = None
    unroll_map: bool = False          # MatSetValuesLocal on dof indices; lgmaps are then per dof (mat.py:700-716)

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def global_kernel_arg(self):
        ((rdim, cdim),), = self.data.dims
        return MatKernelArg((rdim, cdim), tuple(_map_kernel_arg(m) for m in self.maps), unroll=bool(self.unroll_map),
                            lgmaps=self.lgmaps is not None)

    @property
    def parloop_arg(self):
        return MatParloopArg(self.data, self.maps, self.lgmaps)


@dataclass
class MixedDatLegacyArg:             # pyop2/parloop.py:618-641
    data: MixedDat
    map_: MixedMap
    access: Access

    def __post_init__(self):
        if not isinstance(self.map_, MixedMap) or len(self.map_) != len(self.data):
            raise MapValueError("a MixedDat is accessed through a MixedMap with one Map per component")

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def split(self):
        return [DatLegacyArg(d, m, self.access) for d, m in zip(self.data.split, self.map_.split)]

    @property
    def global_kernel_arg(self):
        return MixedDatKernelArg(tuple(a.global_kernel_arg for a in self.split))

    @property
    def parloop_arg(self):
        return MixedDatParloopArg(self.data, self.map_)


@dataclass
class MixedMatLegacyArg:             # pyop2/parloop.py:679-706
    data: Mat
    maps: Tuple[MixedMap, MixedMap]
    access: Access
    lgmaps: Optional[Any] = None
    unroll_map: bool = False

    def __post_init__(self):
        shape = self.data.sparsity.shape
        if not all(isinstance(m, MixedMap) for m in self.maps) or (len(self.maps[0]), len(self.maps[1])) != shape:
            raise MapValueError("a mixed Mat is accessed through a pair of MixedMaps matching its block shape")
        if self.lgmaps is not None and len(self.lgmaps) != shape[0] * shape[1]:
            raise ValueError("lgmaps of a mixed Mat: one (row, col) pair per block, in row-major order")

    @property
    def dtype(self):
        return self.data.dtype

    @property
    def split(self):
        """Rows of per-block legacy arguments."""
        rmaps, cmaps = self.maps
        return [[MatLegacyArg(self.data[i, j], (rm, cm), self.access,
                              None if self.lgmaps is None else self.lgmaps[i * len(cmaps) + j], self.unroll_map)
                 for j, cm in enumerate(cmaps)] for i, rm in enumerate(rmaps)]

    @property
    def global_kernel_arg(self):
        return MixedMatKernelArg(tuple(b.global_kernel_arg for row in self.split for b in row),
                                 self.data.sparsity.shape)

    @property
    def parloop_arg(self):
        return MixedMatParloopArg(self.data, self.maps, self.lgmaps)


def _tuple_or_none(x):
    return None if x is None else tuple(int(v) for v in x)


def _map_kernel_arg(m):
    """One MapKernelArg per base Map object, so GlobalKernel de-duplicates by identity."""
    if m is None:
        return None
    base = m._base()
    # (kept ON the Map: a module-level table keyed by id() with the Map as its value kept every Map ever used in a parloop alive --
    #  its device values, the plans and derived orders cached on it: 26 MB per problem of 384 k cells, profiles/r6s3_leak_probe.txt)
    mk = base.__dict__.get("_map_kernel_arg")
    if mk is None:
        mk = base.__dict__["_map_kernel_arg"] = MapKernelArg(base.arity, base.offset, _tuple_or_none(base.offset_quotient))
    if isinstance(m, PermutedMap):
        return PermutedMapKernelArg(mk, tuple(int(p) for p in m.permutation))
    return mk


class LocalityOrder:
    """A backend-derived entity order: the nodes of the loop's position field are partitioned into k-d leaves (fd_kd_order)
    and every entity joins the lowest leaf among its nodes (fd_group_entities).  ``buf`` = device int32 list of the entities,
    ``blocks`` = block boundaries (positions in that list) at the leaf changes, and a process-unique ``serial`` (cache keys
    must not use device addresses: a freed buffer's address is reused)."""
    _serial = [0]

    def __init__(self, buf, leaf_starts, target):
        self.buf, self.target = buf, int(target)
        self.ptr = buf.ptr
        LocalityOrder._serial[0] += 1
        self.serial = LocalityOrder._serial[0]
        self.blocks = np.asarray(leaf_starts, dtype=np.int64)

    @staticmethod
    def cut(counts, target):
        """Block boundaries from the entities per leaf: empty leaves dropped, a leaf above 1.5 x target split into equal parts
        (leaves at a domain boundary hold fewer entities; nothing is merged: a block is one box)."""
        sizes = []
        for c in counts[counts > 0].tolist():
            if c > (3 * target) // 2:
                parts = -(-c // target)
                q, r = divmod(c, parts)
                sizes.extend([q + 1] * r + [q] * (parts - r))
            else:
                sizes.append(c)
        return np.concatenate([[0], np.cumsum(np.asarray(sizes, dtype=np.int64))]).astype(np.int64)


def kd_order(points_ptr, pdim, n, base, leaf_size):
    """(device int32 order of the n points, leaf boundaries) of the k-d partition into leaves of ``leaf_size`` points."""
    import ctypes
    buf = DeviceBuffer(max(n, 1) * 4)
    max_leaves = int(-(-n // max(leaf_size, 1))) + 1
    starts = np.zeros(max_leaves + 1, dtype=np.int32)
    nl = ctypes.c_int32()
    _lib.call("fd_kd_order", points_ptr, int(pdim), int(n), int(base), int(leaf_size), buf.ptr, starts.ctypes.data, max_leaves,
              ctypes.byref(nl), None)
    return buf, starts[:nl.value + 1].astype(np.int64)


def kd_order_of(dat, n, leaf_size):
    """kd_order of the first ``n`` points of the position field ``dat``, kept on the Dat for its current version: the staged loops
    (entity groups around leaves of nodes) and the owner-computes-rows loops (row blocks = leaves of nodes) of one mesh ask for the
    same partition when their leaf sizes agree (``kd_leaf_size`` makes 260 and 256 agree), and a k-d order of 10 M points is 30-40 ms."""
    if getattr(dat, "dat_version", None) is None:      # a borrowed carrier: no version to tell a moved mesh by
        return kd_order(dat._dev_ptr(False), dat.cdim, n, 0, leaf_size)
    cache = dat.__dict__.setdefault("_kd_orders", {})
    key = (dat.dat_version, int(n), int(leaf_size))
    hit = cache.get(key)
    if hit is None:
        for k in [k for k in cache if k[0] != dat.dat_version]:
            cache.pop(k)                       # (a moved mesh)
        while len(cache) >= 2:
            cache.pop(next(iter(cache)))
        hit = cache[key] = kd_order(dat._dev_ptr(False), dat.cdim, n, 0, leaf_size)
    return hit


def kd_leaf_size(v):
    """Leaf sizes within a few percent of each other are the same request: multiples of 32 from 128 points up."""
    v = max(int(v), 1)
    return v if v < 128 else max(32 * int(round(v / 32.0)), 32)


class VirtualSpace(tuple):
    """Iteration space of a staged / owner-computes-rows loop over a Subset or an extruded set: the (position, layer) cells in
    position-major order.  Constant layers: the tuple (cells per entity, first layer), a cell's id is position * cells + layer
    offset.  Variable layers (``ragged``): ``prefix`` = first cell of every position, ``col`` / ``layer`` = position and absolute
    layer of every cell -- the tables the wrapper decodes a cell with where it needs the base entity or the layer argument."""

    def __new__(cls, pair, counts=None, first=None, bottom=None):
        self = super().__new__(cls, pair)
        self.ragged = counts is not None
        if self.ragged:
            counts = np.asarray(counts, dtype=np.int64)
            self.prefix = np.concatenate([[0], np.cumsum(counts)])
            self.col = np.repeat(np.arange(len(counts), dtype=np.int64), counts)
            self.layer = np.asarray(first, dtype=np.int64)[self.col] + (np.arange(int(self.prefix[-1]), dtype=np.int64) - self.prefix[self.col])
            self.bottom = np.asarray(bottom, dtype=np.int64)
            self._dev = None
        return self

    def range(self, start, end):
        """cells of the positions [start, end)"""
        if self.ragged:
            return int(self.prefix[start]), int(self.prefix[end])
        return int(start) * self[0], int(end) * self[0]

    def size(self, n):
        return self.range(0, n)[1]

    def tables_dev(self):
        """(position, layer) of every cell as two int32 device arrays (variable layers)"""
        if self._dev is None:
            self._dev = (DeviceBuffer.from_numpy(np.ascontiguousarray(self.col, dtype=np.int32)),
                         DeviceBuffer.from_numpy(np.ascontiguousarray(self.layer, dtype=np.int32)))
        return self._dev


# most entries a per-entry place table of 32-bit places can address (RowOrder.gpos; a module constant so that a test can lower it)
WIDE_PLACES = 2 ** 31 - 1


class PlanDoesNotFit(_lib.FDHipError):
    """A staged / owner-computes-rows plan exceeds the LDS or the plan builder's per-block capacity: the Parloop demotes
    the loop to the next wrapper shape (ocr -> staged -> direct) before anything is launched."""


# ---- Parloop ------------------------------------------------------------------------------------
class Parloop:
    """pyop2/parloop.py:167-540."""

    def __init__(self, global_knl: GlobalKernel, iterset: Set, arguments):
        if len(global_knl.arguments) != len(arguments):
            raise ValueError("You are trying to pass in a different number of arguments than the kernel is expecting")
        if global_knl.is_mixed:
            # Mixed arguments are split into their parts (one pointer per part, builder.py:872-916); the flattened
            # kernel's adaptor presents the concatenated packs to the local kernel (GlobalKernel.flattened)
            global_knl = global_knl.flattened()
            arguments = [q for pa in arguments
                         for q in (pa.split if isinstance(pa, (MixedDatParloopArg, MixedMatParloopArg)) else [pa])]
            assert len(global_knl.arguments) == len(arguments)
        for la, pa in zip(global_knl.local_kernel.arguments, arguments):
            if not isinstance(pa, MatParloopArg) and hasattr(pa.data, "dtype") and np.dtype(la.dtype) != pa.data.dtype:
                raise ValueError("Data types of the local kernel and the data carrier do not match")   # parloop.py:182-185
        self.global_kernel = global_knl
        self.iterset = iterset
        self.arguments = list(arguments)
        self._check_maps()
        self._check_frozen_access_modes()
        self._prepared = None
        self._lgmap_dev = {}
        # owner-computes-rows matrix assembly on a partitioned mesh: also execute the ghost entities
        # [size, total_size) (their non-owned rows are masked by the lgmaps)
        self.compute_ghost = False
        self._glob_saved = {}

    local_kernel = property(lambda self: self.global_kernel.local_kernel)
    accesses = property(lambda self: self.local_kernel.accesses)

    def _check_maps(self):          # parloop.py:472-501
        if not configuration["type_check"]:
            return
        it = self.iterset
        for pa in self.arguments:
            for m in getattr(pa, "maps", ()):
                if m.iterset.superset is not it.superset and m.iterset is not it:
                    if not (isinstance(it, ExtrudedSet) and m.iterset is it.parent):
                        raise MapValueError("Iterset of arg doesn't match ParLoop iterset.")
            if isinstance(pa, DatParloopArg) and pa.map_ is None:
                ds = pa.data.dataset.set
                base = it.parent if isinstance(it, ExtrudedSet) else it.superset
                if ds is not it and ds is not base and ds is not getattr(base, "parent", None):
                    raise MapValueError("Iterset of direct arg does not match parloop iterset")     # parloop.py:494-497

    def _check_frozen_access_modes(self):          # parloop.py:503-513
        for la, pa in zip(self.global_kernel.local_kernel.arguments, self.arguments):
            d = pa.data
            if isinstance(d, Dat) and d._halo_frozen and d._frozen_access_mode != la.access:
                raise RuntimeError("Dats with frozen halos must always be accessed with the same access mode")

    # -- preparation: compile + choose launch geometry + build plans
    def _prepare(self):
        if self._prepared is not None:
            return self._prepared
        _lib.require_gpu()
        cw = self.global_kernel.compile(getattr(self, "_forced_mode", None))
        src = cw.src
        maps = []
        for pa in self.arguments:
            for m in getattr(pa, "maps", ()):
                if all(m._base() is not q for q in maps):
                    maps.append(m._base())
        assert len(maps) == src.nmaps
        prep = {"cw": cw, "maps": maps}
        if src.mode.startswith("staged") or src.mode.startswith("ocr"):
            prep["parts"] = {}
        self._prepared = prep
        return prep

    def _staged_geometry(self, start, end):
        """Pick ents_per_block so that the block's staged rows fit the LDS budget; build plans."""
        prep = self._prepared
        key = (start, end)
        geo = prep["parts"].get(key)
        if geo is not None:
            return geo
        src = prep["cw"].src
        self._reject_negative_mat_maps()
        maps = [self._plan_map(m) for m in prep["maps"]]
        maxar = max(maps[mi].arity for mi in src.staged_maps)
        limit = configuration["lds_limit"]


        from .codegen import lds_stride, mode_variant

        def lds_bytes(plans, mplans):
            lds = 0
            nd = {mi: lds_stride(p.max_nd) for mi, p in plans.items()}
            for item in src.lds_items:
                if item[0] == "dat":
                    _, mi, c, isz, accum = item
                    lds += ((nd[mi] * c * isz) + 15) // 16 * 16
                else:
                    _, k, rm, cm, lg = item
                    mp = mplans[k]
                    lds += ((mp.max_nnz * 8) + 15) // 16 * 16 + ((nd[rm] + 1) * 4 + 15) // 16 * 16
                    if lg:
                        lds += (nd[rm] + 15) // 16 * 16 + (nd[cm] + 15) // 16 * 16
            return lds

        pstart, pend, order = start, end, None

        def build(epb, blocks):
            plans = {mi: maps[mi].plan(pstart, pend, epb, blocks, lane_threads=src.lane_threads) for mi in src.staged_maps}
            mplans = {}
            for item in src.lds_items:
                if item[0] != "dat":
                    _, k, rm, cm, lg = item
                    pa = self.arguments[k]
                    mplans[k] = pa.data.sparsity.matplan(plans[rm], plans[cm], pa.maps)
            return plans, mplans, lds_bytes(plans, mplans)

        # 1. block boundaries suggested by the data producer (mesh traversal tiles), if they fit
        blocks = None
        cand = [getattr(maps[mi], "preferred_blocks", None) for mi in src.staged_maps]
        if configuration["use_preferred_blocks"] and all(c is not None for c in cand) and all(c is cand[0] or np.array_equal(c, cand[0]) for c in cand):
            pb = np.asarray(cand[0])
            i0, i1 = np.searchsorted(pb, start), np.searchsorted(pb, end)
            if i0 < len(pb) and i1 < len(pb) and pb[i0] == start and pb[i1] == end and i1 > i0:
                bl = pb[i0:i1 + 1].astype(np.int32)
                if int(np.diff(bl).max()) * maxar <= 32768:
                    blocks = bl
        plans = None
        if blocks is not None:
            plans, mplans, lds = build(0, blocks)
            epb = int(np.diff(blocks).max())
            if lds > limit:
                plans = None
        if plans is None:
            blocks = None

            def uniform():
                # uniform blocks, halved until the staged rows fit the LDS budget
                epb = configuration["ents_per_block"]
                while epb * maxar > 32768:
                    epb //= 2
                while True:
                    plans, mplans, lds = build(epb, None)
                    if lds <= limit or epb <= 32:
                        return epb, plans, mplans, lds
                    epb = max(32, epb // 2)

            # 2. no usable producer hints: uniform blocks of the caller's order ...
            epb, plans, mplans, lds = uniform()
            # 3. ... or of a backend-derived locality order of the entities (Morton key of the entity centroids in the loop's
            #    position field, fd_locality_order; plans over the map rows gathered in that order) -- kept when its blocks
            #    touch clearly fewer distinct nodes (the staged traffic) than the caller's order does
            cand_order = self._locality_order(start, end)
            if cand_order is not None:
                n = end - start
                base_maps, base = maps, (epb, plans, mplans, lds)
                pstart, pend = 0, n
                cand = None
                for attempt in range(4):
                    okey = ("order", start, end, cand_order.serial)
                    maps = [m.derived_dev(okey, n, (lambda m=m, o=cand_order: self._gather_rows(m, o, n))) if mi in src.staged_maps else m
                            for mi, m in enumerate(base_maps)]
                    bl = cand_order.blocks.astype(np.int32)
                    if int(np.diff(bl).max()) * maxar <= 32768:
                        p_, mp_, lds_ = build(0, bl)
                        if lds_ <= limit:
                            cand = (int(np.diff(bl).max()), p_, mp_, lds_)
                            break
                    # tiles too large for the LDS budget / the plan builder: bin again with half the target (not after the last
                    # attempt: ``maps`` holds the rows gathered in THIS order, and the kernel is handed this order's table)
                    if attempt < 3:
                        cand_order = self._locality_order(start, end, target=max(cand_order.target // 2, 32))
                if cand is None:
                    cand = uniform()                    # (blocks of the derived order cut uniformly; maps gathered in cand_order)
                touched = lambda pl: sum(p.list_len for p in pl.values())          # noqa: E731
                if touched(cand[1]) < 0.9 * touched(base[1]):
                    order = cand_order
                    epb, plans, mplans, lds = cand
                else:
                    maps, pstart, pend = base_maps, start, end
                    epb, plans, mplans, lds = base
        if lds > 160 * 1024:
            raise PlanDoesNotFit("staged wrapper does not fit LDS even at 32 entities per block")
        kb = 2 if any(mp.kbytes == 2 for mp in mplans.values()) else 1
        # maps whose staged rows see no reuse inside any block (every node list entry belongs to one map entry: the cell loop of a
        # discontinuous space): LDS staging buys nothing there -- no gather is saved, no reduction happens -- and costs a dependent
        # load chain, LDS capacity and a flush phase.  Their Dat arguments go straight from / to global memory ("_d" variants); maps
        # a matrix plan hangs on, and the last staged map, stay
        direct_maps = []
        if configuration["staged_direct_noreuse"] and not mplans and self._virtual() is None:        # (plain sets: [start, end) are entity ids)
            n_ent = pend - pstart
            for mi in src.staged_maps:
                if plans[mi].list_len == n_ent * maps[mi].arity and len(direct_maps) + 1 < len(src.staged_maps) \
                        and order is None and self._affine_map(prep["maps"][mi], start, end):
                    direct_maps.append(mi)
        kept = [mi for mi in src.staged_maps if mi not in direct_maps]
        variant = mode_variant("stagedo" if order is not None else "staged", kb, [plans[mi].max_nd for mi in kept], direct_maps=direct_maps)
        # geometry-specific variant (16-bit matrix offsets, compile-time LDS strides, entity order, direct maps); the argument list
        # is built from the VARIANT's layout (_arglist)
        cw = prep["cw"] if variant == src.mode else self.global_kernel.compile(variant)
        if direct_maps:
            # ("_d" variants run in entity order: the kept plans are rebuilt without the lane order)
            plans = {mi: maps[mi].plan(pstart, pend, epb if blocks is None else 0, blocks, lane_threads=cw.src.lane_threads) for mi in kept}
            nd_ = {mi: lds_stride(p.max_nd) for mi, p in plans.items()}
            lds = sum(((nd_[item[1]] * item[2] * item[3]) + 15) // 16 * 16 for item in cw.src.lds_items if item[0] == "dat")
        else:
            assert [d for d in cw.src.layout if d[0] != "order"] == [d for d in src.layout if d[0] != "order"]
        geo = {"epb": epb, "plans": plans, "mplans": mplans, "lds": lds, "cw": cw, "order": order, "range": (pstart, pend)}
        if configuration["debug"]:
            import sys
            print(f"[fdhip] {self.global_kernel.name} staged variant {variant}", file=sys.stderr)
            for mi, pl in plans.items():
                print(f"[fdhip] {self.global_kernel.name} [{start},{end}) epb={epb} map{mi}: blocks={pl.nblocks} "
                      f"max_nd={pl.max_nd} list_len={pl.list_len} lds={lds}", file=sys.stderr)
            for k, mp in mplans.items():
                print(f"[fdhip]   matplan arg{k}: block-nz total={mp.total} max_nnz={mp.max_nnz} max_rowlen={mp.max_rowlen} "
                      f"kbytes={mp.kbytes} exclusive={mp.n_exclusive} zero_list={mp.n_zero}", file=sys.stderr)
        prep["parts"][key] = geo
        return geo

    @staticmethod
    def _affine_map(m, start, end):
        """map[e][i] == arity*e + i for every entity of [start, end): the cell-node map of a discontinuous space (Firedrake numbers
        DG nodes cell by cell).  Checked once per map and range on the host values; maps known only by a device pointer are not."""
        b = m._base()
        cache = b.__dict__.setdefault("_affine_ranges", {})
        hit = cache.get((start, end))
        if hit is None:
            v = getattr(b, "_values", None)
            hit = False
            if isinstance(v, np.ndarray) and v.ndim == 2 and end <= len(v):
                ar = v.shape[1]
                hit = bool(np.array_equal(v[start:end], (np.arange(start, end, dtype=np.int64)[:, None] * ar + np.arange(ar)[None, :])))
            cache[(start, end)] = hit
        return hit

    def _reject_negative_mat_maps(self):
        """MatSetValuesLocal ignores negative indices (builder.py:573-625), so the maps of a Mat argument may hold -1 entries
        (composed maps with undefined intermediate entries).  The direct and the row-sliced wrappers honour that; block
        localisation plans and whole-entity instance tables do not -- such loops are demoted (checked on the host values,
        once per map; maps known only by a device pointer are trusted)."""
        for pa in self.arguments:
            if not isinstance(pa, MatParloopArg):
                continue
            for m in pa.maps:
                b = m._base()
                neg = b.__dict__.get("_has_negative")
                if neg is None:
                    v = getattr(b, "_values", None)
                    neg = bool(isinstance(v, np.ndarray) and v.size and int(v.min()) < 0)
                    b.__dict__["_has_negative"] = neg
                if neg:
                    raise PlanDoesNotFit("negative entries in the map of a Mat argument")

    # -- virtual iteration space of staged loops over subsets / extruded sets
    def _virtual(self, staged=None):
        """The virtual iteration space the staged / owner-computes-rows wrappers run over when the set is a Subset or extruded: a
        ``VirtualSpace`` -- (cells iterated per entity, first layer) for constant layers (usable as that tuple), a ragged space with
        per-position tables for variable layers (set.py:326-337) -- else None."""
        gk = self.global_kernel
        if staged is None:
            mode = self._prepared["cw"].src.mode
            staged = mode.startswith("staged") or mode.startswith("ocr")
        if not (gk._extruded or gk._subset) or not staged:
            return None
        if not gk._extruded:
            return VirtualSpace((1, 0))
        from .op2types import ON_BOTTOM, ON_INTERIOR_FACETS, ON_TOP
        reg = gk._iteration_region
        it = self.iterset
        if not gk._constant_layers:
            hit = self.__dict__.get("_virtual_var")
            if hit is None:
                # variable layers: entity e iterates the cells [bottom_e, top_e - 1) of its column (ALL), its first or its last one;
                # positions of a Subset index the superset's rows (the layers array belongs to the superset, builder.py:744-752)
                la = np.asarray(it.layers_array, dtype=np.int64)
                if isinstance(it, Subset):
                    la = la[np.asarray(it.indices, dtype=np.int64)]
                ncell = np.maximum(la[:, 1] - 1 - la[:, 0], 0)
                if reg == ON_BOTTOM:                       # (one trip per entity whatever its column holds: builder.py:790-812)
                    cnt, first = np.ones_like(ncell), la[:, 0]
                elif reg == ON_TOP:
                    cnt, first = np.ones_like(ncell), la[:, 1] - 2
                elif reg == ON_INTERIOR_FACETS:
                    # one trip per pair of stacked cells of the entity's own column: layers [bottom_e, top_e - 2) (builder.py:806-809
                    # with per-entity bounds, :754-776); a one-cell column has none
                    cnt, first = np.maximum(ncell - 1, 0), la[:, 0]
                else:
                    cnt, first = ncell, la[:, 0]
                hit = self.__dict__["_virtual_var"] = VirtualSpace((0, 0), counts=cnt, first=first, bottom=la[:, 0])
            return hit
        bottom, top = (int(v) for v in it.layers_array[0])
        if reg == ON_BOTTOM:
            return VirtualSpace((1, bottom))
        if reg == ON_TOP:
            return VirtualSpace((1, top - 2))
        if reg == ON_INTERIOR_FACETS:                  # one trip per pair of stacked cells (builder.py:806-809; periodic: top/bottom too)
            return VirtualSpace((top - 1 - bottom if gk._extruded_periodic else max(top - 2 - bottom, 0), bottom))
        return VirtualSpace((top - 1 - bottom, bottom))

    def _plan_map(self, m, staged=None):
        """The Map the block-localisation plans of this loop are built on: ``m`` itself, or its derived map over the
        virtual iteration space (subset position x layer)."""
        v = self._virtual(staged)
        if v is None:
            return m
        nlit, llo = v
        it = self.iterset
        sub = it.indices if isinstance(it, Subset) else None
        if v.ragged:
            key = ("virtual-var", None if sub is None else id(it), int(self.global_kernel._iteration_region))

            def build_var():
                rows = np.asarray(m.values_with_halo)
                if sub is not None:
                    rows = rows[sub]
                off = np.asarray(m.offset, dtype=np.int64)
                rel = v.layer - v.bottom[v.col]                    # cells above the entity's own bottom (builder.py:94-124)
                below = rows[v.col] + off[None, :] * rel[:, None]
                from .op2types import ON_INTERIOR_FACETS as facets_
                if self.global_kernel._iteration_region == facets_:
                    # an interior facet sees the cell below and the cell above it: a row of 2 x arity nodes (f = 0, 1)
                    return np.concatenate([below, below + off[None, :]], axis=1).astype(rows.dtype)
                return below.astype(rows.dtype)
            return m.derived(key, build_var)
        bottom = int(it.layers_array[0][0]) if self.global_kernel._extruded else 0
        periodic = bool(self.global_kernel._extruded and self.global_kernel._extruded_periodic)
        from .op2types import ON_INTERIOR_FACETS
        facets = bool(self.global_kernel._extruded and self.global_kernel._iteration_region == ON_INTERIOR_FACETS)
        key = ("virtual", None if sub is None else id(it), nlit, llo, periodic, facets)

        def build():
            rows = np.asarray(m.values_with_halo)
            if sub is not None:
                rows = rows[sub]
            if self.global_kernel._extruded:
                off = np.asarray(m.offset, dtype=np.int64)
                lay = np.arange(llo - bottom, llo - bottom + nlit, dtype=np.int64)
                def cells(lay_):
                    if periodic:
                        # builder.py:101-123: the layer offset wraps around the column's nl cell layers -- entry i of layer l sits
                        # offset_i * ((l + quotient_i) mod nl - quotient_i mod nl) above its bottom value (quotient 0 without one)
                        nl = int(it.layers_array[0][1]) - 1 - bottom
                        quot = np.zeros(m.arity, dtype=np.int64) if m.offset_quotient is None else np.asarray(m.offset_quotient, dtype=np.int64)
                        rel = (lay_[:, None] + quot[None, :]) % nl - (quot % nl)[None, :]
                        return rows[:, None, :] + off[None, None, :] * rel[None, :, :]
                    return rows[:, None, :] + off[None, None, :] * lay_[None, :, None]
                if facets:       # an interior facet sees the cell below and the cell above it: a row of 2 x arity nodes (builder.py:94-124, f = 0, 1)
                    rows = np.concatenate([cells(lay), cells(lay + 1)], axis=2).reshape(-1, 2 * m.arity)
                else:
                    rows = cells(lay).reshape(-1, m.arity)
            return rows
        return m.derived(key, build)

    # -- backend-derived locality (un-hinted maps)
    def _position_arg(self):
        """The loop's position field: the first fp64 Dat of 2 or 3 components READ through a map -- the coordinate argument
        of a TSFC kernel (tsfc/kernel_interface/firedrake_loopy.py:432-522).  None if the loop has none."""
        for pa, acc in zip(self.arguments, self.accesses):
            if (isinstance(pa, DatParloopArg) and pa.map_ is not None and acc == READ and not isinstance(pa.map_, PermutedMap)
                    and np.dtype(pa.data.dtype) == np.float64 and pa.data.cdim in (2, 3) and getattr(pa.data, "index", None) is None):
                return pa
        return None

    def _locality_order(self, start, end, virtual=False, target=None):
        """``LocalityOrder`` of the entities of [start, end) -- grouped around the k-d leaves of the position field's nodes,
        ~``target`` entities per group (fd_kd_order + fd_group_entities): entity list + block boundaries -- or None when the loop
        keeps the caller's order (switched off, tiny range, no position field).  ``virtual``: [start, end) are positions of
        the virtual iteration space (owner-computes-rows loops over subsets / extruded sets, which need the order only to
        derive a row order); staged loops over virtual spaces keep the caller's order."""
        n = end - start
        if not configuration["locality_order"] or n < configuration["locality_min_entities"] or (self._virtual() is not None and not virtual):
            return None
        pa = self._position_arg()
        if pa is None or not getattr(pa.data.dataset.set, "total_size", None):
            return None                        # (a borrowed carrier whose node count is unknown: function-level seam, unregistered map)
        # (small loops: leaves sized to give the device ~4 blocks per CU -- 8192 cells in leaves of 1536 are 6 workgroups on 256 CUs)
        target = int(target or small_loop_leaf(end - start, int(configuration["locality_tile_entities"]), 256))
        pmap = self._plan_map(pa.map_._base(), staged=True) if virtual else pa.map_._base()
        cache = pmap.__dict__.setdefault("_locality_orders", {})
        key = (start, end, id(pa.data), pa.data.dat_version, target)
        lo = cache.get(key)
        if lo is None:
            # k-d leaves of the position field's NODES, sized so that the entities around one leaf number ~target
            pdim, nnodes = pa.data.cdim, pa.data.dataset.set.total_size
            leaf_nodes = kd_leaf_size(round(target * nnodes / max(n, 1)))
            norder, nstarts = kd_order_of(pa.data, nnodes, leaf_nodes)
            nleaves = len(nstarts) - 1
            label_d = DeviceBuffer(max(nnodes, 1) * 4)
            ns32 = np.ascontiguousarray(nstarts, dtype=np.int32)
            _lib.call("fd_leaf_labels", norder.ptr, nnodes, ns32.ctypes.data, nleaves, label_d.ptr, None)
            buf = DeviceBuffer(n * 4)
            counts = np.zeros(nleaves, dtype=np.int32)
            # (the DERIVED map's arity: interior-facet rows of a virtual space hold both stacked cells)
            _lib.call("fd_group_entities", pmap._dev_values(), pmap.arity, int(start), int(end), label_d.ptr, nnodes, nleaves,
                      buf.ptr, counts.ctypes.data, None)
            del label_d
            starts = LocalityOrder.cut(counts.astype(np.int64), target)
            for k in [k for k in cache if k[:3] == key[:3] and k[3] != key[3]]:
                cache.pop(k)                   # orders of an earlier state of the position field (a moved mesh)
            lo = LocalityOrder(buf, starts, target)
            cache[key] = lo
        return lo

    @staticmethod
    def _gather_rows(m, order, n):
        buf = DeviceBuffer(max(n, 1) * m.arity * 4)
        _lib.call("fd_gather_rows", m._dev_values(), m.arity, order.ptr, n, buf.ptr, None)
        return buf

    # -- argument list in the kernel's parameter order
    def _arglist(self, start, end):
        prep = self._prepare()
        src = prep["cw"].src
        geo = self._staged_geometry(start, end) if src.mode.startswith("staged") else None
        if geo is not None:
            src = geo["cw"].src
        out = [g(self) for g in self._getters_of(geo if geo is not None else prep, src, lambda: self._arg_getters(prep, src, geo))]
        if _CHECK_ARGLISTS:               # (tests: the branch-per-entry statement of the same list; version counters as one call leaves them)
            vers = [(pa.data, pa.data.dat_version) for pa in self.arguments if hasattr(pa.data, "dat_version")]
            ref = self._args_chain(prep, src, geo)
            _restore_versions(vers)
            _same_arglists(src.layout, out, ref)
        return out, geo

    def _getters_of(self, holder, wrapper, build):
        """The argument getters of (geometry or prepared state ``holder``, wrapper): kept on the Parloop, a few entries, identity-checked."""
        store = self.__dict__.get("_getter_store")
        if store is None:
            store = self._getter_store = {}
        ent = store.get(id(holder))
        if ent is None or ent[0] is not holder or ent[1] is not wrapper:
            while len(store) >= 8:
                store.pop(next(iter(store)))
            ent = store[id(holder)] = (holder, wrapper, build())
        return ent[2]

    def _args_chain(self, prep, src, geo):
        """The argument list of a staged / direct / tensor-product launch, one branch per layout entry (what ``_arg_getters``
        pre-dispatches; tests run both and compare)."""
        out = []
        for desc in src.layout:
            kind = desc[0]
            if kind in ("virt_col", "virt_layer"):
                out.append(self._virtual(staged=True).tables_dev()[0 if kind == "virt_col" else 1].ptr)
            elif kind == "layers":
                out.append(self.iterset._layers_dev())
            elif kind == "subset":
                out.append(self.iterset._indices_dev())
            elif kind == "arg":
                pa = self.arguments[desc[1]]
                acc = self.accesses[desc[1]]
                if isinstance(pa, MatParloopArg):
                    pa.data.dat_version += 1
                    out.append(pa.data._values_dev().ptr)
                else:
                    out.append(pa.data._dev_ptr(write=acc != READ))
            elif kind == "map":
                out.append(prep["maps"][desc[1]]._dev_values())
            elif kind == "bstart":
                out.append(next(iter(geo["plans"].values())).bstart if geo else 0)
            elif kind == "order":
                out.append(geo["order"].ptr)
            elif kind == "plan_blkoff":
                out.append(geo["plans"][desc[1]].blkoff)
            elif kind == "plan_list":
                out.append(geo["plans"][desc[1]].list)
            elif kind == "plan_lmap":
                out.append(geo["plans"][desc[1]].lmap)
            elif kind == "plan_maxnd":
                out.append(geo["plans"][desc[1]].max_nd)
            elif kind == "plan_copy":
                out.append(self._plan_copy(geo, desc[1], geo["plans"][desc[2]]))
            elif kind == "matplan_off":
                out.append(geo["mplans"][desc[1]].mb_off)
            elif kind == "matplan_gpos":
                out.append(geo["mplans"][desc[1]].gpos)
            elif kind == "matplan_lrp":
                out.append(geo["mplans"][desc[1]].lrp)
            elif kind == "matplan_kidx":
                out.append(geo["mplans"][desc[1]].kidx)
            elif kind == "matplan_maxnnz":
                out.append(geo["mplans"][desc[1]].max_nnz)
            elif kind == "matplan_flags":
                out.append(0)
            elif kind == "mat_table":
                pa = self.arguments[desc[1]]
                out.append(pa.data.sparsity.elem_table(*pa.maps).ptr)
            elif kind == "mat_node_rowptr":
                pa = self.arguments[desc[1]]
                pa.data.sparsity._build()
                out.append(pa.data.sparsity._node_rowptr.ptr)
            elif kind == "mat_rowptr":
                pa = self.arguments[desc[1]]
                pa.data.sparsity._build()
                out.append(pa.data.sparsity._rowptr.ptr)
            elif kind == "mat_colidx":
                pa = self.arguments[desc[1]]
                pa.data.sparsity._build()
                out.append(pa.data.sparsity._colidx.ptr)
            elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
                pa = self.arguments[desc[1]]
                lg = pa.lgmaps[0 if kind == "mat_row_lgmap" else 1]
                out.append(self._lgmap(lg))
            elif kind == "tp_offtab":
                out.append(self._tp_offtab(desc[1]).ptr)
            elif kind == "tp_tables":
                out.append(self._tp_tables().ptr)
            else:
                raise AssertionError(kind)
        return out

    def _arg_getters(self, prep, src, geo):
        """One callable ``f(parloop)`` per layout entry, built once per (geometry, wrapper source): see ``_ocr_arg_getters``."""
        args, accs = self.arguments, self.accesses
        g = []
        for desc in src.layout:
            kind = desc[0]
            if kind in ("virt_col", "virt_layer"):
                g.append(lambda pl, w=0 if kind == "virt_col" else 1: pl._virtual(staged=True).tables_dev()[w].ptr)
            elif kind == "layers":
                g.append(lambda pl: pl.iterset._layers_dev())
            elif kind == "subset":
                g.append(lambda pl: pl.iterset._indices_dev())
            elif kind == "arg":
                pa = args[desc[1]]
                if isinstance(pa, MatParloopArg):
                    def mat_arg(pl, pa=pa):
                        mat = pa.data                        # (read at every call: see _ocr_arg_getters)
                        mat.dat_version += 1
                        return mat._values_dev().ptr
                    g.append(mat_arg)
                else:
                    g.append(lambda pl, pa=pa, w=accs[desc[1]] != READ: pa.data._dev_ptr(write=w))
            elif kind == "map":
                g.append(lambda pl, k=desc[1]: prep["maps"][k]._dev_values())
            elif kind == "bstart":
                g.append((lambda pl: next(iter(geo["plans"].values())).bstart) if geo else (lambda pl: 0))
            elif kind == "order":
                g.append(lambda pl: geo["order"].ptr)
            elif kind == "plan_blkoff":
                g.append(lambda pl, k=desc[1]: geo["plans"][k].blkoff)
            elif kind == "plan_list":
                g.append(lambda pl, k=desc[1]: geo["plans"][k].list)
            elif kind == "plan_lmap":
                g.append(lambda pl, k=desc[1]: geo["plans"][k].lmap)
            elif kind == "plan_maxnd":
                g.append(lambda pl, k=desc[1]: geo["plans"][k].max_nd)
            elif kind == "plan_copy":
                g.append(lambda pl, k=desc[1], m=desc[2]: pl._plan_copy(geo, k, geo["plans"][m]))
            elif kind == "matplan_off":
                g.append(lambda pl, k=desc[1]: geo["mplans"][k].mb_off)
            elif kind == "matplan_gpos":
                g.append(lambda pl, k=desc[1]: geo["mplans"][k].gpos)
            elif kind == "matplan_lrp":
                g.append(lambda pl, k=desc[1]: geo["mplans"][k].lrp)
            elif kind == "matplan_kidx":
                g.append(lambda pl, k=desc[1]: geo["mplans"][k].kidx)
            elif kind == "matplan_maxnnz":
                g.append(lambda pl, k=desc[1]: geo["mplans"][k].max_nnz)
            elif kind == "matplan_flags":
                g.append(lambda pl: 0)
            elif kind == "mat_table":
                g.append(lambda pl, pa=args[desc[1]]: pa.data.sparsity.elem_table(*pa.maps).ptr)
            elif kind in ("mat_node_rowptr", "mat_rowptr", "mat_colidx"):
                def sparsity_array(pl, pa=args[desc[1]], name={"mat_node_rowptr": "_node_rowptr", "mat_rowptr": "_rowptr", "mat_colidx": "_colidx"}[kind]):
                    sp = pa.data.sparsity
                    sp._build()
                    return getattr(sp, name).ptr
                g.append(sparsity_array)
            elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
                g.append(lambda pl, pa=args[desc[1]], w=0 if kind == "mat_row_lgmap" else 1: pl._lgmap(pa.lgmaps[w]))
            elif kind == "tp_offtab":
                g.append(lambda pl, k=desc[1]: pl._tp_offtab(k).ptr)
            elif kind == "tp_tables":
                g.append(lambda pl: pl._tp_tables().ptr)
            else:
                raise AssertionError(kind)
        return g

    # -- tensor-product wrappers (csrc/fd_tensor.h) ------------------------------------------------------------------
    def zero_ahead(self):
        """Perform the zeroing this loop would do at its launch now (benchmarks bracket the kernel without it)."""
        for pa in self.arguments:
            if isinstance(pa, MatParloopArg):
                pa.data._values_dev()

    def _tp_tables(self):
        """1-D tabulation of the element (values and derivatives of the CG_k GLL basis at the Gauss points, points,
        weights) on the device -- what FInAT hands TSFC as constant tables (tsfc/fem.py:711-735)."""
        t = self._prepared.get("tp_tables")
        if t is None:
            from .tensor import gll_gauss_tables
            tp = self.global_kernel.local_kernel.tp
            L, DL, qp, qw = gll_gauss_tables(tp["degree"], tp["nq"])
            # the action template keeps half of each table (nodes and points symmetric about 1/2: fd_tensor.h, hex_qk_action)
            if not (np.allclose(L, L[::-1, ::-1], rtol=0, atol=1e-13 * np.abs(L).max())
                    and np.allclose(DL, -DL[::-1, ::-1], rtol=0, atol=1e-13 * np.abs(DL).max())):
                raise ValueError("tensor-product wrappers need 1-D tables symmetric about the midpoint of the interval")
            L, DL = 0.5 * (L + L[::-1, ::-1]), 0.5 * (DL - DL[::-1, ::-1])
            t = DeviceBuffer.from_numpy(np.concatenate([L.ravel(), DL.ravel(), qp, qw]))
            self._prepared["tp_tables"] = t
        return t

    def _tp_offtab(self, k):
        """uint16 [ncol][3][nd*nd]: position of every element-matrix entry inside its CSR row for the bottom, an interior
        and the top cell of every column.  The interior cells of an extruded column are translates of each other (node =
        map + offset*layer for rows and columns alike), so one table serves them all: 3 x nd^2 x 2 B per COLUMN instead
        of nd^2 x 4 B per cell."""
        tabs = self._prepared.setdefault("tp_offtab", {})
        t = tabs.get(k)
        if t is None:
            pa = self.arguments[k]
            m = pa.maps[0]._base()
            sp = pa.data.sparsity
            sp._build()
            nl = self.iterset.layers - 1
            ncol = m.values_with_halo.shape[0]
            nd = m.arity
            off = np.asarray(m.offset, dtype=np.int32)
            lay = np.array([0, min(1, nl - 1), nl - 1], dtype=np.int32)
            rows = (np.asarray(m.values_with_halo, dtype=np.int32)[:, None, :] + off[None, None, :] * lay[None, :, None])
            rows_d = DeviceBuffer.from_numpy(np.ascontiguousarray(rows.reshape(ncol * 3, nd)))
            t = DeviceBuffer(ncol * 3 * nd * nd * 2)
            _lib.call("fd_csr_elem_row_offsets", sp._node_rowptr.ptr, sp._node_colidx.ptr, rows_d.ptr, rows_d.ptr, ncol * 3, nd, nd,
                      2, t.ptr, None)
            tabs[k] = t
        return t

    def _lgmap(self, lg):
        if hasattr(lg, "_fd_dev_ptr"):           # already on the device (bridge.DeviceMat.set_lgmaps)
            return lg._fd_dev_ptr
        a = np.ascontiguousarray(lg, dtype=np.int32)
        key = id(lg)
        d = self._lgmap_dev.get(key)
        if d is None or d[0] is not lg:
            d = (lg, DeviceBuffer.from_numpy(a))
            self._lgmap_dev[key] = d
        return d[1].ptr

    # -- execution (parloop.py:243-260)
    def __call__(self):
        self.compute()

    def _parts(self):
        parts = [self.iterset.core_part, self.iterset.owned_part]
        if self.compute_ghost:
            parts.append((self.iterset.size, self.iterset.total_size - self.iterset.size))
        return [p for p in parts if p[1] > 0]

    def _ensure_geometry(self):
        """Build every plan this loop needs BEFORE anything is launched; a plan that does not fit demotes the loop to the
        next wrapper shape (owner-computes-rows -> staged matrix plans -> direct) instead of failing half way."""
        for _ in range(3):
            mode = self._prepare()["cw"].src.mode
            try:
                if mode.startswith("ocr"):
                    self._ocr_geometry()
                elif mode.startswith("staged"):
                    v = self._virtual()
                    for off, size in self._parts():
                        rng_ = v.range(off, off + size) if v else (off, off + size)
                        if rng_[1] > rng_[0]:
                            self._staged_geometry(*rng_)
                return
            except PlanDoesNotFit as exc:
                from .codegen import _ocr_shape, staged_eligible
                nxt = "staged" if (mode.startswith("ocr") and staged_eligible(self.global_kernel)) else "direct"
                if getattr(exc, "retry_mode", None) == "ocrs" and not mode.startswith("ocrs") \
                        and _ocr_shape(self.global_kernel, mats_on_virtual=True, allow_unroll=True) is not None:
                    nxt = "ocrs"
                if configuration["debug"]:
                    import sys
                    print(f"[fdhip] {self.global_kernel.name}: {exc}; falling back to the {nxt} wrapper", file=sys.stderr)
                self._forced_mode, self._prepared = nxt, None

    def _compute_event(self):            # pyop2/parloop.py:221-222
        if not configuration["trace"]:
            return _NO_EVENT
        from .profiling import timed_region
        return timed_region(self._event_name)

    @property
    def _event_name(self):
        return f"Parloop_{getattr(self.iterset, 'name', None) or 'set'}_{self.global_kernel.name}"

    @property
    def num_flops(self):                 # pyop2/global_kernel.py:377-387: flops of ONE entity (x the layers a column iterates)
        per = self.global_kernel.local_kernel.flop_count or 0
        return per * (self._nlayers_iterated() if self.iterset._extruded else 1)

    def _log_flops(self, size):          # PETSc.Log.logFlops(part.size * self.num_flops), pyop2/parloop.py:231
        if configuration["trace"]:
            from .profiling import log_flops
            log_flops(self._event_name, size, size * self.num_flops)

    def compute(self):
        self._halos_now = self._halos_present()
        try:
            self._compute_all()
        finally:
            self._halos_now = None

    def _compute_all(self):
        self._zero_global_temporaries()
        self._ensure_geometry()
        if self._prepare()["cw"].src.mode.startswith("ocr"):
            # owner-computes-rows: one launch over the row blocks; entities are visited through the instance
            # lists, so there is no core/owned split to overlap the halo exchange with
            self.global_to_local_begin()
            self.global_to_local_end()
            with self._compute_event():
                self._log_flops(self.iterset.total_size if self.compute_ghost else self.iterset.size)
                self._compute_ocr()
            self.reduction_begin()
            self.reduction_end()
            return
        self.global_to_local_begin()
        self._compute(self.iterset.core_part)
        self.global_to_local_end()
        self._compute(self.iterset.owned_part)
        if self.compute_ghost:
            self._compute((self.iterset.size, self.iterset.total_size - self.iterset.size))
        self.reduction_begin()
        self.local_to_global_begin()
        self.reduction_end()
        self.local_to_global_end()
        self.finalize_assembly()
        for pa, acc in zip(self.arguments, self.accesses):
            if acc != READ and hasattr(pa.data, "_after_device_write"):
                pa.data._after_device_write()      # live writable host views stay current (op2types._Mirrored)

    def _compute(self, part):
        offset, size = part
        if size <= 0:
            return
        with self._compute_event():
            self._log_flops(size)
            self._compute_part(offset, size)

    def _compute_part(self, offset, size):
        start, end = offset, offset + size
        self._prepare()
        v = self._virtual()
        if v is not None:
            start, end = v.range(start, end)            # positions in the virtual (entity x layer) space
            if end <= start:
                return                                  # (no cell to visit: one-layer columns under ON_INTERIOR_FACETS, empty columns)
        args, geo = self._arglist(start, end)
        cw = geo["cw"] if geo is not None else self._prepared["cw"]
        src = cw.src
        threads = src.block_threads
        if src.mode.startswith("staged"):
            nb = next(iter(geo["plans"].values())).nblocks
            ps, pe = geo["range"]               # plan coordinates: [start, end), or [0, n) of the derived entity order
            cw.launch(ps, pe, args, block_threads=threads, ents_per_block=geo["epb"], nblocks=nb, lds_bytes=geo["lds"])
        elif src.mode.startswith("tp_"):
            # matrix: one wavefront per 16-row panel of the padded element matrix (Q4: two workgroups of four per cell); action:
            # one workgroup per tp["action_cells"] cells of the (column, layer) space (codegen.tensor_geometry)
            ncell = size * (self.iterset.layers - 1)
            from .codegen import tensor_matrix_groups
            vd = int(self.global_kernel.local_kernel.tp.get("vdim", 1))         # (Q_k)^D: D^2 scalar blocks per element matrix
            nb = ncell * tensor_matrix_groups(src.tp, vd) if src.mode == "tp_matrix" else -(-ncell // src.tp["action_cells"])
            cw.launch(start, end, args, block_threads=threads, ents_per_block=1, nblocks=nb)
        else:
            total = size
            if self.iterset._extruded and src.layer_parallel:
                total *= self._nlayers_iterated()
            nblocks = max(1, (total + threads - 1) // threads)
            cw.launch(start, end, args, block_threads=threads, ents_per_block=threads, nblocks=nblocks)

    # -- owner-computes-rows ---------------------------------------------------------------------------
    def _ocr_geometry(self, start=0, end=None):
        """Owner-computes-rows plan over the entities [start, end) (default: every entity the loop executes)."""
        prep = self._prepared
        if end is None:
            end = self.iterset.total_size if self.compute_ghost else self.iterset.size
        gkey = ("ocr", int(start), int(end))
        geo = prep["parts"].get(gkey)
        if geo is not None:
            return geo
        from .op2types import OcrPlan
        from .codegen import lds_stride, mode_variant
        src = prep["cw"].src
        if src.mode.startswith("ocrs"):
            return self._ocrs_geometry(start, end, gkey)
        start0, end0, kd_rows = start, end, None
        self._reject_negative_mat_maps()
        # subsets / extruded sets: every map is replaced by its derived map over the virtual (position x layer) space
        maps = [self._plan_map(m._base(), staged=True) for m in prep["maps"]]
        v = self._virtual(staged=True)
        if v is not None:
            start, end = v.range(start, end)
        (k, pa), = [(k, a) for k, a in enumerate(self.arguments) if isinstance(a, MatParloopArg)]
        rmap, cmap = (self._plan_map(m._base(), staged=True) for m in pa.maps)
        sp = pa.data.sparsity
        sp._build()
        nrows = rmap.toset.size                                   # owned rows only
        rp = sp._node_rowptr_host()
        limit = src.ocr_lds_limit or configuration["lds_limit"]
        hint = getattr(rmap._base(), "preferred_node_blocks", None)
        row_order = None
        if hint is not None and configuration["use_preferred_blocks"]:
            rb = np.asarray(hint, dtype=np.int64)
            rb = np.unique(np.concatenate([rb[rb < nrows], [0, nrows]]))
        else:
            # no producer hints: a backend-derived row order when the loop has a position field, else the caller's row order in
            # greedy ranges of ~cap CSR entries
            from .op2types import RowOrder
            pos_ = self._position_arg()
            prp = rp[:nrows + 1]
            cap = configuration["ocr_nnz_per_block"]
            rb = None
            usable = (configuration["locality_order"] and pos_ is not None and (end - start) >= configuration["locality_min_entities"]
                      and bool(getattr(pos_.data.dataset.set, "total_size", None)))
            if usable and v is None and pos_.map_._base() is pa.maps[0]._base():
                # the rows ARE the nodes of the position field: partition them by their own coordinates into k-d leaves of
                # equal row count -- boxes of rows whose accumulators fill the LDS budget exactly
                cap = configuration["ocr_nnz_per_block_ordered"]
                rows_per_block = max(cap // max(int(np.ceil(rp[nrows] / max(nrows, 1))), 1), 1)
                rows_per_block = small_loop_leaf(nrows, rows_per_block, 32)     # (small loops: ~4 row blocks per CU)
                if kd_leaf_size(rows_per_block) <= rows_per_block:
                    rows_per_block = kd_leaf_size(rows_per_block)           # (never above the LDS budget the cap stands for)
                rows_per_block = kd_rows = prep.get("ocr_leaf_rows", {}).get(gkey, rows_per_block)    # (a leaf shrunk below: see the end)
                plist, rb = kd_order_of(pos_.data, nrows, rows_per_block)
                row_order = RowOrder.from_plist(plist, nrows, rp, rowptr_dev=sp._node_rowptr.ptr)
            elif usable:
                # rows of another space: first touch under the k-d order of the entities, cut where the entity leaf changes
                # (leaves hold equal numbers of entities, not of rows: 10 % slack before a block is halved)
                order = self._locality_order(start, end, virtual=v is not None)
                if order is not None:
                    row_order = RowOrder(rmap, order, end - start, nrows, rp, rowptr_dev=sp._node_rowptr.ptr)
                    cap = configuration["ocr_nnz_per_block_ordered"]
                    rb = row_order.tile_cuts(order.blocks, cap + cap // 10)
            if row_order is not None:
                if int(row_order.prowptr_host[-1]) > WIDE_PLACES:
                    # the whole-entity flush of a derived row order streams 32-bit places (RowOrder.gpos): refused HERE, where
                    # _ensure_geometry still can move the loop to another shape, not at launch time -- and the shape that serves
                    # such patterns is the row-sliced one (run-coded flush: 64-bit row starts, one byte per entry), whatever the
                    # size of the element matrix
                    exc = PlanDoesNotFit("whole-entity owner-computes-rows under a derived row order holds 32-bit places: "
                                         "the pattern has 2^31 entries or more")
                    exc.retry_mode = "ocrs"
                    raise exc
                prp = row_order.prowptr_host
            if rb is None:
                targets = np.arange(0, int(prp[nrows]) + cap, cap)
                rb = np.unique(np.concatenate([np.searchsorted(prp[:nrows + 1], targets, side="left"), [0, nrows]]))
                rb = rb[rb <= nrows]
        staged = {mi: maps[mi] for mi in src.staged_maps}
        maxar = max(m.arity for m in staged.values())

        def lds_bytes(op):
            lds = 0
            nd = {mi: lds_stride(p.max_nd, ocr=True) for mi, p in op.plans.items()}
            for item in src.lds_items:
                if item[0] == "dat":
                    _, mi, c, isz, accum = item
                    lds += ((nd[mi] * c * isz) + 15) // 16 * 16
                else:
                    _, kk, rm, cmi, lg = item
                    lds += ((op.max_nnz * 8) + 15) // 16 * 16 + (nd[rm] * 4 + 15) // 16 * 16
                    if cmi != rm:
                        lds += (nd[cmi] + 15) // 16 * 16
            return lds

        for attempt in range(14):
            # split row blocks until the LDS rows and the instance lists fit
            try:
                op = OcrPlan(sp, rmap, cmap, staged, start, end, rb, lane_threads=src.lane_threads, row_order=row_order)
            except _lib.FDHipError as exc:
                if "map entries" not in str(exc):
                    raise
                d = np.diff(rb)                                    # an instance list is too long: halve every wide block
                rb = np.unique(np.concatenate([rb, (rb[:-1] + d // 2)[d > 1]]))
                continue
            lds = lds_bytes(op)
            if lds <= limit and op.max_inst * maxar <= 32768:
                break
            d = np.diff(rb)
            nn = np.diff((row_order.prowptr_host if row_order is not None else rp)[rb])
            ni = np.diff(op.inst_off_host)
            big = (nn > 0.7 * nn.max()) | (ni * maxar > 32768) if lds > limit else (ni * maxar > 32768)
            big &= d > 1
            if not big.any():
                break
            rb = np.unique(np.concatenate([rb, (rb[:-1] + d // 2)[big]]))
        if lds > 160 * 1024 or op.max_inst * maxar > 32768:
            raise PlanDoesNotFit("owner-computes-rows plan does not fit (LDS or instance list)")
        three = (160 * 1024) // 3
        if kd_rows and not src.ocr_lds_limit and three < lds <= three + three // 4 and kd_rows >= 64 \
                and len(prep.setdefault("ocr_leaf_tries", {}).setdefault(gkey, [])) < 2:
            # The LDS footprint of a row block decides how many 512-lane groups a CU holds, and small element matrices want three
            # (3 x 53 KB).  The leaf size comes from an AVERAGE row length; a leaf a few percent too large leaves two groups per CU
            # (P1 on the 215^3 cube, profiles/r6y2_sweep_ocr_nnz_fp64*.txt: 288 rows = 52.9 KB run at 0.921 ms, 307 rows = 56.1 KB
            # at 0.98 ms): such a plan is rebuilt once or twice on leaves scaled to fit
            prep["ocr_leaf_tries"][gkey].append((kd_rows, lds))
            prep.setdefault("ocr_leaf_rows", {})[gkey] = max(32, int(kd_rows * three / lds * 0.98) // 32 * 32)
            return self._ocr_geometry(start0, end0)
        nds = [op.plans[mi].max_nd for mi in src.staged_maps]
        base = "ocrp" if row_order is not None else "ocr"
        fx_wanted = (int(configuration["ocr_fixed_point"]) > 0 and int(sp.dsets[0].cdim) * int(sp.dsets[1].cdim) == 1
                     and np.dtype(pa.data.dtype) == np.dtype("float64"))
        if (base == "ocrp" and configuration["ocr_flush_colmask"] and bool(pa.lgmaps) and pa.maps[0]._base() is pa.maps[1]._base()
                and not fx_wanted):
            base = "ocrpm"          # BC columns masked by the flush's place table, not by a select per contribution
        rec = None
        if configuration["ocr_records"] and len(src.staged_maps) <= 8:
            from .codegen import record_layout
            maxlen = max(sp._max_node_rowlen(), 1)
            rec = record_layout([staged[mi].arity for mi in src.staged_maps], nds, rmap.arity, cmap.arity, maxlen,
                                pa.maps[0]._base() is pa.maps[1]._base())
            if rec[3] * 4 >= sum(staged[mi].arity for mi in src.staged_maps) * 2 + rmap.arity * cmap.arity * op.kbytes:
                rec = None                                            # no smaller than the plain rows
        variant = mode_variant(base, op.kbytes, nds, rec)
        fxbufs = None
        if int(configuration["ocr_fixed_point"]) > 0 and int(sp.dsets[0].cdim) * int(sp.dsets[1].cdim) == 1 \
                and np.dtype(pa.data.dtype) == np.dtype("float64"):
            # checked fixed-point LDS accumulators (codegen mode suffix "_fx"): one 32-byte scale record per row block, written by the
            # block itself at the end of every launch (zero = no scale yet = an fp64 pass), and two counters
            # The 63-bit sums hold 2^12 contributions of |x S| < 2^50 per entry: an entry (r, c) collects at most one contribution per
            # entity around row node r, so the bound is the largest entity valence of a row node; maps that exceed it (degenerate or
            # star-shaped meshes) keep the fp64 atomics
            if self._max_row_valence(rmap, start, end) < 4096:
                variant += "_fx"
                fxbufs = (DeviceBuffer(max(op.nblocks, 1) * 32), DeviceBuffer(16))
                for b_ in fxbufs:
                    b_.zero()
        geo = {"ocr": op, "lds": lds, "k": k, "nnz": sp._nnz, "rec": rec, "fx": fxbufs,
               "cw": prep["cw"] if variant == src.mode else self.global_kernel.compile(variant), "row_order": row_order}
        prep["parts"][gkey] = geo
        if configuration["debug"]:
            import sys
            print(f"[fdhip] {self.global_kernel.name} OCR variant {variant}: record {rec}", file=sys.stderr)
            print(f"[fdhip] {self.global_kernel.name} OCR [{start},{end}): row blocks={op.nblocks} instances={op.ninst} "
                  f"(x{op.ninst / max(end - start, 1):.2f} entities) max_inst={op.max_inst} max_nnz={op.max_nnz} max_nown={op.max_nown} "
                  f"lds={lds} kbytes={op.kbytes}", file=sys.stderr)
        return geo

    def _ocrs_geometry(self, start, end, gkey):
        """Row-sliced owner-computes-rows plan (codegen.generate_sliced_wrapper): no instance is redundant, so the row blocks
        are sized for occupancy alone -- ranges of ~``ocrs_nnz_per_block`` accumulator entries of the caller's row order
        (producer hints present: that order is compact already) or of the backend-derived one."""
        from .op2types import RowOrder, SlicedOcrPlan
        from .codegen import lds_stride, mode_variant
        prep = self._prepared
        src = prep["cw"].src
        start0, end0 = start, end
        # subsets / extruded sets: every map is replaced by its derived map over the virtual (position x layer) space
        maps = {mi: self._plan_map(m._base(), staged=True) for mi, m in enumerate(prep["maps"])}
        v = self._virtual(staged=True)
        if v is not None:
            start, end = v.range(start, end)                    # positions in the virtual space
        (k, pa), = [(k, a) for k, a in enumerate(self.arguments) if isinstance(a, MatParloopArg)]
        rmap, cmap = (self._plan_map(m._base(), staged=True) for m in pa.maps)
        sp = pa.data.sparsity
        sp._build()
        nrows = rmap.toset.size
        rp = sp._node_rowptr_host()
        row_order = None
        prp = rp[:nrows + 1]
        hint = getattr(rmap._base(), "preferred_node_blocks", None)
        if hint is None or not configuration["use_preferred_blocks"]:
            # (virtual spaces too: an extruded numbering is column-major, so ranges of ITS rows are vertical pencils)
            order = self._locality_order(start, end, virtual=v is not None)
            if order is not None:
                row_order = RowOrder(rmap, order, end - start, nrows, rp, rowptr_dev=sp._node_rowptr.ptr)
                prp = row_order.prowptr_host
        B = int(sp.dsets[0].cdim) * int(sp.dsets[1].cdim)
        cap = max(int(configuration["ocrs_nnz_per_block"]) // B, int(np.diff(prp).max()) if nrows else 1)
        staged = {mi: maps[mi] for mi in src.staged_maps}
        limit = configuration["ocrs_lds_limit"] or configuration["lds_limit"]
        per_dof = bool(pa.lgmaps) and bool(self.global_kernel.arguments[k].unroll)
        # two rows per instance (one evaluation of the local kernel, one index record for both): scalar matrices with node lgmaps whose
        # element matrix has enough rows for the shared part to matter; the pairs are chosen once, from the first block cut
        want_pairs = (int(configuration["ocrs_pairs"]) > 0 and B == 1 and not per_dof and configuration["ocr_records"]
                      and 4 <= rmap.arity <= 32 and len(src.staged_maps) <= 8 and sp._max_node_rowlen() <= 254)
        groups = None
        for attempt in range(8):
            targets = np.arange(0, int(prp[nrows]) + cap, cap)
            rb = np.unique(np.concatenate([np.searchsorted(prp[:nrows + 1], targets, side="left"), [0, nrows]]))
            rb = rb[rb <= nrows]
            if want_pairs and groups is None:
                groups = SlicedOcrPlan.choose_groups(rmap, start, end, rb, row_order)
            try:
                op = SlicedOcrPlan(sp, rmap, cmap, staged, start, end, rb, row_order=row_order, groups=groups)
            except _lib.FDHipError as exc:
                if "too many" in str(exc):
                    raise PlanDoesNotFit(str(exc))          # 32-bit instance indices: the loop falls back (staged / direct)
                if "map entries" not in str(exc):
                    raise
                cap //= 2
                continue
            lds = ((op.max_nnz * B * 8) + 15) // 16 * 16
            for item in src.lds_items:
                if item[0] == "dat":
                    _, mi, c, isz, _ = item
                    lds += ((lds_stride(op.plans[mi].max_nd, ocr=True) * c * isz) + 15) // 16 * 16
            if lds <= limit and op.max_nnz < 0xffff:
                break
            cap //= 2
        else:
            raise PlanDoesNotFit("row-sliced owner-computes-rows plan does not fit")
        nds = [op.plans[mi].max_nd for mi in src.staged_maps]
        base, runs = ("ocrsp" if row_order is not None else "ocrs"), None
        if row_order is not None and B == 1 and configuration["ocrs_run_flush"]:
            runs = row_order.runs(op.row_blocks)
            if runs[3] <= 256 and lds + 256 * _lib.NNZ_BYTES <= limit:         # (the block's run displacements, fd_nnz_t each)
                base, lds = "ocrspr", lds + 256 * _lib.NNZ_BYTES
            else:
                runs = None
        rec = None
        if configuration["ocr_records"] and B == 1 and not per_dof and len(src.staged_maps) <= 8 and op.kbytes == 1:
            from .codegen import sliced_record_layout
            maxlen = max(sp._max_node_rowlen(), 1)
            nri = op.rows_per_inst
            rec = sliced_record_layout([staged[mi].arity for mi in src.staged_maps], nds, cmap.arity, maxlen, op.max_nnz, rows=nri)
            if rec[1] > 8 or rec[2] > 16 or rec[3] * 4 >= sum(staged[mi].arity for mi in src.staged_maps) * 2 + nri * (cmap.arity + 2):
                rec = None
        if groups is not None and rec is None:
            # (paired instances read everything from their record; fields that do not fit: back to one row per instance)
            configuration_pairs = configuration["ocrs_pairs"]
            configuration["ocrs_pairs"] = 0
            try:
                return self._ocrs_geometry(start0, end0, gkey)
            finally:
                configuration["ocrs_pairs"] = configuration_pairs
        variant = mode_variant(base, op.kbytes, nds, rec, groups=groups)
        geo = {"ocr": op, "lds": lds, "k": k, "nnz": sp._nnz, "runs": runs, "rec": rec, "groups": groups,
               "cw": prep["cw"] if variant == src.mode else self.global_kernel.compile(variant), "row_order": row_order}
        prep["parts"][gkey] = geo
        if configuration["debug"]:
            import sys
            print(f"[fdhip] {self.global_kernel.name} OCR row-sliced variant {variant}", file=sys.stderr)
            print(f"[fdhip] {self.global_kernel.name} OCR row-sliced [{start},{end}): row blocks={op.nblocks} instances={op.nreal} "
                  f"(+{op.ninst - op.nreal} padding, x{op.nreal / max((end - start) * rmap.arity, 1):.2f} of the map entries"
                  f"{'' if groups is None else ', row groups ' + str(groups)}) "
                  f"max_inst={op.max_inst} max_nnz={op.max_nnz} max_nown={op.max_nown} lds={lds} kbytes={op.kbytes}", file=sys.stderr)
        return geo

    def _compute_ocr(self, start=0, end=None):
        geo = self._ocr_geometry(start, end)
        cw = geo["cw"]
        src = cw.src
        op = geo["ocr"]
        if op.nblocks == 0 or op.ninst == 0:
            return
        getters = self._getters_of(geo, cw, lambda: self._ocr_arg_getters(geo, cw))
        if _CHECK_ARGLISTS:
            pending = [(pa.data, pa.data._zero_pending) for pa in self.arguments if isinstance(pa, MatParloopArg)]
        out = [g(self) for g in getters]
        if _CHECK_ARGLISTS:               # (tests: the branch-per-entry statement of the same list, from the same state)
            for mat, was in pending:
                mat._zero_pending = was
            vers = [(pa.data, pa.data.dat_version) for pa in self.arguments if hasattr(pa.data, "dat_version")]
            ref = self._ocr_args_chain(geo, cw)
            _restore_versions(vers)
            _same_arglists(src.layout, out, ref)
        cw.launch(0, op.ninst, out, block_threads=src.block_threads, ents_per_block=op.max_inst, nblocks=op.nblocks,
                  lds_bytes=geo["lds"])

    def _ocr_args_chain(self, geo, cw):
        """The argument list of an owner-computes-rows launch, one branch per layout entry: what ``_ocr_arg_getters`` pre-dispatches
        (kept as the statement of what each entry is; tests run both and compare: ``parloop._CHECK_ARGLISTS``)."""
        prep = self._prepared
        src = cw.src
        op = geo["ocr"]
        out = []
        for desc in src.layout:
            kind = desc[0]
            if kind == "layers":
                out.append(self.iterset._layers_dev())
            elif kind == "subset":
                out.append(self.iterset._indices_dev())
            elif kind == "arg":
                pa = self.arguments[desc[1]]
                if isinstance(pa, MatParloopArg):
                    mat = pa.data
                    mat.dat_version += 1
                    vals = mat._values_raw()
                    flag = 0
                    if mat._zero_pending:
                        # rows outside the blocks (ghost rows) are the only part the loop does not overwrite
                        tail = (geo["nnz"] - op.vals_end) * 8
                        if tail > 0:
                            _lib.call("fd_memset", vals.ptr + op.vals_end * 8, 0, tail, None)
                        mat._zero_pending = False
                        flag = 1
                    self._ocr_flag = flag
                    out.append(vals.ptr)
                else:
                    out.append(pa.data._dev_ptr(write=False))
            elif kind == "map":
                out.append(prep["maps"][desc[1]]._dev_values())
            elif kind == "bstart":
                out.append(op.inst_off)
            elif kind == "ocr_inst_ent":
                out.append(op.inst_ent)
            elif kind == "plan_blkoff":
                out.append(op.plans[desc[1]].blkoff)
            elif kind == "plan_list":
                out.append(op.plans[desc[1]].list)
            elif kind == "plan_lmap":
                out.append(op.plans[desc[1]].lmap)
            elif kind == "plan_maxnd":
                out.append(op.plans[desc[1]].max_nd)
            elif kind == "plan_copy":
                out.append(self._plan_copy(geo, desc[1], op.plans[desc[2]]))
            elif kind == "ocr_rblk":
                out.append(op.rblk)
            elif kind == "ocr_rowptr":
                out.append(self.arguments[desc[1]].data.sparsity._node_rowptr.ptr)
            elif kind == "ocr_kidx":
                out.append(op.kidx.ptr)
            elif kind == "ocrs_chunk_role":
                out.append(op.chunk_role)
            elif kind in ("ocrs_slot", "ocrs_kk", "ocrs_rowlen", "ocrs_rmask", "ocrs_cmask"):
                lg = self.arguments[desc[1]].lgmaps
                per_dof = bool(lg) and bool(self.global_kernel.arguments[desc[1]].unroll)
                tabs = op.tables(lg[0] if lg else None, lg[1] if lg else None, self._lgmap, per_dof=per_dof)
                out.append(tabs[{"ocrs_slot": 0, "ocrs_kk": 1, "ocrs_rowlen": 2, "ocrs_rmask": 3, "ocrs_cmask": 4}[kind]].ptr)
            elif kind == "ocr_maxnnz":
                out.append(op.max_nnz)
            elif kind == "ocr_maxnown":
                out.append(op.max_nown)
            elif kind == "ocr_flags":
                out.append(self._ocr_flag)
            elif kind == "ocr_prowptr":
                out.append(geo["row_order"].prowptr.ptr)
            elif kind == "ocr_nstart":
                out.append(geo["row_order"].nstart.ptr)
            elif kind == "ocr_gstart":
                out.append(geo["row_order"].gstart.ptr)
            elif kind == "ocr_srow":
                out.append(self._ocr_node_words(geo, desc[1], desc[2], diag=len(desc) > 3))
            elif kind == "ocr_rec" and src.mode.startswith("ocrs"):
                lbits, kbits, sbits, words = geo["rec"]
                lg = self.arguments[desc[1]].lgmaps
                out.append(op.records(lg[0] if lg else None, lg[1] if lg else None, self._lgmap, src.staged_maps, lbits, kbits, sbits, words).ptr)
            elif kind == "ocr_rec":
                lbits, kbits, diag, words = geo["rec"]
                rm_, cm_ = self.arguments[desc[1]].maps
                out.append(op.records(src.staged_maps, lbits, kbits, diag, words, rm_.arity, cm_.arity).ptr)
            elif kind in ("ocr_grun", "ocr_brun", "ocr_rdelta"):
                out.append(geo["runs"][{"ocr_grun": 0, "ocr_brun": 1, "ocr_rdelta": 2}[kind]].ptr)
            elif kind == "ocr_gpos":
                if src.mode.startswith("ocrpm"):
                    pa_ = self.arguments[desc[1]]
                    out.append(geo["row_order"].gpos_masked(pa_.data.sparsity, pa_.lgmaps[1], self._lgmap).ptr)
                else:
                    out.append(geo["row_order"].gpos().ptr)
            elif kind == "ocr_npos":
                out.append(geo["row_order"].npos)
            elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
                pa = self.arguments[desc[1]]
                out.append(self._lgmap(pa.lgmaps[0 if kind == "mat_row_lgmap" else 1]))
            elif kind in ("virt_col", "virt_layer"):
                out.append(self._virtual(staged=True).tables_dev()[0 if kind == "virt_col" else 1].ptr)
            elif kind in ("fx_scale", "fx_stat"):
                out.append(geo["fx"][0 if kind == "fx_scale" else 1].ptr)
            elif kind == "phase_times":
                if geo.get("phase_times") is None:
                    geo["phase_times"] = DeviceBuffer(max(op.nblocks, 1) * 40)
                out.append(geo["phase_times"].ptr)
            else:
                raise AssertionError(kind)
        return out

    def _ocr_arg_getters(self, geo, cw):
        """One callable ``f(parloop)`` per layout entry of the launch, built once per (geometry, wrapper): the per-call marshalling of
        SURVEY.md 7 hard part (f) was a chain of string comparisons per entry and call.  Entries whose value depends on the state of
        a carrier (a Mat's pending zero, a Dat's device pointer, a plan copy, the lgmap tables) stay calls; everything is looked up
        through ``geo`` at call time, so tables replaced inside a geometry are seen.  The callables take the Parloop as their argument and
        are kept on it (``_getters_of``), not in ``geo``: closures over ``self`` stored in ``geo`` would tie Parloop, geometry and
        plans into a reference cycle, and the device memory of a dropped problem would wait for the cycle collector."""
        prep = self._prepared
        src = cw.src
        args = self.arguments
        g = []
        for desc in src.layout:
            kind = desc[0]
            if kind == "layers":
                g.append(lambda pl: pl.iterset._layers_dev())
            elif kind == "subset":
                g.append(lambda pl: pl.iterset._indices_dev())
            elif kind == "arg":
                pa = args[desc[1]]
                if isinstance(pa, MatParloopArg):
                    def mat_arg(pl, pa=pa):
                        mat = pa.data                        # (read at every call: the reference's assemblers swap the output tensor
                        mat.dat_version += 1                 #  of a cached Parloop -- parloop.arguments[0].data = ..., assemble.py:1073-1077)
                        vals = mat._values_raw()
                        flag = 0
                        if mat._zero_pending:
                            # rows outside the blocks (ghost rows) are the only part the loop does not overwrite
                            op = geo["ocr"]
                            tail = (geo["nnz"] - op.vals_end) * 8
                            if tail > 0:
                                _lib.call("fd_memset", vals.ptr + op.vals_end * 8, 0, tail, None)
                            mat._zero_pending = False
                            flag = 1
                        pl._ocr_flag = flag
                        return vals.ptr
                    g.append(mat_arg)
                else:
                    g.append(lambda pl, pa=pa: pa.data._dev_ptr(write=False))
            elif kind == "map":
                g.append(lambda pl, k=desc[1]: prep["maps"][k]._dev_values())
            elif kind == "bstart":
                g.append(lambda pl: geo["ocr"].inst_off)
            elif kind == "ocr_inst_ent":
                g.append(lambda pl: geo["ocr"].inst_ent)
            elif kind == "plan_blkoff":
                g.append(lambda pl, k=desc[1]: geo["ocr"].plans[k].blkoff)
            elif kind == "plan_list":
                g.append(lambda pl, k=desc[1]: geo["ocr"].plans[k].list)
            elif kind == "plan_lmap":
                g.append(lambda pl, k=desc[1]: geo["ocr"].plans[k].lmap)
            elif kind == "plan_maxnd":
                g.append(lambda pl, k=desc[1]: geo["ocr"].plans[k].max_nd)
            elif kind == "plan_copy":
                g.append(lambda pl, k=desc[1], m=desc[2]: pl._plan_copy(geo, k, geo["ocr"].plans[m]))
            elif kind == "ocr_rblk":
                g.append(lambda pl: geo["ocr"].rblk)
            elif kind == "ocr_rowptr":
                g.append(lambda pl, k=desc[1]: args[k].data.sparsity._node_rowptr.ptr)
            elif kind == "ocr_kidx":
                g.append(lambda pl: geo["ocr"].kidx.ptr)
            elif kind == "ocrs_chunk_role":
                g.append(lambda pl: geo["ocr"].chunk_role)
            elif kind in ("ocrs_slot", "ocrs_kk", "ocrs_rowlen", "ocrs_rmask", "ocrs_cmask"):
                which = {"ocrs_slot": 0, "ocrs_kk": 1, "ocrs_rowlen": 2, "ocrs_rmask": 3, "ocrs_cmask": 4}[kind]

                def ocrs_table(pl, k=desc[1], which=which):
                    lg = args[k].lgmaps
                    per_dof = bool(lg) and bool(pl.global_kernel.arguments[k].unroll)
                    return geo["ocr"].tables(lg[0] if lg else None, lg[1] if lg else None, pl._lgmap, per_dof=per_dof)[which].ptr
                g.append(ocrs_table)
            elif kind == "ocr_maxnnz":
                g.append(lambda pl: geo["ocr"].max_nnz)
            elif kind == "ocr_maxnown":
                g.append(lambda pl: geo["ocr"].max_nown)
            elif kind == "ocr_flags":
                g.append(lambda pl: pl._ocr_flag)
            elif kind == "ocr_prowptr":
                g.append(lambda pl: geo["row_order"].prowptr.ptr)
            elif kind == "ocr_nstart":
                g.append(lambda pl: geo["row_order"].nstart.ptr)
            elif kind == "ocr_gstart":
                g.append(lambda pl: geo["row_order"].gstart.ptr)
            elif kind == "ocr_srow":
                g.append(lambda pl, k=desc[1], m=desc[2], dg=len(desc) > 3: pl._ocr_node_words(geo, k, m, diag=dg))
            elif kind == "ocr_rec" and src.mode.startswith("ocrs"):
                def rec_sliced(pl, k=desc[1]):
                    lbits, kbits, sbits, words = geo["rec"]
                    lg = args[k].lgmaps
                    return geo["ocr"].records(lg[0] if lg else None, lg[1] if lg else None, pl._lgmap, src.staged_maps, lbits, kbits, sbits, words).ptr
                g.append(rec_sliced)
            elif kind == "ocr_rec":
                def rec_whole(pl, k=desc[1]):
                    lbits, kbits, diag, words = geo["rec"]
                    rm_, cm_ = args[k].maps
                    return geo["ocr"].records(src.staged_maps, lbits, kbits, diag, words, rm_.arity, cm_.arity).ptr
                g.append(rec_whole)
            elif kind in ("ocr_grun", "ocr_brun", "ocr_rdelta"):
                g.append(lambda pl, w={"ocr_grun": 0, "ocr_brun": 1, "ocr_rdelta": 2}[kind]: geo["runs"][w].ptr)
            elif kind == "ocr_gpos":
                if src.mode.startswith("ocrpm"):
                    g.append(lambda pl, pa_=args[desc[1]]: geo["row_order"].gpos_masked(pa_.data.sparsity, pa_.lgmaps[1], pl._lgmap).ptr)
                else:
                    g.append(lambda pl: geo["row_order"].gpos().ptr)
            elif kind == "ocr_npos":
                g.append(lambda pl: geo["row_order"].npos)
            elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
                g.append(lambda pl, pa=args[desc[1]], w=0 if kind == "mat_row_lgmap" else 1: pl._lgmap(pa.lgmaps[w]))
            elif kind in ("virt_col", "virt_layer"):
                g.append(lambda pl, w=0 if kind == "virt_col" else 1: pl._virtual(staged=True).tables_dev()[w].ptr)
            elif kind in ("fx_scale", "fx_stat"):
                g.append(lambda pl, w=0 if kind == "fx_scale" else 1: geo["fx"][w].ptr)
            elif kind == "phase_times":
                def phase_times(pl):
                    if geo.get("phase_times") is None:
                        geo["phase_times"] = DeviceBuffer(max(geo["ocr"].nblocks, 1) * 40)
                    return geo["phase_times"].ptr
                g.append(phase_times)
            else:
                raise AssertionError(kind)
        return g

    @staticmethod
    def _max_row_valence(rmap, start, end):
        """Largest number of entities of [start, end) around one node of ``rmap`` (host pass, opt-in fixed-point mode only)."""
        vals = np.asarray(rmap.values_with_halo if hasattr(rmap, "values_with_halo") else rmap.values)[start:end].ravel()
        vals = vals[vals >= 0]
        return int(np.bincount(vals).max()) if len(vals) else 0

    def fixed_point_state(self):
        """Diagnostics of the checked fixed-point accumulation of this loop's owner-computes-rows parts (blocks the stream): per
        part the number of row blocks, how many of them hold a scale, the range of their limit exponents L (a contribution of
        such a block must stay below 2^L), and the running counts of blocks that fell back from a scale to fp64 / ran without one."""
        out = []
        for key, geo in (self._prepared or {}).get("parts", {}).items():
            if key[0] != "ocr" or not isinstance(geo, dict) or geo.get("fx") is None:
                continue
            nb = geo["ocr"].nblocks
            rec = geo["fx"][0].download(np.uint8, (max(nb, 1) * 32,)).view(np.dtype([("S", "<f8"), ("invS", "<f8"), ("lim", "<u4"), ("low", "<u4"),
                                                                                      ("L", "<i4"), ("pad", "<u4")]))[:nb]
            st = geo["fx"][1].download(np.uint32, (4,))
            have = rec["S"] > 0
            out.append({"part": key[1:], "blocks": int(nb), "scaled_blocks": int(have.sum()),
                        "limit_exponents": (int(rec["L"][have].min()), int(rec["L"][have].max())) if have.any() else None,
                        "fallback_blocks": int(st[0]), "unscaled_blocks": int(st[1])})
        return out

    def _plan_copy(self, geo, k, plan):
        """Device pointer of READ Dat ``k`` in PLAN order (row of list entry i at i*cdim), or 0: a field that did not change since
        the previous call of this loop (coordinates, a forcing term; ``dat_version`` decides) is copied once -- one coalesced
        gather over the plan's node list -- and streamed by the staging phase from then on; a field that changes between
        calls (the state of a Newton iteration) keeps the in-kernel gather, and a copy is dropped the moment its Dat changes."""
        if not configuration["plan_copies"]:
            return 0
        from . import op2types
        if op2types._capture_log is not None:
            return 0          # (a recorded step gathers from the Dat itself: a copy baked into the graph would never see the Dat change)
        d = self.arguments[k].data
        d = getattr(d, "_parent", d)
        ver = getattr(d, "dat_version", None)
        if ver is None or not hasattr(d, "cdim"):
            return 0
        cache = geo.setdefault("plan_copies", {})
        ent = cache.get(k)
        if ent is not None and ent["of"] is not d:
            ent = None                                        # (another Dat in this slot: nothing kept of the old one)
        if ent is not None and ent["version"] == ver and ent["buf"] is not None:
            return ent["buf"].ptr
        if ent is None or ent["version"] != ver:
            cache[k] = {"version": ver, "buf": None, "of": d}          # first sight of this state of the Dat: gather in the kernel
            return 0
        words = d.cdim * np.dtype(d.dtype).itemsize // 4      # unchanged since the last call: make the copy (rows as 32-bit words)
        if words * 4 != d.cdim * np.dtype(d.dtype).itemsize or plan.list_len == 0:
            return 0
        buf = DeviceBuffer(plan.list_len * words * 4)
        _lib.call("fd_gather_rows", d._dev_ptr(False), words, plan.list, plan.list_len, buf.ptr, None)
        ent["buf"] = buf
        return buf.ptr

    def _ocr_node_words(self, geo, k, rm, diag=False):
        """Plan-ordered row words of Mat argument ``k`` (fd_ocr_node_words) for its current pair of lgmaps: built on first use
        per pair and cached by identity -- the reference swaps lgmaps per call (parloop.py:279-314), a set of boundary
        conditions is assembled many times."""
        pa = self.arguments[k]
        lg = pa.lgmaps or (None, None)
        ident = lambda o: ("dev", o._fd_dev_ptr, getattr(o, "_fd_token", None)) if hasattr(o, "_fd_dev_ptr") else id(o)   # noqa: E731
        key = (k, ident(lg[0]), ident(lg[1]), bool(diag))
        cache = geo.setdefault("srow", {})
        hit = cache.get(key)
        if hit is None:
            op, ro = geo["ocr"], geo["row_order"]
            plan = op.plans[rm]
            sp = pa.data.sparsity
            buf = DeviceBuffer(max(plan.list_len, 1) * 4)
            base = ro.nstart.ptr if ro is not None else sp._node_rowptr.ptr
            starts = ro.prowptr.ptr if ro is not None else sp._node_rowptr.ptr
            _lib.call("fd_ocr_node_words", plan.blkoff, plan.list, op.nblocks, op.rblk, base, starts, 1 if ro is not None else 0,
                      ro.npos if ro is not None else 0, self._lgmap(lg[0]) if lg[0] is not None else None,
                      self._lgmap(lg[1]) if lg[1] is not None else None, buf.ptr, None)
            if diag:
                # the place of a row's diagonal entry rides in the node word (records without the (i, i) offsets)
                _lib.call("fd_ocr_node_diag", plan.list, plan.list_len, int(sp.dsets[0].set.total_size), sp._node_rowptr.ptr,
                          sp._node_colidx.ptr, buf.ptr, None)
            while len(cache) >= 4:
                cache.pop(next(iter(cache)))
            hit = (lg, buf)
            cache[key] = hit
        return hit[1].ptr

    def _nlayers_iterated(self):
        from .op2types import ON_BOTTOM, ON_TOP, ON_INTERIOR_FACETS
        L = self.iterset.layers
        reg = self.global_kernel._iteration_region
        if reg in (ON_BOTTOM, ON_TOP):
            return 1
        if reg == ON_INTERIOR_FACETS:
            return max(L - 1 if self.global_kernel._extruded_periodic else L - 2, 0)
        return L - 1

    # -- halo protocol (parloop.py:320-409)
    def _indirect_dats(self):
        """(Dat, access) of the indirectly accessed Dats, each data carrier once (the reference's ``seen`` sets,
        pyop2/parloop.py:343-352, 393-403): one exchange per distinct Dat and direction.  The ARGUMENTS of a Parloop are fixed and
        looked up once; their ``data`` is read at every call (the reference's assemblers swap the output tensor of a cached Parloop,
        firedrake/assemble.py:1073-1077)."""
        pas = self.__dict__.get("_indirect_args")
        if pas is None:
            pas = self._indirect_args = [(pa, acc) for pa, acc in zip(self.arguments, self.accesses)
                                         if isinstance(pa, DatParloopArg) and pa.map_ is not None]
        lst, seen = [], set()
        for pa, acc in pas:
            d = getattr(pa.data, "_parent", pa.data)          # a DatView exchanges its parent's storage
            if id(d) in seen:
                continue
            seen.add(id(d))
            lst.append((pa.data, acc))
        return lst

    def _halos_present(self):
        """Does any Dat argument live on a Set with a halo right now?  (One rank: no -- the four exchange phases and the
        ``halo_valid`` bookkeeping are then skipped as a whole; every one of them returns at once for such a Dat anyway.)"""
        for pa in self.arguments:
            if isinstance(pa, DatParloopArg) and pa.data.dataset.set.halo is not None:
                return True
        return False

    _halos_now = None         # set by compute() for the duration of one call; None = look at every Dat (calls from outside)

    def global_to_local_begin(self):
        if self._halos_now is False:
            return
        for d, acc in self._indirect_dats():
            if acc != WRITE:
                d.global_to_local_begin(acc)

    def global_to_local_end(self):
        if self._halos_now is False:
            return
        for d, acc in self._indirect_dats():
            if acc != WRITE:
                d.global_to_local_end(acc)

    def local_to_global_begin(self):
        if self._halos_now is False:
            return
        for d, acc in self._indirect_dats():
            if acc in (INC, MIN, MAX):
                d.local_to_global_begin(acc)

    def local_to_global_end(self):
        if self._halos_now is False:
            return
        for d, acc in self._indirect_dats():
            if acc in (INC, MIN, MAX):
                d.local_to_global_end(acc)
        for pa, acc in zip(self.arguments, self.accesses):
            if isinstance(pa, DatParloopArg) and acc != READ:
                if pa.data.dataset.set.halo is not None:
                    pa.data.halo_valid = False

    def _global_reductions(self):
        lst = self.__dict__.get("_reduction_list")
        if lst is None:
            lst = self._reduction_list = [(k, pa, acc) for k, (pa, acc) in enumerate(zip(self.arguments, self.accesses))
                                          if isinstance(pa, GlobalParloopArg) and acc in (INC, MIN, MAX)]
        return lst

    def _zero_global_temporaries(self):
        """pyop2/parloop.py:274-277, 516-532: with more than one rank an INC Global accumulates this loop's
        contributions in a zeroed temporary; the all-reduced temporary is then added to the Global."""
        self._glob_saved = {}
        if not self._global_reductions():
            return
        from .halo import world_size
        if world_size() == 1:
            return
        for k, pa, acc in self._global_reductions():
            if acc == INC:
                self._glob_saved[k] = np.array(pa.data.data_ro, copy=True)
                pa.data.data[...] = 0

    def reduction_begin(self):        # parloop.py:411-442 (MPI_Iallreduce of Globals)
        pass

    def reduction_end(self):
        red = self._global_reductions()
        if not red:
            return
        from .halo import allreduce_global
        for k, pa, acc in red:
            allreduce_global(pa.data, acc, self.iterset.comm)
            if k in self._glob_saved:
                pa.data.data[...] += self._glob_saved[k]

    def finalize_assembly(self):
        pass


def parloop(knl, *args, **kwargs):
    """pyop2/parloop.py:746-763."""
    if isinstance(knl, GlobalKernel):
        Parloop(knl, *args, **kwargs)()
    else:
        LegacyParloop(knl, *args, **kwargs)()


class LegacyParloop(Parloop):
    """pyop2/parloop.py:709-743: build the GlobalKernel from dat(access, map)-style args."""

    def __init__(self, local_knl, iterset, *args, **kwargs):
        if not isinstance(local_knl, CStringLocalKernel):
            from .exceptions import KernelTypeError
            raise KernelTypeError("par_loop needs a Kernel")             # parloop.py:752-756
        if not isinstance(iterset, Set):
            raise SetTypeError("Iteration set is of the wrong type")
        for a in args:
            if not isinstance(a, (DatLegacyArg, GlobalLegacyArg, MatLegacyArg, MixedDatLegacyArg, MixedMatLegacyArg)):
                raise ValueError("par_loop arguments must be created by calling a Dat/Global/Mat with an access mode")
        if local_knl.accesses is None:
            local_knl = local_knl.with_signature([a.access for a in args], [a.dtype for a in args])
        extruded = iterset._extruded
        subset = isinstance(iterset, Subset)
        gk = _global_kernel_cached(local_knl, args, extruded=extruded,
                                   constant_layers=bool(extruded and iterset.constant_layers), subset=subset,
                                   extruded_periodic=bool(extruded and iterset._extruded_periodic),
                                   iteration_region=kwargs.get("iteration_region"),
                                   pass_layer_arg=kwargs.get("pass_layer_arg", False))
        super().__init__(gk, iterset, [a.parloop_arg for a in args])


_gk_cache = {}


def _global_kernel_cached(lk, args, **flags):
    gargs = [a.global_kernel_arg for a in args]
    gk = GlobalKernel(lk, gargs, **flags)
    hit = _gk_cache.get(gk.cache_key)
    if hit is not None:
        return hit
    _gk_cache[gk.cache_key] = gk
    return gk


def par_loop(kernel, iterset, *args, **kwargs):
    """pyop2/parloop.py:746-763 ``par_loop``: build and run immediately."""
    LegacyParloop(kernel, iterset, *args, **kwargs)()


ParLoop = LegacyParloop
