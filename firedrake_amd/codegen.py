"""Wrapper code generation: GlobalKernel -> HIP kernel source for gfx950.

This is the backend's counterpart of pyop2/codegen/builder.py (WrapperBuilder,
DatPack/GlobalPack/MatPack) + pyop2/codegen/rep2loopy.py:409-593.  The reference turns
the pack / call / unpack description into one sequential C loop per MPI rank; here the same
description becomes a HIP kernel built from the hand-written device pieces in
csrc/fd_wrapper.h.  Two shapes are emitted:

``staged``  (the fast path; SURVEY.md 8a rows a2+a3)
    one workgroup per block of iteration-set entities (fd_plan_*).  The distinct Dat rows
    the block touches are gathered once -- coalesced over the block's sorted node list --
    into LDS; one lane per entity reads its element pack from LDS through a uint16 local
    map, calls the local kernel in registers, and reduces INC contributions in LDS
    (ds_add_f64); the block then issues ONE global atomic per distinct node.
    Eligible when every indirect Dat is READ or INC, not extruded, no subset.

``direct``  (always available)
    one lane per entity (x layer), gather/scatter straight from global memory with
    hardware atomics.  Covers RW/WRITE/MIN/MAX through maps, subsets, extruded columns
    (builder.py:94-124, 790-831), permuted maps (builder.py:144-176), integer Dats.

Matrix arguments (MatPack, builder.py:520-625) are scattered into the device CSR with
fp64 atomics, locating each entry either through the precomputed element->nonzero table
(fd_csr_elem_offsets) or by row search; BC-masked lgmaps drop rows/cols exactly as PETSc
drops negative indices (parloop.py:279-302).

The kernel's parameter list starts with the reference's own positional list
(builder.py:962-981): start, end, [layers], [subset_indices], one pointer per
Dat/Global/Mat, one pointer per distinct Map; backend-private tables follow.
"""
from __future__ import annotations

import re
from dataclasses import dataclass, field
from typing import List

import numpy as np

from .configuration import configuration
from .kernel import (DatKernelArg, GlobalKernel, GlobalKernelArg, MapKernelArg, MatKernelArg, PassthroughKernelArg,
                     PermutedMapKernelArg)
from .op2types import (ALL, INC, MAX, MIN, ON_BOTTOM, ON_INTERIOR_FACETS, ON_TOP, READ, RW, WRITE)

CTYPE = {np.dtype("float64"): "double", np.dtype("float32"): "float", np.dtype("int32"): "int",
         np.dtype("uint32"): "unsigned int", np.dtype("int64"): "int64_t", np.dtype("uint64"): "uint64_t"}

STAGEABLE_INC = {np.dtype("float64"), np.dtype("float32"), np.dtype("int32"), np.dtype("uint32")}


def stages_in_lds(access, dtype) -> bool:
    """An indirect Dat argument the staged wrapper keeps in LDS: READ rows are gathered once per block, INC rows reduced there.
    WRITE / RW / MIN / MAX (and INC of a type without an LDS atomic) go straight to global memory from the lane, as in the
    direct wrapper -- beside the staged arguments of the same loop."""
    return access == READ or (access == INC and np.dtype(dtype) in STAGEABLE_INC)


@dataclass
class WrapperSource:
    source: str
    symbol: str
    mode: str
    layout: List[tuple]                                   # kernel parameters after (start, end)
    nmaps: int
    staged_maps: List[int] = field(default_factory=list)  # distinct-map indices that need a plan
    lds_items: List[tuple] = field(default_factory=list)  # (map index, elements per node, itemsize)
    layer_parallel: bool = True
    block_threads: int = 256
    kbytes: int = 1
    mat_staged: dict = field(default_factory=dict)
    lane_threads: int = 0                                  # >0: plans must be in lane order for this many lanes
    ocr_lds_limit: int = 0                                 # LDS budget of an OCR row block (0 = configuration["lds_limit"])
    extra_flags: tuple = ()                                # per-kernel hipcc flags chosen at JIT time (kernel.GlobalKernel.compile)
    tp: dict = field(default_factory=dict)                 # tensor-product wrappers: tensor_geometry() of the element


def _distinct_maps(gk: GlobalKernel):
    """Distinct base maps in first-use order (global_kernel.py:309-314, parloop.py:210-212)."""
    order, index = [], {}
    for a in gk.arguments:
        for m in getattr(a, "maps", ()):
            base = m.base_map if isinstance(m, PermutedMapKernelArg) else m
            if id(base) not in index:
                index[id(base)] = len(order)
                order.append(base)
    return order, index


def tensor_eligible(gk: GlobalKernel):
    """'matrix' / 'action' when the loop can take the tensor-product wrappers of csrc/fd_tensor.h: a
    TensorProductLocalKernel of degree k = 1..8 with up to 11 Gauss points per axis (tensor_geometry) over an extruded set
    with constant layers, the whole column (iteration region ALL), no subset, and the argument shapes
        matrix:  Mat INC (dims (D, D), both maps the (k+1)^3-node Q_k map, offset k)  +  coordinates READ (dim 3, 8-node Q1 map)
        action:  Dat INC (dim D, Q_k map)  +  coordinates READ  +  Dat READ (dim D, the same Q_k map)
    with D = the descriptor's ``vdim`` (1: scalar space)."""
    tp = getattr(gk.local_kernel, "tp", None)
    if not tp or not configuration["tensor_wrappers"] or tensor_geometry(tp["degree"], tp["nq"]) is None:
        return None
    nd, k = (tp["degree"] + 1) ** 3, tp["degree"]
    if not gk._extruded or not gk._constant_layers or gk._subset or gk._extruded_periodic or gk._iteration_region != ALL or gk._pass_layer_arg:
        return None
    args, las = gk.arguments, gk.local_kernel.arguments
    f64 = np.dtype("float64")

    def plain(m, arity, off):
        return isinstance(m, MapKernelArg) and m.arity == arity and m.offset is not None and tuple(m.offset) == (off,) * arity

    def coords_ok(a, la):
        return isinstance(a, DatKernelArg) and a.index is None and la.access == READ and la.dtype == f64 and tuple(a.dim) == (3,) \
            and plain(a.map_, 8, 1)

    nc, D = int(tp.get("ncoef", 0)), int(tp.get("vdim", 1))

    def coefs_ok(first, qmap):
        """the descriptor's coefficient arguments: nc scalar fp64 READ Dats after the standard arguments, each on the Q_k map or on
        the Q1 map of the coordinates (tensor_coefficient_spaces)"""
        rest = list(zip(args[first:], las[first:]))
        return len(rest) == nc and all(isinstance(a, DatKernelArg) and a.index is None and int(np.prod(a.dim)) == 1
                                       and (a.map_ is qmap or a.map_ is args[1].map_)
                                       and la.access == READ and la.dtype == f64 for a, la in rest)

    if tp["kind"] == "matrix" and len(args) == 2 + nc:
        a, la = args[0], las[0]
        if isinstance(a, MatKernelArg) and la.access == INC and not a.unroll and a.maps[0] is a.maps[1] and plain(a.maps[0], nd, k) \
                and int(np.prod(a.dims[0])) == D and int(np.prod(a.dims[1])) == D and coords_ok(args[1], las[1]) and coefs_ok(2, a.maps[0]):
            return "matrix"
    if tp["kind"] == "action" and len(args) == 3 + nc:
        y, u = args[0], args[2]
        if all(isinstance(d, DatKernelArg) and d.index is None and int(np.prod(d.dim)) == D for d in (y, u)) \
                and las[0].access == INC and las[2].access == READ and las[0].dtype == f64 and las[2].dtype == f64 \
                and y.map_ is u.map_ and plain(y.map_, nd, k) and coords_ok(args[1], las[1]) and coefs_ok(3, y.map_):
            return "action"
    return None


def tensor_coefficient_spaces(gk: GlobalKernel, kind: str):
    """'k' / '1' per coefficient argument of a tensor-product loop (in TSFC's order): on the Q_k map of the unknown or on the Q1 map
    of the coordinates."""
    first = 2 if kind == "matrix" else 3
    return ["1" if a.map_ is gk.arguments[1].map_ else "k" for a in gk.arguments[first:]]


def tensor_geometry(degree, nq):
    """Launch shape of the tensor-product templates for Q_degree with nq Gauss points per axis, as csrc/fd_tensor.h derives it
    (tp_tiles, tp_waves, tp_action_cells), or None outside the instantiated range: ``matrix_threads`` lanes and ``matrix_groups``
    workgroups per cell (one wavefront per 16-row panel of the padded element matrix), ``action_cells`` cells per 128-lane
    workgroup of the action."""
    k1, q1 = int(degree) + 1, int(nq)
    # degree <= 8 and max(k + 1, nq) <= 11: one cell's line set (max^2 lanes) fits the action's 128-lane workgroup.  The matrix template
    # keeps the per-point weights of a whole cell in LDS up to 48 KB and computes them plane by plane beyond (fd_tensor.h tp_weight_slabs)
    if not (1 <= degree <= 8 and 1 <= q1 <= 11):
        return None
    nt = (k1 ** 3 + 15) // 16
    # Q5+ (more than 8 tiles per side): a 16-row panel is cut into column chunks of <= 8 tiles, one wavefront per (panel, chunk)
    ct = int(configuration["tp_chunk_tiles"])
    ncs = 1 if nt <= int(configuration["tp_max_panel_tiles"]) else (nt + ct - 1) // ct
    items = nt * ncs
    wpb = 4 if items % 4 == 0 else (2 if items % 2 == 0 else 1)
    m = max(k1, q1)
    return {"k1": k1, "q1": q1, "nd": k1 ** 3, "tiles": nt, "col_splits": ncs, "col_tiles": -(-nt // ncs), "matrix_threads": 64 * wpb,
            "matrix_groups": items // wpb, "action_cells": 128 // (m * m)}


def tensor_matrix_groups(geom, vdim):
    """workgroups per cell of the matrix template: one per (panel group, component pair) of a (Q_k)^vdim space, or -- small elements,
    fd_tensor.h tp_fused -- all vdim^2 pairs of a panel group in one"""
    fused = (vdim > 1 and 4 * geom["tiles"] * vdim * vdim <= 96 and geom["col_splits"] == 1
             and geom["q1"] ** 3 * 16 * vdim * vdim * 8 <= int(configuration["tp_weight_lds"]))
    return geom["matrix_groups"] * (1 if fused else vdim * vdim)


def generate_tensor_wrapper(gk: GlobalKernel) -> WrapperSource:
    """The wrapper of a tensor-product loop: the reference's positional list for an extruded loop (start, end, layers, one
    pointer per argument, one per distinct Map, builder.py:962-981), then the backend-private tables, around the device
    templates of csrc/fd_tensor.h with the kernel's weight callback inlined."""
    kind = tensor_eligible(gk)
    lk = gk.local_kernel
    geom = tensor_geometry(lk.tp["degree"], lk.tp["nq"])
    sym = f"wrap_{lk.name}"
    wname = f"{lk.name}_weights"
    layout = [("layers",)]
    tune = [f"#define {macro} {int(configuration[key])}" for macro, key, default in
            (("FD_TP_MAX_PANEL_TILES", "tp_max_panel_tiles", 8), ("FD_TP_CHUNK_TILES", "tp_chunk_tiles", 8), ("FD_TP_WEIGHT_LDS", "tp_weight_lds", 48 * 1024))
            if int(configuration[key]) != default]
    head = tune + ['#include "fd_tensor.h"', "#include <math.h>", "namespace fdk {", "#pragma clang force_cuda_host_device begin",
            lk.tp["weights_code"], "#pragma clang force_cuda_host_device end", "}  // namespace fdk", ""]
    # coefficient arguments (READ Dats on the Q_k map or on the Q1 map of the coordinates): evaluated at the Gauss points by the
    # templates, which hand the callback C = [Q_k coefficients ..., Q1 coefficients ...]; the lambda restores TSFC's order
    nc = int(lk.tp.get("ncoef", 0))
    first_c = 2 if kind == "matrix" else 3
    spaces = tensor_coefficient_spaces(gk, kind) if nc else []
    ck = [m for m in range(nc) if spaces[m] == "k"]
    c1 = [m for m in range(nc) if spaces[m] == "1"]
    nck, nc1 = len(ck), len(c1)
    tpos = {m: p_ for p_, m in enumerate(ck + c1)}            # position of TSFC coefficient m in the templates' C
    grad, D = bool(lk.tp.get("coef_gradients")), int(lk.tp.get("vdim", 1))
    if nc:
        perm = ", ".join(f"C[{tpos[m]}]" for m in range(nc))
        wcall = f"const double Cu[{nc}] = {{{perm}}}; "
        if grad:
            gperm = ", ".join(f"DC[{3 * tpos[m] + a}]" for m in range(nc) for a in range(3))
            wcall += f"const double DCu[{3 * nc}] = {{{gperm}}}; fdk::{wname}(J, X, wq, Cu, DCu, W);"
        else:
            wcall += f"fdk::{wname}(J, X, wq, Cu, W); (void)DC;"
    else:
        wcall = f"fdk::{wname}(J, X, wq, W); (void)C; (void)DC;"
    call_w = (f"[](const double J[3][3], const double X[3], double wq, const double *C, const double *DC, double W[{16 * D * D}]) "
              f"{{ {wcall} }}")
    targs = f"{geom['k1']}, {geom['q1']}, {nck}, {nc1}, {'true' if grad else 'false'}, {D}"
    cparams = [f"const double *__restrict__ arg{first_c + m}" for m in range(nc)]
    clayout = [("arg", first_c + m) for m in range(nc)]
    cfdecl = ("  const double *const cf[%d] = {%s};\n  const double *const c1[%d] = {%s};"
              % (max(nck, 1), ", ".join(f"arg{first_c + m}" for m in ck) or "nullptr",
                 max(nc1, 1), ", ".join(f"arg{first_c + m}" for m in c1) or "nullptr"))
    if kind == "matrix":
        lg = bool(gk.arguments[0].lgmaps)
        layout += [("arg", 0), ("arg", 1)] + clayout + [("map", 0), ("map", 1), ("mat_node_rowptr", 0), ("tp_offtab", 0)]
        params = ["const int *__restrict__ layers", "double *__restrict__ arg0", "const double *__restrict__ arg1"] + cparams + [
                  "const int *__restrict__ map0", "const int *__restrict__ map1", "const fd_nnz_t *__restrict__ rp0",
                  "const unsigned short *__restrict__ tpo0"]
        if lg:
            layout += [("mat_row_lgmap", 0), ("mat_col_lgmap", 0)]
            params += ["const int *__restrict__ rlg0", "const int *__restrict__ clg0"]
        layout += [("tp_tables",)]
        params += ["const double *__restrict__ tptab"]
        body = (f"{cfdecl}\n  fdt::hex_qk_matrix<{targs}>(start, end, layers, arg0, arg1, cf, c1, map0, map1, rp0, tpo0, "
                f"{'rlg0, clg0' if lg else 'nullptr, nullptr'}, tptab, {call_w});")
        threads = geom["matrix_threads"]
        # 4 NT accumulator registers per lane: Q4 (NT = 8) fits three wavefronts per SIMD
        bounds = f"{threads}, 3" if geom["tiles"] <= 8 and threads == 256 else f"{threads}"
        if D > 1 and tensor_matrix_groups(geom, D) == geom["matrix_groups"]:
            # all D^2 blocks in one workgroup (8 NT D^2 accumulator registers, (Q2)^3: 144): two wavefronts per SIMD = 256 registers
            bounds = f"{threads}, 2"
    else:
        layout += [("arg", 0), ("arg", 1), ("arg", 2)] + clayout + [("map", 0), ("map", 1), ("tp_tables",)]
        params = ["const int *__restrict__ layers", "double *__restrict__ arg0", "const double *__restrict__ arg1",
                  "const double *__restrict__ arg2"] + cparams + ["const int *__restrict__ map0", "const int *__restrict__ map1",
                  "const double *__restrict__ tptab"]
        body = (f"{cfdecl}\n  fdt::hex_qk_action<{targs}>(start, end, layers, arg0, arg1, arg2, cf, c1, map0, map1, tptab, "
                f"{call_w});")
        threads = 128
        # (a vector-valued unknown or coefficient gradients carry D times / twice the lines in registers: no occupancy floor)
        # (... and from Q6 on the lines themselves are 7+ doubles per array)
        bounds = "128" + (f", {int(configuration['tp_action_waves'])}" if configuration["tp_action_waves"] and D == 1 and not grad and geom["k1"] <= 6 else "")
    src = head + [f'extern "C" __global__ __launch_bounds__({bounds}) void {sym}(int start, int end, {", ".join(params)})', "{", body, "}"]
    return WrapperSource("\n".join(src) + "\n", sym, "tp_" + kind, layout, 2, block_threads=threads, tp=geom)


def select_mode(gk: GlobalKernel) -> str:
    want = configuration["mode"]
    if want != "direct" and tensor_eligible(gk):
        return "tp_" + tensor_eligible(gk)
    ok = staged_eligible(gk)
    if want == "staged" and not ok:
        raise ValueError("FDHIP_MODE=staged but this parloop is not eligible for the staged wrapper")
    if want == "direct":
        return "direct"
    if configuration["mat_ocr"] and sliced_eligible(gk):
        return "ocrs"
    if configuration["mat_ocr"] and ocr_eligible(gk):
        return "ocr"
    return "staged" if ok else "direct"


def staged_eligible(gk: GlobalKernel, mats_on_virtual: bool = False, need_indirect_dat: bool = True) -> bool:
    """Staged wrapper (LDS gather / reduction over block-localisation plans).  Subsets and extruded sets qualify too: the
    plan is then built on a DERIVED map over the virtual iteration space -- the map rows of the subset's entities, resp.
    one row ``map + offset*layer`` per (column, layer) cell (builder.py:94-124 folded into the table once) -- so the kernel
    itself addresses nothing but local indices; only direct arguments need the base entity.  Restrictions: Dat-only
    loops, regions ALL / ON_BOTTOM / ON_TOP (constant, variable or periodic layers), and no direct Dat written on an
    extruded set (all layers of a column share its row: parloop.py:494-497)."""
    if gk._extruded or gk._subset:
        if any(isinstance(a, MatKernelArg) for a in gk.arguments) and not mats_on_virtual:
            return False          # (matrix loops over virtual spaces: row-sliced owner-computes-rows only, sliced_eligible)
        if gk._extruded:
            if gk._iteration_region == ON_INTERIOR_FACETS and (
                    (not mats_on_virtual and any(isinstance(a, MatKernelArg) for a in gk.arguments))
                    or any(isinstance(m, PermutedMapKernelArg) for a in gk.arguments for m in (getattr(a, "maps", None) or ()))):
                return False          # (variable layers qualify: the derived map and the cell tables are ragged, set.py:326-337;
                                      #  periodic columns too: the wrap of builder.py:101-123 is folded into the derived map's rows;
                                      #  interior facets -- of constant- and, since round 6, variable-layer columns -- in Dat loops: a
                                      #  derived row holds both stacked cells)
            if any(isinstance(a, DatKernelArg) and not a.is_indirect and la.access != READ
                   for a, la in zip(gk.arguments, gk.local_kernel.arguments)):
                return False
    n_ind = 0
    for a, la in zip(gk.arguments, gk.local_kernel.arguments):
        if isinstance(a, MatKernelArg) and any(isinstance(m, PermutedMapKernelArg) for m in a.maps):
            return False          # matrix plans / row-offset tables are built on the base maps
        if isinstance(a, DatKernelArg) and a.is_indirect:
            if a.index is not None:
                return False
            if stages_in_lds(la.access, la.dtype):
                n_ind += 1
                continue
            # WRITE / RW / MIN / MAX through a map (an interpolation's output, a limiter's bounds): the lane reads and writes global
            # memory itself while the READ / INC arguments of the loop are staged -- on plain sets, subsets and constant-layer
            # extruded sets (cell regions; the lane applies the layer arithmetic of builder.py:94-124 to the base entity's map row as
            # the direct wrapper does), and not in matrix loops
            # (round 6: variable layers and interior facets too -- the lane takes the column's own bottom from the layers array and
            # walks both stacked cells of a facet; periodic columns of variable height stay direct)
            if any(isinstance(b, MatKernelArg) for b in gk.arguments) or (
                    gk._extruded and not gk._constant_layers and gk._extruded_periodic):
                return False
    return n_ind > 0 or not need_indirect_dat


def _ocr_shape(gk: GlobalKernel, mats_on_virtual: bool = False, allow_unroll: bool = False):
    """The Mat argument of a loop that can assemble by owner-computes-rows -- its only output is ONE Mat with INC access
    (addressed per node unless ``allow_unroll``), everything else READ (entities are visited redundantly, so nothing else
    may be modified) -- or None."""
    if not staged_eligible(gk, mats_on_virtual, need_indirect_dat=not mats_on_virtual):
        return None                 # (the sliced wrapper -- the only caller with mats_on_virtual -- needs no staged Dat)
    mats = []
    for a, la in zip(gk.arguments, gk.local_kernel.arguments):
        if isinstance(a, MatKernelArg):
            if la.access != INC or (a.unroll and not allow_unroll):
                return None
            mats.append(a)
        elif la.access != READ:
            return None
    return mats[0] if len(mats) == 1 else None


def ocr_eligible(gk: GlobalKernel) -> bool:
    """Owner-computes-rows matrix assembly with whole-entity instances: scalar blocks, per-node lgmaps; also over subsets and
    extruded sets (constant layers, regions ALL / ON_BOTTOM / ON_TOP: the plans are built on derived maps)."""
    a = _ocr_shape(gk, mats_on_virtual=True)
    if gk._extruded and gk._iteration_region == ON_INTERIOR_FACETS:
        return False              # (two stacked cells per trip: the row-sliced wrapper takes them as maps of twice the arity)
    # (the element tensor and its row-offset table live in registers: bounded element matrices only)
    return a is not None and int(np.prod(a.dims[0])) * int(np.prod(a.dims[1])) == 1 \
        and a.maps[0].arity * a.maps[1].arity <= configuration["ocr_sliced_max_entries"]


def sliced_eligible(gk: GlobalKernel) -> bool:
    """Row-sliced owner-computes-rows (generate_sliced_wrapper): an owner-computes-rows loop whose element matrix has at
    least ``ocr_sliced_min_arity`` scalar rows (row-map arity x row block size) -- the size from which one node's rows cost
    much less than the whole (P2 tets: 10 rows; measured on the P2 stiffness kernel: 233 fp64 instructions for one row
    against 599 for all ten, i.e. the rows share little beyond the geometry, while an unsliced row block recomputes whole
    entities x2.2-3.4).  The threshold is a proxy for "the rows dominate what they share"; measured: P1 tets (4 rows) 2.09 ms
    sliced against 1.03 whole, Q1 hexahedra with the geometry at 8 Gauss points (8 rows) 1.98 against 2.69, P2 tets (10
    rows) 1.10 against 1.98.  Vector-valued blocks (MatSetValuesBlockedLocal, builder.py:573-625) are sliced per NODE: an
    instance owns the ``rbs`` scalar rows of one node."""
    a = _ocr_shape(gk, mats_on_virtual=True, allow_unroll=True) if configuration["ocr_sliced"] else None
    if a is None:
        return False
    if a.unroll and (int(np.prod(a.dims[0])) > 8 or a.maps[1].arity * int(np.prod(a.dims[1])) > 64):
        return False          # per-dof lgmaps (MatSetValuesLocal on dof indices) travel as an 8-bit row and a 64-bit column mask
    # the local kernel is inlined once per row-map entry and must unroll completely in each copy: bounded element matrices
    # only (Q2 hexahedra 27x27 and vector P2 tetrahedra 30x30 qualify; the 125x125 of Q4 has its own wrapper, else direct)
    nf = 2 if (gk._extruded and gk._iteration_region == ON_INTERIOR_FACETS) else 1      # (a facet sees two stacked cells)
    entries = nf * a.maps[0].arity * nf * a.maps[1].arity * int(np.prod(a.dims[0])) * int(np.prod(a.dims[1]))
    if nf * a.maps[0].arity > configuration["ocr_sliced_max_arity"] or entries > configuration["ocr_sliced_max_entries"]:
        return False
    if nf > 1 or not ocr_eligible(gk):
        # vector-valued blocks, per-dof lgmaps: whole-entity instances do not cover them, and the only other path scatters
        # every entry with a global atomic -- sliced whatever the size of the element matrix
        return True
    return a.maps[0].arity * int(np.prod(a.dims[0])) >= configuration["ocr_sliced_min_arity"]


def _hoist_includes(code: str):
    from .kernel import strip_host_only_includes
    code = strip_host_only_includes(code)
    inc = re.findall(r"^\s*#\s*include[^\n]*$", code, flags=re.M)
    body = re.sub(r"^\s*#\s*include[^\n]*$", "", code, flags=re.M)
    return inc, body


def generate_wrapper(gk: GlobalKernel, mode: str, min_waves: int = 0) -> WrapperSource:
    if mode.startswith("tp_"):
        return generate_tensor_wrapper(gk)
    if mode.startswith("ocrs"):
        return generate_sliced_wrapper(gk, mode)
    lk = gk.local_kernel
    maps, map_index = _distinct_maps(gk)
    # requires_zeroed_output_arguments: MIN/MAX packs start from zero like INC/WRITE ones (builder.py:276-279, 368-371)
    also_zero = (MIN, MAX) if lk.requires_zeroed_output_arguments else ()
    full_mode = mode
    # "_fx": the LDS accumulators of a whole-entity owner-computes-rows loop are CHECKED 64-bit fixed-point sums (integer LDS
    # atomics instead of ds_add_f64; fd_wrapper.h, fx_block_t): every row block keeps its own scale record, and a block whose
    # largest contribution leaves the window of its scale redoes its rows in fp64 inside the same launch
    fx = mode.endswith("_fx")
    if fx:
        mode = mode[:-3]
    # "<mode>_s<S0>x<S1>...": compile-time node strides of the staged maps (in staged_maps order), see lds_stride()
    sm_ = re.search(r"_s(\d+(?:x\d+)*)$", mode)
    strides = [int(v) for v in sm_.group(1).split("x")] if sm_ else None
    if sm_:
        mode = mode[:sm_.start()]
    # "_d<m0>x<m1>...": maps whose staged rows would see NO reuse inside a block (every node touched by one entity: cell loops of
    # discontinuous spaces) -- their READ / INC arguments are gathered and scattered from the lane through the map row, like the
    # WRITE / RW / MIN / MAX arguments beside staged ones, instead of passing through LDS (Parloop._staged_geometry decides)
    dm_ = re.search(r"_d(\d+(?:x\d+)*)$", mode) if mode.startswith("staged") else None
    direct_maps = set(int(v) for v in dm_.group(1).split("x")) if dm_ else set()
    if dm_:
        mode = mode[:dm_.start()]

    def in_lds(info):
        return stages_in_lds(info["acc"], info["dtype"]) and info.get("m") not in direct_maps
    # "_q<L0>x<L1>..k<K>[d]": the per-instance index rows of an owner-computes-rows loop arrive as ONE bit-packed record per
    # instance (fd_ocr_pack_records): local-map entries of staged map m at L_m bits, row offsets at K bits; "d" = the offsets
    # of the diagonal entries (i, i) are not stored -- they are a property of the row node and ride in its LDS word
    rq_ = re.search(r"_q(\d+(?:x\d+)*)k(\d+)(d?)$", mode)
    rec = None
    if rq_:
        rec = {"lbits": [int(v) for v in rq_.group(1).split("x")], "kbits": int(rq_.group(2)), "diag": bool(rq_.group(3))}
        mode = mode[:rq_.start()]
    ocr = mode.startswith("ocr")
    # "stagedo": staged over a backend-derived entity ORDER (fd_locality_order): slot -> entity through fd_order_, plans on
    # the map rows gathered in that order (Parloop._staged_geometry, un-hinted maps)
    ordered = mode.startswith("stagedo")
    # "ocrp": owner-computes-rows over row POSITIONS of a backend-derived row order (fd_first_touch_order): a block's rows
    # are a set of CSR rows -- accumulated contiguously in LDS, flushed row by row
    ocrp = mode.startswith("ocrp")
    # "ocrpm": the column lgmap of the Mat is folded into the flush's place table (fd_row_entry_positions_masked) instead of a select
    # per contribution in the main loop
    ocrpm = mode.startswith("ocrpm")
    staged = mode.startswith("staged") or ocr
    ktype, kbytes = ("unsigned short", 2) if mode.endswith("_k16") else ("unsigned char", 1)
    extruded = gk._extruded
    periodic = bool(extruded and gk._extruded_periodic)
    varlay = bool(extruded and not gk._constant_layers)      # per-entity [bottom, top) rows (set.py:326-337)
    region = gk._iteration_region
    ih = extruded and region == ON_INTERIOR_FACETS
    nf = 2 if ih else 1
    threads = configuration["block_threads"]
    if not threads:
        # staged loops over high-arity maps (P2: 10 nodes per cell, ~2600 nodes per block) stage and flush several
        # nodes per lane: 512-lane groups halve those phases (P2 residual 0.48 -> 0.37 ms); P1 is best at 256
        threads = 512 if (mode.startswith("staged") and max((m.arity for m in maps), default=0) >= 8) else 256
    if mode.startswith("ocr"):
        if configuration["ocr_block_threads"]:
            threads = configuration["ocr_block_threads"]
        else:
            threads = 512     # 8 wavefronts amortise the per-block staging and flush phases (256 VGPRs per lane still fit)
        # Larger element matrices kept whole (5..7 rows, or FDHIP_OCR_SLICED=0) have long rows, so a 64 KiB row block owns few
        # rows and most of its instances are border entities computed again by the neighbours.  Give such loops the whole
        # CU's LDS: one 512-lane group per CU, ~2x the rows per block (P2 Jacobian unsliced: 2.57 -> 1.87 ms; sliced it
        # takes 1.1 ms, generate_sliced_wrapper).  Small element matrices (P1: 4x4) do better with three 47 KiB groups
        # per CU overlapping their phases.
        entries = max(int(np.prod(a.dims[0])) * int(np.prod(a.dims[1])) * a.maps[0].arity * a.maps[1].arity
                      for a in gk.arguments if isinstance(a, MatKernelArg))
        ocr_lds_limit = configuration["ocr_lds_limit"] or (159 * 1024 if entries > 32 else 0)

    params: List[str] = []
    layout: List[tuple] = []

    def P(decl, desc):
        params.append(decl)
        layout.append(desc)

    if extruded:
        P("const int *__restrict__ layers", ("layers",))
    if gk._subset:
        P("const int *__restrict__ subset_indices", ("subset",))

    # ---- classify arguments
    infos = []
    for k, (a, la) in enumerate(zip(gk.arguments, lk.arguments)):
        ct = CTYPE[np.dtype(la.dtype)]
        info = {"k": k, "arg": a, "acc": la.access, "ct": ct, "dtype": np.dtype(la.dtype)}
        if isinstance(a, DatKernelArg):
            info["kind"] = "dat"
            info["c"] = int(np.prod(a.dim))
            # DatView (dat.py:714-805, builder.py:347-349): the kernel sees ONE component of every node; `vi` is its flat
            # position inside the node's row, the row stride stays the parent's
            info["vi"] = None if a.index is None else int(np.ravel_multi_index(tuple(a.index), tuple(a.dim)))
            if a.is_indirect:
                m = a.map_
                base = m.base_map if isinstance(m, PermutedMapKernelArg) else m
                info["m"] = map_index[id(base)]
                info["ar"] = base.arity
                info["perm"] = tuple(m.permutation) if isinstance(m, PermutedMapKernelArg) else None
                info["off"] = base.offset if extruded else None
        elif isinstance(a, GlobalKernelArg):
            info["kind"] = "global"
            info["n"] = int(np.prod(a.dim))
        elif isinstance(a, MatKernelArg):
            info["kind"] = "mat"
            (rdim, cdim) = a.dims
            info["rbs"], info["cbs"] = int(np.prod(rdim)), int(np.prod(cdim))
            rm, cm = a.maps
            info["rperm"] = tuple(rm.permutation) if isinstance(rm, PermutedMapKernelArg) else None
            info["cperm"] = tuple(cm.permutation) if isinstance(cm, PermutedMapKernelArg) else None
            rm = rm.base_map if isinstance(rm, PermutedMapKernelArg) else rm
            cm = cm.base_map if isinstance(cm, PermutedMapKernelArg) else cm
            info["rm"], info["cm"] = map_index[id(rm)], map_index[id(cm)]
            info["ar"], info["ac"] = rm.arity, cm.arity
            info["roff"] = rm.offset if extruded else None
            info["coff"] = cm.offset if extruded else None
            if la.access not in (INC, WRITE):
                raise ValueError("Mat arguments must be INC or WRITE")
        elif isinstance(a, PassthroughKernelArg):
            info["kind"] = "pass"
        else:
            raise TypeError(f"unsupported kernel argument {a!r}")
        infos.append(info)

    for info in infos:
        k, ct = info["k"], info["ct"]
        const = "const " if info["acc"] == READ and info["kind"] != "mat" else ""
        P(f"{const}{ct} *__restrict__ arg{k}" if info["kind"] != "pass" else f"void *arg{k}", ("arg", k))
    for mi in range(len(maps)):
        P(f"const int *__restrict__ map{mi}", ("map", mi))

    # ---- backend-private parameters
    P("const int *__restrict__ bstart_", ("bstart",))
    if ocr:
        P("const int *__restrict__ inst_ent_", ("ocr_inst_ent",))
    if ordered:
        P("const int *__restrict__ fd_order_", ("order",))
    # variable layers: (position, layer) of every cell of the virtual iteration space, read where a direct argument or the layer
    # argument needs them
    vtab = bool(staged and varlay and (gk._pass_layer_arg or any(i_["kind"] == "dat" and ("m" not in i_ or not in_lds(i_)) for i_ in infos)))
    if vtab:
        P("const int *__restrict__ fd_vcol_", ("virt_col",))
        P("const int *__restrict__ fd_vlay_", ("virt_layer",))
    staged_maps = []
    lds_items = []
    mat_staged = {}
    for info in infos:
        if info["kind"] == "mat":
            mat_staged[info["k"]] = bool(staged and not ocr and configuration["mat_staged"] and info["rbs"] * info["cbs"] == 1
                                         and info["acc"] == INC and not info["arg"].unroll
                                         and info["ar"] * info["ac"] <= configuration["ocr_sliced_max_entries"])
    if staged:
        for info in infos:
            if info["kind"] == "dat" and "m" in info and in_lds(info) and info["m"] not in staged_maps:
                staged_maps.append(info["m"])
            if info["kind"] == "mat" and (mat_staged[info["k"]] or ocr):
                for mi in (info["rm"], info["cm"]):
                    if mi not in staged_maps:
                        staged_maps.append(mi)
        for mi in staged_maps:
            P(f"const int *__restrict__ p{mi}_blkoff", ("plan_blkoff", mi))
            P(f"const int *__restrict__ p{mi}_list", ("plan_list", mi))
            P(f"const unsigned short *__restrict__ p{mi}_lmap", ("plan_lmap", mi))
            P(f"long long p{mi}_maxnd", ("plan_maxnd", mi))
    def stage_unroll(mi):
        """nodes a lane stages (and flushes) per batch: ceil(nodes per block / lanes), from the compile-time node stride of the map;
        owner-computes-rows blocks (run-time strides, 600-1000 nodes for 512 lanes) take 2; 1 = the one-node-per-trip loops"""
        if not configuration["stage_batch"]:
            return 1
        if ocr:
            # owner-computes-rows blocks keep one node per trip: their staging depends on no node id (plan-ordered tables and
            # copies), and two nodes per lane measured 1-4 % SLOWER on the P1 Jacobian (profiles/r6q_ab_stage_batch.txt)
            return 1
        if strides is None or mi not in staged_maps or len(strides) != len(staged_maps):
            return 1
        # (a batch keeps its rows in registers between the loads and the LDS stores: at most ~32 doubles per lane)
        words = 1 + sum(i_["c"] for i_ in infos if i_["kind"] == "dat" and i_.get("m") == mi and i_["acc"] == READ and in_lds(i_))
        return max(1, min(int(configuration["stage_batch"]), 32 // words, -(-strides[staged_maps.index(mi)] // threads)))

    def srow_table(info):
        """whole-entity owner-computes-rows, row map = column map: the per-node row words come from a plan-ordered table
        (fd_ocr_node_words) instead of per-node gathers of a row start and two lgmap entries"""
        return bool(ocr and info["rm"] == info["cm"])

    use_table = {}
    for info in infos:
        if info["kind"] != "mat":
            continue
        k = info["k"]
        # (the element->nonzero table is built on the base maps: a Mat reached through PermutedMaps searches its rows)
        table = (configuration["mat_scatter"] == "table") and not extruded and info["rperm"] is None and info["cperm"] is None
        use_table[k] = table
        if ocr:
            P(f"const int *__restrict__ oc{k}_rblk", ("ocr_rblk", k))
            P(f"const fd_nnz_t *__restrict__ oc{k}_rowptr", ("ocr_rowptr", k))
            P(f"const {ktype} *__restrict__ oc{k}_k", ("ocr_kidx", k))
            P(f"long long oc{k}_maxnnz", ("ocr_maxnnz", k))
            P(f"long long oc{k}_maxnown", ("ocr_maxnown", k))
            P(f"long long oc{k}_flags", ("ocr_flags", k))
            if srow_table(info):
                P(f"const unsigned int *__restrict__ oc{k}_srowtab", ("ocr_srow", k, info["rm"]) + (("diag",) if (rec and rec["diag"]) else ()))
            if ocrp:
                P(f"const fd_nnz_t *__restrict__ oc{k}_prowptr", ("ocr_prowptr", k))
                P(f"const fd_nnz_t *__restrict__ oc{k}_nstart", ("ocr_nstart", k))
                P(f"const int *__restrict__ oc{k}_gpos", ("ocr_gpos", k))
                P(f"long long oc{k}_npos", ("ocr_npos", k))
            if rec:
                P(f"const unsigned int *__restrict__ oc{k}_rec", ("ocr_rec", k))
            if fx:
                P(f"fdw::fx_block_t *__restrict__ fx{k}_scale", ("fx_scale", k))
                P(f"unsigned int *__restrict__ fx{k}_stat", ("fx_stat", k))
        elif mat_staged[k]:
            P(f"const int *__restrict__ mp{k}_off", ("matplan_off", k))
            P(f"const int *__restrict__ mp{k}_gpos", ("matplan_gpos", k))
            P(f"const int *__restrict__ mp{k}_lrp", ("matplan_lrp", k))
            P(f"const {ktype} *__restrict__ mp{k}_k", ("matplan_kidx", k))
            P(f"long long mp{k}_maxnnz", ("matplan_maxnnz", k))
            P(f"long long mp{k}_flags", ("matplan_flags", k))
        elif table:
            P(f"const int *__restrict__ tab{k}", ("mat_table", k))
            if info["rbs"] * info["cbs"] != 1:
                P(f"const fd_nnz_t *__restrict__ nrp{k}", ("mat_node_rowptr", k))
        else:
            P(f"const fd_nnz_t *__restrict__ rp{k}", ("mat_rowptr", k))
            P(f"const int *__restrict__ ci{k}", ("mat_colidx", k))
        if info["arg"].lgmaps and not (ocr and srow_table(info)):
            P(f"const int *__restrict__ rlg{k}", ("mat_row_lgmap", k))
            P(f"const int *__restrict__ clg{k}", ("mat_col_lgmap", k))

    # profiling aid (FDHIP_PHASE_TIMES=1, tools/phase_times.py): lane 0 of every block stores the 100 MHz wall clock at the start, after
    # the staging barrier, after the main loop's barrier and at the end, plus the hardware id of its wavefront
    ptimes = bool(ocr and configuration["phase_times"])
    if ptimes:
        P("long long *__restrict__ fd_times", ("phase_times",))

    # ---- static tables (offsets / permutations)
    decls = []
    for mi, m in enumerate(maps):
        if extruded and m.offset is not None:
            decls.append(f"__device__ static const int map{mi}_off[{m.arity}] = {{{', '.join(str(int(o)) for o in m.offset)}}};")
            if periodic and m.offset_quotient is not None:
                decls.append(f"__device__ static const int map{mi}_quot[{m.arity}] = {{{', '.join(str(int(o)) for o in m.offset_quotient)}}};")

    def node(mi, ar, i, off, perm=None, f="0", ent="e"):
        ii = _permi(perm, i)
        e = f"map{mi}[(size_t){ent}*{ar} + {ii}]"
        if mi in direct_maps:
            # "_d" maps are AFFINE (Parloop._staged_geometry checked map[e][i] == arity*e + i): no index load at all, and the rows of
            # an entity are contiguous -- one wide load / a run of adjacent atomics per lane
            e = f"({ent}*{ar} + {ii})"
        if extruded and off is not None:
            # a permuted map permutes its offsets (and quotients) with its values (builder.py:160-169)
            rel = f"(layer - lay[0] + {f})"
            if periodic:
                # builder.py:101-123: the layer offset wraps around the column of fd_nl cell layers
                if maps[mi].offset_quotient is None:
                    rel = f"fdw::wrap_layer({rel}, fd_nl)"
                else:
                    rel = (f"(fdw::wrap_layer({rel} + map{mi}_quot[{ii}], fd_nl) - "
                           f"fdw::wrap_layer(map{mi}_quot[{ii}], fd_nl))")
            e = f"({e} + map{mi}_off[{ii}]*{rel})"
        return e

    pre, pack, call_args, unpack, post = [], [], [], [], []
    direct_prefetch = []       # (argument, C type, values per entity) of READ arguments on affine "_d" maps
    unpack_fx = []       # "_fx": the fixed-point trip's unpack of the Mat (everything else in such a loop is READ)
    flush_pre_decl, flush_pre = [], []     # table flushes: registers of the first batch of places, and its loads (ahead of the barrier)
    node_actions = {}    # per staged map: [(load statements, LDS store statements)] templated on I_U / G_U
    lds_decl, stage, flush, mat_stage_pre = [], [], [], []
    # LDS carve-up order: staged Dat rows, then the per-node matrix tables (sizes fixed by the node strides), then the
    # matrix accumulators (sized by the block's nonzero count, known only at run time) -- with compile-time strides
    # every array then starts at a compile-time offset
    lds_tail_const, lds_tail_var = [], []

    # LDS carving for staged args
    if staged:
        lds_decl.append("size_t fd_off = 0;")

    for info in infos:
        k, ct, acc = info["k"], info["ct"], info["acc"]
        if info["kind"] == "pass":
            call_args.append(f"arg{k}")
        elif info["kind"] == "global":
            n = info["n"]
            if acc == READ:
                call_args.append(f"const_cast<{ct} *>(arg{k})")
                continue
            ident = {INC: "0", WRITE: "0", RW: f"arg{k}[q]", MIN: f"arg{k}[q]", MAX: f"arg{k}[q]"}[acc]
            pre.append(f"{ct} g{k}[{n}]; for (int q = 0; q < {n}; ++q) g{k}[q] = {ident};")
            if acc == INC:
                # private zeroed pack per entity then += (builder.py:292-319)
                pack.append(f"{ct} t{k}[{n}]; for (int q = 0; q < {n}; ++q) t{k}[q] = 0;")
                unpack.append(f"for (int q = 0; q < {n}; ++q) g{k}[q] += t{k}[q];")
                call_args.append(f"t{k}")
                op, at = "OpAdd", "atomic_add"
            elif acc in (MIN, MAX):
                op, at = ("OpMin", "atomic_min") if acc == MIN else ("OpMax", "atomic_max")
                if acc in also_zero:
                    # the kernel sees a zeroed pack per entity, combined into the running value afterwards
                    pack.append(f"{ct} t{k}[{n}]; for (int q = 0; q < {n}; ++q) t{k}[q] = 0;")
                    unpack.append(f"for (int q = 0; q < {n}; ++q) g{k}[q] = fdw::{op}<{ct}>::f(g{k}[q], t{k}[q]);")
                    call_args.append(f"t{k}")
                else:
                    call_args.append(f"g{k}")
            else:
                raise ValueError("Global arguments may be READ, INC, MIN or MAX in a parloop")
            post.append(f"for (int q = 0; q < {n}; ++q) {{ {ct} r = fdw::block_reduce<{ct}, fdw::{op}<{ct}>>(g{k}[q], ({ct} *)fd_red); "
                        f"if (threadIdx.x == 0) fdw::{at}<{ct}>(&arg{k}[q], r); }}")
        elif info["kind"] == "dat" and "m" not in info:
            c = info["c"]
            cast = f"const_cast<{ct} *>" if acc == READ else ""
            call_args.append(f"{cast}(&arg{k}[(size_t)e*{c}" + (f" + {info['vi']}" if info["vi"] is not None else "") + "])")
        elif info["kind"] == "dat":
            c, ar, mi = info["c"], info["ar"], info["m"]
            perm, off = info["perm"], info["off"]
            size = nf * ar * c
            pack.append(f"{ct} t{k}[{size}];")
            if staged and in_lds(info):
                lds_items.append(("dat", mi, c, info["dtype"].itemsize, acc != READ))
                lds_decl.append(f"{ct} *s{k} = ({ct} *)(fd_lds + fd_off); fd_off += (((size_t)p{mi}_maxnd*{c}*sizeof({ct})) + 15) & ~(size_t)15;")
                if acc == READ:
                    soa = c > 1            # component-major LDS layout for vector Dats (conflict-free lane strides)
                    # a READ Dat that has not changed since an earlier call is also kept in PLAN order (one row per (block, staged
                    # node), Parloop._plan_copy): the staging phase then streams it instead of gathering rows by node id
                    P(f"const {ct} *__restrict__ pl{k}", ("plan_copy", k, mi))
                    node_actions.setdefault(mi, []).append(
                        ([f"{ct} v{k}_U[{c}];",
                          # (uniform condition, streaming branch, gathering branch): a batched staging loop hoists the condition
                          # over the batch, so the streamed rows are requested without waiting for the node ids
                          ("IF", f"pl{k}", f"for (int j = 0; j < {c}; ++j) v{k}_U[j] = pl{k}[(size_t)(l0_{mi} + I_U)*{c} + j];",
                           f"for (int j = 0; j < {c}; ++j) v{k}_U[j] = arg{k}[(size_t)G_U*{c} + j];")],
                         [f"for (int j = 0; j < {c}; ++j) s{k}[{'j*(int)p%d_maxnd + I_U' % mi if soa else 'I_U*%d + j' % c}] = v{k}_U[j];"]))
                    if nf == 1:
                        idx = f"j*(int)p{mi}_maxnd + lm{mi}[{_permi(perm, 'i')}]" if soa else f"lm{mi}[{_permi(perm, 'i')}]*{c} + j"
                        pack.append(f"for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) t{k}[i*{c}+j] = s{k}[{idx}];")
                    else:
                        li = f"lm{mi}[f*{ar} + {_permi(perm, 'i')}]"
                        idx = f"j*(int)p{mi}_maxnd + {li}" if soa else f"{li}*{c} + j"
                        pack.append(f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) "
                                    f"t{k}[(f*{ar}+i)*{c}+j] = s{k}[{idx}];")
                else:  # INC
                    stage.append((mi, f"for (int q = tid; q < nd{mi}*{c}; q += nthr) s{k}[q] = 0;"))
                    pack.append(f"for (int q = 0; q < {size}; ++q) t{k}[q] = 0;")
                    if nf == 1:
                        unpack.append(f"for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) "
                                      f"atomicAdd(&s{k}[lm{mi}[{_permi(perm, 'i')}]*{c} + j], t{k}[i*{c}+j]);")
                    else:
                        unpack.append(f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) "
                                      f"atomicAdd(&s{k}[lm{mi}[f*{ar} + {_permi(perm, 'i')}]*{c} + j], t{k}[(f*{ar}+i)*{c}+j]);")
                    fu = stage_unroll(mi)
                    if fu == 1:
                        flush.append((mi, f"for (int q = tid; q < nd{mi}*{c}; q += nthr) {{ const int i = q / {c}; "
                                          f"fdw::atomic_add<{ct}>(&arg{k}[(size_t)p{mi}_list[l0_{mi} + i]*{c} + (q - i*{c})], "
                                          f"s{k}[q]); }}"))
                    else:
                        # the node ids the flush scatters through are requested BEFORE the barrier that ends the main loop (a batch
                        # per lane, like the places of the owner-computes-rows flush): the flush itself waits for no load
                        ld = (f"for (int f = 0; f < {fu}; ++f) {{ const int q = Q0 + f*nthr; "
                              f"fl{k}[f] = q < nd{mi}*{c} ? p{mi}_list[l0_{mi} + q / {c}] : 0; }}")
                        flush_pre_decl.append(f"int fl{k}[{fu}];")
                        flush_pre.append(ld.replace("Q0", "tid"))
                        flush.append((mi, f"for (int q0 = tid; q0 < nd{mi}*{c}; q0 += {fu}*nthr) {{ "
                                          f"if (q0 >= {fu}*nthr) {{ {ld.replace('Q0', 'q0')} }} "
                                          f"for (int f = 0; f < {fu}; ++f) {{ const int q = q0 + f*nthr; if (q < nd{mi}*{c}) "
                                          f"fdw::atomic_add<{ct}>(&arg{k}[(size_t)fl{k}[f]*{c} + (q - (q / {c})*{c})], s{k}[q]); }} }}"))
                call_args.append(f"t{k}")
                continue
            nexpr = node(mi, ar, "i", off, perm, "f")
            if info["vi"] is not None:
                # a view packs one value per node: t[f][i] <-> dat[node][vi]
                loop = f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i)"
                lhs = f"arg{k}[(size_t){nexpr}*{c} + {info['vi']}]"
                rhs = f"t{k}[f*{ar}+i]"
                vsize = nf * ar
            else:
                loop = f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j)"
                lhs = f"arg{k}[(size_t){nexpr}*{c} + j]"
                rhs = f"t{k}[(f*{ar}+i)*{c}+j]"
                vsize = size
            if acc in (INC, WRITE) + also_zero:
                pack.append(f"for (int q = 0; q < {vsize}; ++q) t{k}[q] = 0;")
            elif (mi in direct_maps and acc == READ and info["vi"] is None and nf == 1 and staged and configuration["prefetch"]
                  and not (extruded or gk._subset or mode.startswith("stagedo"))):
                # rows of an affine map ride in the software pipeline like the index rows: the NEXT entity's values are requested
                # before the current entity's local kernel runs (direct_prefetch: consumed below, once idx_loads exists)
                direct_prefetch.append((k, ct, ar * c))
                pack.append(f"for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) t{k}[i*{c}+j] = dv{k}[{_permi(perm, 'i')}*{c} + j];")
            else:
                pack.append(f"{loop} {rhs} = {lhs};")
            call_args.append(f"t{k}")
            if acc == INC and mi in direct_maps:
                # an affine map is injective: no other entity of this launch touches these rows, so += is a plain read-modify-write of
                # the entity's contiguous rows (wide loads and stores) instead of one atomic per scalar
                unpack.append(f"{loop} {lhs} += {rhs};")
            elif acc == INC:
                unpack.append(f"{loop} fdw::atomic_add<{ct}>(&{lhs}, {rhs});")
            elif acc == MIN:
                unpack.append(f"{loop} fdw::atomic_min<{ct}>(&{lhs}, {rhs});")
            elif acc == MAX:
                unpack.append(f"{loop} fdw::atomic_max<{ct}>(&{lhs}, {rhs});")
            elif acc in (WRITE, RW):
                unpack.append(f"{loop} {lhs} = {rhs};")
        elif info["kind"] == "mat":
            ar, ac, rbs, cbs = info["ar"], info["ac"], info["rbs"], info["cbs"]
            rm, cm = info["rm"], info["cm"]
            size = nf * ar * rbs * nf * ac * cbs
            pack.append(f"double t{k}[{size}]; for (int q = 0; q < {size}; ++q) t{k}[q] = 0;")
            call_args.append(f"t{k}")
            lg = info["arg"].lgmaps
            if ocr:
                lds_items.append(("ocr", k, rm, cm, bool(lg)))
                lds_tail_var.append(f"double *sm{k} = (double *)(fd_lds + fd_off); fd_off += (((size_t)oc{k}_maxnnz*8) + 15) & ~(size_t)15;")
                # one LDS word per gathered node: bits 0..29 = 1 + offset of the node's row inside the block's
                # accumulator (0 = row not owned here or BC-masked), bit 31 = column is BC-masked
                lds_tail_const.append(f"unsigned int *srow{k} = (unsigned int *)(fd_lds + fd_off); fd_off += (((size_t)p{rm}_maxnd*4) + 15) & ~(size_t)15;")
                if cm != rm:
                    lds_tail_const.append(f"unsigned char *smc{k} = fd_lds + fd_off; fd_off += ((size_t)p{cm}_maxnd + 15) & ~(size_t)15;")
                rp_ = f"oc{k}_prowptr" if ocrp else f"oc{k}_rowptr"       # row starts in the order the blocks are cut in
                mat_stage_pre.extend([f"const int n0_{k} = oc{k}_rblk[b], nown{k} = oc{k}_rblk[b+1] - n0_{k};",
                                      f"const fd_nnz_t r0_{k} = {rp_}[n0_{k}]; const int nnzb{k} = (int)({rp_}[n0_{k} + nown{k}] - r0_{k});"])
                # (by capacity when the index rows are requested early: the block's row starts -- a second level of dependent scalar
                # loads -- are then needed by the flush only, and nothing ahead of the main loop waits for them)
                zcount = f"(int)oc{k}_maxnnz" if configuration["prefetch"] else f"nnzb{k}"
                stage.append((rm, f"for (int q = tid; q < {zcount}; q += nthr) sm{k}[q] = 0;"))
                # column masking (BC columns, pyop2/parloop.py:279-302) stays in the loop: a masked contribution adds 0.0.  Moving
                # it to the row flush (one bit per CSR entry) removes 48 VALU instructions per instance and is NOT faster --
                # the kernel is bound by the LDS pipe (profiles/r3b_ab_colmask.txt: 0.968 vs 0.952 ms tiled, 1.31 vs 1.19 un-hinted)
                colmask = bool(lg) and not ocrpm
                if ocrpm and (fx or not lg or cm != rm):
                    raise ValueError('"ocrpm" serves fp64 accumulation of a Mat with lgmaps and one map on both sides')
                rowmask = f" && rlg{k}[g] >= 0" if lg else ""
                colbit = (f" | ((clg{k}[g] < 0) ? 0x80000000u : 0u)" if (colmask and cm == rm) else "")
                if srow_table(info):
                    loads = [f"const unsigned w{k}_U = oc{k}_srowtab[l0_{rm} + I_U];"]
                elif ocrp:
                    # the accumulator offset of the node's row (by NODE) decides ownership too: the block's rows are exactly
                    # those whose offsets fall into [r0, r0 + nnzb) -- one lookup, no row-position table in the kernel
                    loads = [f"const fd_nnz_t p{k}_U = (G_U < (int)oc{k}_npos) ? oc{k}_nstart[G_U] - r0_{k} : -1;",
                             f"const unsigned w{k}_U = ((p{k}_U >= 0 && p{k}_U < nnzb{k}{rowmask.replace('[g]', '[G_U]')}) ? "
                             f"(unsigned)(p{k}_U + 1) : 0u){colbit.replace('[g]', '[G_U]')};"]
                else:
                    rowpos = f"(unsigned)(oc{k}_rowptr[g] - r0_{k} + 1)"
                    loads = [f"const int g = G_U; const unsigned w{k}_U = ((g >= n0_{k} && g < n0_{k} + nown{k}{rowmask}) ? {rowpos} : 0u){colbit};".replace("const int g = G_U; ", "").replace("(g ", "(G_U ").replace(" g ", " G_U ").replace(" g,", " G_U,").replace("[g]", "[G_U]")]
                node_actions.setdefault(rm, []).append((loads, [f"srow{k}[I_U] = w{k}_U;"]))
                if colmask and cm != rm:
                    node_actions.setdefault(cm, []).append(([f"const bool m{k}_U = clg{k}[G_U] < 0;"], [f"smc{k}[I_U] = m{k}_U;"]))
                lines = [f"unsigned int rw{k}[{ar}];", f"for (int i = 0; i < {ar}; ++i) rw{k}[i] = srow{k}[lm{rm}[i]];"]
                if rec and rec["diag"]:
                    # (i, i): the place of a row's diagonal entry belongs to the row node (bits 20..27 of its word)
                    if not srow_table(info) or ar != ac:
                        raise ValueError("diagonal-free records need one map on both sides of the Mat")
                    lines.append(f"for (int i = 0; i < {ar}; ++i) kk{k}[i*{ac} + i] = (int)((rw{k}[i] >> 20) & 0xffu);")
                if colmask and cm != rm:
                    lines += [f"bool cmk{k}[{ac}];", f"for (int j = 0; j < {ac}; ++j) cmk{k}[j] = smc{k}[lm{cm}[j]];"]
                lines += [f"for (int i = 0; i < {ar}; ++i) {{",
                          f"  const int base = (int)(rw{k}[i] & 0xfffffu) - 1;",
                          "  if (base < 0) continue;          /* row owned by another block (or BC-masked) */",
                          f"  for (int j = 0; j < {ac}; ++j) {{"]
                # a BC-masked column adds 0.0 to its (existing) position instead of branching around the atomic:
                # the value array is unchanged either way, and the scatter stays one branch per row
                val = f"t{k}[i*{ac} + j]"
                dropped = f"{'cmk%d[j]' % k if cm != rm else '(rw%d[j] >> 31)' % k}"
                if colmask:
                    val = f"(({dropped}) ? 0.0 : {val})"
                if fx:
                    # fixed-point trip: a BC-masked column scales its contributions by zero (no select per contribution); the
                    # fp64 trip of a block that fell back tracks the magnitudes as well (the next launch's scale comes from them)
                    pre_fx = [f"double fdS{k}[{ac}];", f"for (int j = 0; j < {ac}; ++j) fdS{k}[j] = ({dropped}) ? 0.0 : fd_S;"] if colmask else []
                    sj = f"fdS{k}[j]" if colmask else "fd_S"
                    at = lines.index(f"for (int i = 0; i < {ar}; ++i) {{")
                    fixed = lines[:at] + pre_fx + lines[at:] + [f"    fdw::fx_acc(&sm{k}[base + kk{k}[i*{ac} + j]], t{k}[i*{ac} + j], {sj}, fd_mu, fd_mi);", "  }", "}"]
                    unpack_fx.append("\n    ".join(fixed))
                    lines += [f"    fdw::fx_track(t{k}[i*{ac} + j], fd_mu, fd_mi);"]
                lines += [f"    atomicAdd(&sm{k}[base + kk{k}[i*{ac} + j]], {val});", "  }", "}"]
                unpack.append("\n    ".join(lines))
                if ocrp:
                    # complete rows again, but a SET of CSR rows: the flush walks the accumulator entries like the contiguous
                    # flush does and finds each entry's place in the CSR value array in a per-entry position table (4 B per
                    # nonzero, streamed).  (A row-by-row flush, 16 lanes per row with the row descriptors in LDS, issued 4x the
                    # LDS instructions and 1.5x the scalar ones for the same stores: +12 % LDS-pipe cycles in a kernel bound
                    # by that pipe, profiles/r3f_pmc_jacobian_lexicographic.txt.)
                    # The place of an entry is a global load its store depends on, so a trip of the flush loop costs a memory round
                    # trip: FU places are requested together -- and the first trip's (all of them for a block within budget)
                    # BEFORE the barrier that ends the main loop, so the flush itself waits for no load (profiles/r5a_phase_times.txt:
                    # the flush was 4.2 of a block's 17.5 microseconds)
                    preload, FU = True, 8
                    ld = (f"for (int f = 0; f < {FU}; ++f) {{ const int q = Q0 + f*nthr; g{k}[f] = q < nnzb{k} ? oc{k}_gpos[(size_t)r0_{k} + q] : -1; }}")
                    if preload:
                        flush_pre_decl.append(f"int g{k}[{FU}];")
                        flush_pre.append(ld.replace("Q0", "tid"))
                    flush.append((rm, f"for (int q0 = tid; q0 < nnzb{k}; q0 += {FU}*nthr) {{ "
                                      + (f"if (q0 >= {FU}*nthr) {{ {ld.replace('Q0', 'q0')} }} " if preload else f"int g{k}[{FU}]; {ld.replace('Q0', 'q0')} ")
                                      # ("ocrpm": -2 - place = an entry in a masked column -- zero when the rows are overwritten, untouched otherwise)
                                      + (f"if (oc{k}_flags & 1) {{ for (int f = 0; f < {FU}; ++f) {{ const int gq = g{k}[f]; "
                                         f"if (gq >= 0) arg{k}[(size_t)gq] = sm{k}[q0 + f*nthr]; else if (gq < -1) arg{k}[(size_t)(-2 - gq)] = 0.0; }} }} " if ocrpm else
                                         f"if (oc{k}_flags & 1) {{ for (int f = 0; f < {FU}; ++f) if (g{k}[f] >= 0) arg{k}[(size_t)g{k}[f]] = sm{k}[q0 + f*nthr]; }} ") +
                                      # accumulating into existing values (a second integral of the same form): the old values are
                                      # requested together as well (the places of a block are distinct)
                                      f"else {{ double o{k}[{FU}]; for (int f = 0; f < {FU}; ++f) o{k}[f] = g{k}[f] >= 0 ? arg{k}[(size_t)g{k}[f]] : 0.0; "
                                      f"for (int f = 0; f < {FU}; ++f) if (g{k}[f] >= 0) arg{k}[(size_t)g{k}[f]] = o{k}[f] + sm{k}[q0 + f*nthr]; }} }}"))
                    continue
                # complete rows, contiguous in the CSR value array: plain coalesced stores
                flush.append((rm, f"if (oc{k}_flags & 1) {{ for (int q = tid; q < nnzb{k}; q += nthr) arg{k}[(size_t)r0_{k} + q] = sm{k}[q]; }} "
                                  f"else {{ for (int q = tid; q < nnzb{k}; q += nthr) arg{k}[(size_t)r0_{k} + q] += sm{k}[q]; }}"))
                continue
            if mat_staged[k]:
                lds_items.append(("mat", k, rm, cm, bool(lg)))
                lds_tail_var.append(f"double *sm{k} = (double *)(fd_lds + fd_off); fd_off += (((size_t)mp{k}_maxnnz*8) + 15) & ~(size_t)15;")
                lds_tail_const.append(f"int *slrp{k} = (int *)(fd_lds + fd_off); fd_off += (((size_t)(p{rm}_maxnd + 1)*4) + 15) & ~(size_t)15;")
                mat_pre = [f"const int mo{k} = mp{k}_off[b], nnzb{k} = mp{k}_off[b+1] - mo{k};"]
                stage.append((rm, f"for (int q = tid; q < nnzb{k}; q += nthr) sm{k}[q] = 0;"))
                stage.append((rm, f"for (int q = tid; q <= nd{rm}; q += nthr) slrp{k}[q] = mp{k}_lrp[l0_{rm} + b + q];"))
                if lg:
                    lds_tail_const.append(f"unsigned char *smr{k} = fd_lds + fd_off; fd_off += ((size_t)p{rm}_maxnd + 15) & ~(size_t)15;")
                    lds_tail_const.append(f"unsigned char *smc{k} = fd_lds + fd_off; fd_off += ((size_t)p{cm}_maxnd + 15) & ~(size_t)15;")
                    stage.append((rm, f"for (int q = tid; q < nd{rm}; q += nthr) smr{k}[q] = rlg{k}[p{rm}_list[l0_{rm} + q]] < 0;"))
                    stage.append((cm, f"for (int q = tid; q < nd{cm}; q += nthr) smc{k}[q] = clg{k}[p{cm}_list[l0_{cm} + q]] < 0;"))
                mat_stage_pre.extend(mat_pre)
                lines = [f"for (int i = 0; i < {ar}; ++i) {{",
                         f"  const int base = slrp{k}[lm{rm}[i]];"]
                if lg:
                    lines.append(f"  if (smr{k}[lm{rm}[i]]) continue;")
                lines.append(f"  for (int j = 0; j < {ac}; ++j) {{")
                if lg:
                    lines.append(f"    if (smc{k}[lm{cm}[j]]) continue;")
                lines += [f"    atomicAdd(&sm{k}[base + kk{k}[i*{ac} + j]], t{k}[i*{ac} + j]);", "  }", "}"]
                unpack.append("\n    ".join(lines))
                # gpos < 0 marks an entry no other block touches (~gpos is its position); both kinds are added atomically
                flush.append((rm, f"for (int q = tid; q < nnzb{k}; q += nthr) {{ const double v = sm{k}[q]; const int g = mp{k}_gpos[mo{k} + q]; "
                                  f"if (v != 0.0) fdw::atomic_add<double>(&arg{k}[g < 0 ? ~g : g], v); }}"))
                continue
            store = (lambda p, v: f"fdw::atomic_add<double>(&arg{k}[{p}], {v});") if acc == INC else (lambda p, v: f"arg{k}[{p}] = {v};")
            unroll = info["arg"].unroll
            nr_, nc_ = nf * ar, nf * ac
            lines = [f"for (int fi = 0; fi < {nf}; ++fi) for (int i = 0; i < {ar}; ++i) {{",
                     f"  const int rn = {node(rm, ar, 'i', info['roff'], info['rperm'], 'fi')};",
                     f"  for (int fj = 0; fj < {nf}; ++fj) for (int j = 0; j < {ac}; ++j) {{",
                     f"    const int cn = {node(cm, ac, 'j', info['coff'], info['cperm'], 'fj')};",
                     "    if (rn < 0 || cn < 0) continue;"]
            if lg and not unroll:
                lines.append(f"    if (rlg{k}[rn] < 0 || clg{k}[cn] < 0) continue;")
            if use_table[k]:
                lines.append(f"    const int pn = tab{k}[((size_t)e*{ar} + i)*{ac} + j];")
                lines.append("    if (pn < 0) continue;")
                if rbs * cbs != 1:
                    lines.append(f"    const fd_nnz_t r0 = nrp{k}[rn]; const int rl = (int)(nrp{k}[rn+1] - r0);")
            for p in range(rbs):
                for q in range(cbs):
                    v = f"t{k}[((((size_t)(fi*{ar}+i)*{rbs} + {p})*{nc_}) + (fj*{ac}+j))*{cbs} + {q}]"
                    guard = ""
                    if lg and unroll:
                        guard = f"if (rlg{k}[rn*{rbs}+{p}] >= 0 && clg{k}[cn*{cbs}+{q}] >= 0) "
                    if use_table[k]:
                        pos = "pn" if rbs * cbs == 1 else f"(size_t)r0*{rbs * cbs} + (size_t){p}*rl*{cbs} + (size_t)(pn - r0)*{cbs} + {q}"
                        lines.append(f"    {guard}{{ {store(pos, v)} }}")
                    else:
                        lines.append(f"    {guard}{{ const fd_nnz_t ps = fdw::csr_find(rp{k}, ci{k}, rn*{rbs}+{p}, cn*{cbs}+{q}); if (ps >= 0) {{ {store('ps', v)} }} }}")
            lines += ["  }", "}"]
            unpack.append("\n      ".join(lines))
    if gk._pass_layer_arg:
        call_args.append("layer")

    includes, body = _hoist_includes(lk.code)
    src = ['#include "fd_wrapper.h"', "#include <math.h>", *includes, *[f"#include <{h}>" for h in lk.headers],
           "namespace fdk {", "#pragma clang force_cuda_host_device begin", body,
           "#pragma clang force_cuda_host_device end", "}  // namespace fdk", *decls, ""]
    sym = f"wrap_{lk.name}"
    min_waves = min_waves or configuration["min_waves"]
    lb = f"{threads}, {min_waves}" if min_waves else f"{threads}"
    src.append(f'extern "C" __global__ __launch_bounds__({lb}) void {sym}(int start, int end, {", ".join(params)})')
    src.append("{")
    need_red = bool(post)
    if need_red:
        src.append("  __shared__ double fd_red[16];")
    if fx:
        src.append("  __shared__ unsigned fd_fxmax;")
    layer_parallel = True
    if staged:
        src += ["  extern __shared__ __align__(16) unsigned char fd_lds[];",
                "  const int tid = threadIdx.x, nthr = blockDim.x;"]
        src += ["  const int b = fdw::xcd_block(blockIdx.x, gridDim.x);",
                "  const int e0 = bstart_[b], e1 = bstart_[b+1];"]
        if ptimes:
            src.append("  if (tid == 0) { fd_times[5*(size_t)b] = wall_clock64(); fd_times[5*(size_t)b + 4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }")
        src += ["  " + s for s in lds_decl + lds_tail_const + lds_tail_var]
        for mi in staged_maps:
            src.append(f"  const int l0_{mi} = p{mi}_blkoff[b], nd{mi} = p{mi}_blkoff[b+1] - l0_{mi};")
        stage_src = ["  " + s for s in mat_stage_pre]
        stage_src += ["  " + s for _, s in stage]
        # node-major staging: the node-list entry is requested first, then all the rows that depend on it
        for mi, acts in node_actions.items():
            fu = stage_unroll(mi)
            flat = lambda l: l if isinstance(l, str) else f"if ({l[1]}) {{ {l[2]} }} else {{ {l[3]} }}"
            needs_g = any("G_U" in flat(l) for act in acts for part in (0, 1) for l in act[part])
            if fu == 1:
                stage_src.append(f"  for (int i_0 = tid; i_0 < nd{mi}; i_0 += nthr) {{")
                if needs_g:
                    stage_src.append(f"    const int g_0 = p{mi}_list[l0_{mi} + i_0];")
                for part in (0, 1):
                    for act in acts:
                        for l in act[part]:
                            stage_src.append("    " + flat(l).replace("_U", "_0").replace("G_0", "g_0").replace("I_0", "i_0"))
                stage_src.append("  }")
                continue
            # A lane stages several nodes (nodes per block / lanes per block, known from the compile-time stride).  One node per trip
            # made every trip a chain of dependent memory round trips -- node id, then the rows gathered through it -- and the trips
            # followed one another: 4 round trips for the 405 nodes of a P1 block before its main loop could start.  All node ids
            # of a lane's batch are requested first, then the rows of the whole batch argument by argument (plan-ordered copies
            # do not wait for the node ids: their uniform null test is hoisted over the batch), then LDS is written: two round
            # trips whatever the batch.  Indices are clamped to the block's last node and the stores are unconditional too (a lane
            # beyond the end rewrites the last node's row with the same values): any guard makes the compiler sink the loads of
            # the later nodes behind it, one round trip each again
            sub = lambda l, f: l.replace("_U", f"_{f}").replace(f"G_{f}", f"g_{f}").replace(f"I_{f}", f"i_{f}")
            stage_src.append(f"  for (int i_b = tid; i_b < nd{mi}; i_b += {fu}*nthr) {{")
            for f in range(fu):
                stage_src.append(f"    const int i_{f} = min(i_b + {f}*nthr, nd{mi} - 1);")
            if needs_g:
                for f in range(fu):
                    stage_src.append(f"    const int g_{f} = p{mi}_list[l0_{mi} + i_{f}];")
            for act in acts:
                for l in act[0]:
                    if isinstance(l, str):
                        for f in range(fu):
                            stage_src.append("    " + sub(l, f))
                    else:
                        stage_src.append(f"    if ({l[1]}) {{ " + " ".join(sub(l[2], f) for f in range(fu)) + " } else { "
                                         + " ".join(sub(l[3], f) for f in range(fu)) + " }")
            for f in range(fu):
                for act in acts:
                    for l in act[1]:
                        stage_src.append("    " + sub(flat(l), f))
            stage_src.append("  }")
        if fx:
            stage_src.append("  if (tid == 0) fd_fxmax = 0u;")
        stage_src.append("  __syncthreads();")
        if ptimes:
            stage_src.append("  if (tid == 0) fd_times[5*(size_t)b + 1] = wall_clock64();")
        stage_src += ["  " + s for s in pre]
        src += ["  const int fd_first = e0 + tid, fd_step = nthr, fd_last = e1;"]
        # software pipeline: the packed index rows of the NEXT entity are requested before the current
        # entity's local kernel runs, so their HBM latency hides under ~10^2 fp64 instructions
        idx_loads = []      # (register row, its prefetch twin, name, length, load template: II = iteration index, EE = entity)
        raw_decode = []     # unpacking of index rows that travel as raw words (top of the trip)
        for mi in staged_maps:
            ar = maps[mi].arity * (nf if staged and not ocr else 1)       # (interior facets: the derived row holds both stacked cells)
            if configuration["stage_batch"] and ar % 2 == 0 and not ocr:
                # the packed row travels through the software pipeline as RAW words and is unpacked at the top of the trip that
                # uses it: load_lmap's shifts follow its loads directly, which made the request of the first trip's row -- issued
                # ahead of the staging phase on purpose -- a wait of one memory round trip at the head of every block
                idx_loads.append((f"unsigned lw{mi}[{ar // 2}]", f"unsigned nx_lw{mi}[{ar // 2}]", f"lw{mi}", ar // 2,
                                  f"fdw::load_rec<{ar // 2}>((const unsigned *)(p{mi}_lmap + (size_t)(II - start)*{ar}), DST);"))
                raw_decode += [f"int lm{mi}[{ar}];", f"fdw::unpack_lmap<{ar}>(lw{mi}, lm{mi});"]
            else:
                idx_loads.append((f"int lm{mi}[{ar}]", f"int nx_lm{mi}[{ar}]", f"lm{mi}", ar,
                                  f"fdw::load_lmap<{ar}>(p{mi}_lmap + (size_t)(II - start)*{ar}, DST);"))
        for info in infos:
            if info["kind"] == "mat" and mat_staged[info["k"]]:
                k, n = info["k"], info["ar"] * info["ac"]
                idx_loads.append((f"int kk{k}[{n}]", f"int nx_kk{k}[{n}]", f"kk{k}", n,
                                  f"fdw::load_packed<{ktype}, {n}>(mp{k}_k + (size_t)(II - start)*{n}, DST);"))
            if info["kind"] == "mat" and ocr:
                k, n = info["k"], info["ar"] * info["ac"]
                idx_loads.append((f"int kk{k}[{n}]", f"int nx_kk{k}[{n}]", f"kk{k}", n,
                                  f"fdw::load_packed<{ktype}, {n}>(oc{k}_k + (size_t)(II - start)*{n}, DST);"))
        for k_, ct_, n_ in direct_prefetch:
            idx_loads.append((f"{ct_} dv{k_}[{n_}]", f"{ct_} nx_dv{k_}[{n_}]", f"dv{k_}", n_,
                              f"for (int q = 0; q < {n_}; ++q) DST[q] = arg{k_}[(size_t)EE*{n_} + q];"))
        rec_decode = []
        if rec:
            # one record per instance replaces every index row above: the words are prefetched, the fields extracted at the top of
            # the trip that uses them (the prefetched copy is W registers instead of one register per index)
            (info,) = [i for i in infos if i["kind"] == "mat"]
            k, ar_, ac_ = info["k"], info["ar"], info["ac"]
            if len(rec["lbits"]) != len(staged_maps):
                raise ValueError("one local-index width per staged map")
            off = 0
            fields = []
            for mi, lb in zip(staged_maps, rec["lbits"]):
                for i in range(maps[mi].arity):
                    fields.append((f"lm{mi}[{i}]", off, lb))
                    off += lb
            for i in range(ar_):
                for j in range(ac_):
                    if rec["diag"] and i == j:
                        continue
                    fields.append((f"kk{k}[{i * ac_ + j}]", off, rec["kbits"]))
                    off += rec["kbits"]
            W = -(-off // 32)
            for mi in staged_maps:
                rec_decode.append(f"int lm{mi}[{maps[mi].arity}];")
            rec_decode.append(f"int kk{k}[{ar_ * ac_}];")
            for name, o, b in fields:
                rec_decode.append(f"{name} = fdw::rec_field<{o}, {b}>(rc{k});")
            idx_loads = [(f"unsigned rc{k}[{W}]", f"unsigned nx_rc{k}[{W}]", f"rc{k}", W,
                          f"fdw::load_rec<{W}>(oc{k}_rec + (size_t)(II - start)*{W}, DST);")]
        pf = bool(configuration["prefetch"])
        # lane order (fd_plan_set_lane_order): slot k*nthr + t of a block holds the k-th entity of lane t's contiguous
        # run, so the lanes of one trip work on entities that are far apart (no shared nodes -> no serialised LDS
        # atomics) while their index rows stay coalesced.  OCR instance lists are stored in that order already.
        # ("_d" variants keep the entity order: consecutive lanes then read and write consecutive rows of the affine maps)
        lane_threads = threads if (configuration["lane_strided"] and not direct_maps) else 0
        # virtual iteration spaces: positions in a subset / (column, layer) cells / a derived entity order; the staged and the
        # owner-computes-rows wrappers address everything through plans built on derived maps and decode the position only for
        # direct arguments and the layer argument
        virt = bool(extruded or gk._subset or ordered)
        if virt:
            if extruded and not varlay:
                lo, hi = {ALL: ("layers[0]", "layers[1]-1"), ON_BOTTOM: ("layers[0]", "layers[0]+1"),
                          ON_TOP: ("layers[1]-2", "layers[1]-1"),
                          ON_INTERIOR_FACETS: ("layers[0]", "layers[1]-1" if periodic else "layers[1]-2")}[region]
                src.append(f"  const int fd_llo = {lo}, fd_nlit = ({hi}) - fd_llo;")
                if any(i_["kind"] == "dat" and "m" in i_ and not in_lds(i_) for i_ in infos):
                    # (WRITE / RW / MIN / MAX arguments addressed from the lane: base map row + offset * layer, builder.py:94-124)
                    src.append("  const int *__restrict__ lay = layers;")
                    if periodic:
                        src.append("  const int fd_nl = lay[1] - 1 - lay[0];")
        if ocr:
            ent_of = lambda ii: f"inst_ent_[{ii}]"
        elif lane_threads:
            src += [f"  const int fd_q = (e1 - e0) / {threads}, fd_rem = (e1 - e0) - fd_q*{threads};",
                    "  const int fd_ebase = e0 + tid*fd_q + (tid < fd_rem ? tid : fd_rem);"]
            ent_of = lambda ii: f"(fd_ebase + ({ii} - e0) / {threads})"
        else:
            ent_of = lambda ii: ii
        def decode(v):
            """virtual id (position in the subset x layer) -> base entity ``e`` (+ ``layer``)"""
            out = []
            if extruded and varlay:
                out.append(f"    const int fd_col = {'fd_vcol_[%s]' % v if vtab else '0'}; const int layer = {'fd_vlay_[%s]' % v if vtab else '0'};")
                out.append("    const int e = " + ("subset_indices[fd_col];" if gk._subset else "fd_col;"))
                if any(i_["kind"] == "dat" and "m" in i_ and not in_lds(i_) for i_ in infos):
                    # WRITE / RW / MIN / MAX arguments addressed from the lane: the column's own [bottom, top) row (the layers array
                    # belongs to the superset: indexed by the entity, builder.py:754-776)
                    out.append("    const int *__restrict__ lay = layers + 2*(size_t)e;")
            elif extruded:
                out.append(f"    const int fd_col = ({v}) / fd_nlit; const int layer = fd_llo + (({v}) - fd_col*fd_nlit);")
                out.append("    const int e = " + ("subset_indices[fd_col];" if gk._subset else "fd_col;"))
            elif gk._subset:
                out.append(f"    const int e = subset_indices[{v}];")
            else:
                out.append(f"    const int e = fd_order_[{v}];")
            return out

        def prologue():
            """registers of the software pipeline and the index rows of the lane's first entity"""
            out = []
            for cur, nxt, name, n, ld in idx_loads:
                out.append(f"  {cur}; {nxt};")
            out.append("  int e_cur = 0, e_nx = 0;")
            out.append("  if (fd_first < fd_last) {")
            out.append(f"    e_cur = {ent_of('fd_first')};")
            for cur, nxt, name, n, ld in idx_loads:
                out.append("    " + ld.replace("II", "fd_first").replace("EE", "e_cur").replace("DST", name))
            out.append("  }")
            return out

        # the first trip's index rows are requested BEFORE the staging phase (they depend on the block's bounds only), so the
        # block pays one memory round trip ahead of its main loop -- staging and index rows side by side -- instead of two
        early = pf
        if early:
            src += prologue()
        src += stage_src

        def main_loop(unpack_lines, own_prologue=True):
            """the entity loop of the block (its own scope: the "_fx" wrappers hold two of them)"""
            out = ["  {"]
            if pf and own_prologue:
                out += prologue()
            out.append("  for (int it = fd_first; it < fd_last; it += fd_step) {")
            if pf:
                if virt:
                    out.extend(decode("e_cur"))
                else:
                    out.append("    const int e = e_cur;")
                out.append("    const int itn = (it + fd_step < fd_last) ? it + fd_step : it;")
                out.append(f"    e_nx = {ent_of('itn')};")
                for cur, nxt, name, n, ld in idx_loads:
                    out.append("    " + ld.replace("II", "itn").replace("EE", "e_nx").replace("DST", "nx_" + name))
            else:
                if virt:
                    out.append(f"    const int fd_v = {ent_of('it')};")
                    out.extend(decode("fd_v"))
                else:
                    out.append(f"    const int e = {ent_of('it')};")
                for cur, nxt, name, n, ld in idx_loads:
                    out.append(f"    {cur}; " + ld.replace("II", "it").replace("EE", "e").replace("DST", name))
            out += ["    " + s_ for s_ in (rec_decode or raw_decode)]
            out += ["    " + s_ for s_ in pack]
            out.append(f"    fdk::{lk.name}({', '.join(call_args)});")
            out += ["    " + s_ for s_ in unpack_lines]
            if pf:
                for cur, nxt, name, n, ld in idx_loads:
                    out.append(f"    for (int q = 0; q < {n}; ++q) {name}[q] = nx_{name}[q];")
                out.append("    e_cur = e_nx;")
            out += ["  }", "  }"]
            return out

        if fx:
            # Checked fixed-point accumulation: a fixed-point pass when the block has a scale; a block whose largest contribution
            # fell outside the window of that scale (or was not finite) clears its accumulators and redoes its instances with fp64
            # atomics -- its rows are nobody else's -- and either way leaves the record its next launch uses.
            mats = [i_ for i_ in infos if i_["kind"] == "mat"]
            if not ocr or len(mats) != 1 or len(unpack) != 1 or len(unpack_fx) != 1 or post:
                raise ValueError("fixed-point accumulation serves scalar whole-entity owner-computes-rows loops")
            K_ = mats[0]["k"]
            get_re = re.compile(r"(=|\+) sm%d\[([^\]]+)\]" % K_)
            flush_fx = [get_re.sub(lambda m_: "%s fdw::fx_get(sm%d[%s], fd_iS)" % (m_.group(1), K_, m_.group(2)), s_) for _, s_ in flush]
            for line in flush_fx:
                for piece in line.split(";"):
                    if ("sm%d[" % K_) in piece and "fdw::fx_get" not in piece:
                        raise ValueError("fixed-point accumulation: an accumulator access of this flush is not covered: " + piece.strip())
            zcount_ = f"(int)oc{K_}_maxnnz" if configuration["prefetch"] else f"nnzb{K_}"
            src += ["  " + s_ for s_ in flush_pre_decl]
            src += [f"  const fdw::fx_block_t fd_rec = fx{K_}_scale[b];",
                    "  const double fd_S = fd_rec.S, fd_iS = fd_rec.invS;",
                    "  unsigned fd_mu = 0u; int fd_mi = 0;",
                    "  bool fd_fixed = fd_S != 0.0, fd_fell = false;",
                    "  if (fd_fixed) {"]
            src += main_loop(unpack_fx, own_prologue=not early)
            src += ["    " + s_ for s_ in flush_pre]
            src += ["    fdw::fx_block_max(&fd_fxmax, fd_mu, fd_mi);",
                    "    __syncthreads();"] + (["    if (tid == 0) fd_times[5*(size_t)b + 2] = wall_clock64();"] if ptimes else []) + [
                    "    if (fdw::fx_outside(fd_rec, fd_fxmax)) {",
                    "      fd_fixed = false; fd_fell = true;",
                    f"      for (int q = tid; q < {zcount_}; q += nthr) sm{K_}[q] = 0;",
                    "      __syncthreads();",
                    "    }",
                    "  }",
                    "  if (!fd_fixed) {"]
            src += main_loop(unpack)
            src += ["    if (!fd_fell) {"] + ["      " + s_ for s_ in flush_pre] + ["    }"]
            src += ["    if (!fd_fell) fdw::fx_block_max(&fd_fxmax, fd_mu, fd_mi);",
                    "    __syncthreads();"]
            src += ["    " + s_ for _, s_ in flush]
            src.append("  } else {")
            src += ["    " + s_ for s_ in flush_fx]
            src += ["  }", f"  if (tid == 0) fdw::fx_update<{int(configuration['ocr_fx_headroom'])}>(fx{K_}_scale + b, fd_rec, fd_fxmax, fd_fell, fx{K_}_stat);"]
        else:
            src += ["  " + s_ for s_ in flush_pre_decl]
            src += main_loop(unpack, own_prologue=not early)
            src += ["  " + s_ for s_ in flush_pre]
            if flush:
                src.append("  __syncthreads();")
                if ptimes:
                    src.append("  if (tid == 0) fd_times[5*(size_t)b + 2] = wall_clock64();")
                src += ["  " + s_ for _, s_ in flush]
        if ptimes:
            src.append("  if (tid == 0) fd_times[5*(size_t)b + 3] = wall_clock64();")
        src += ["  " + s for s in post]
    else:
        if extruded:
            lo, hi = {ALL: ("lay[0]", "lay[1]-1"), ON_BOTTOM: ("lay[0]", "lay[0]+1"),
                      ON_TOP: ("lay[1]-2", "lay[1]-1"),
                      # periodic columns have one more interior facet: between the top and the bottom cell (builder.py:806-809)
                      ON_INTERIOR_FACETS: ("lay[0]", "lay[1]-1" if periodic else "lay[1]-2")}[region]
            bounds = [f"const int llo = {lo}, lhi = {hi};"]
            if periodic:
                bounds.append("const int fd_nl = lay[1] - 1 - lay[0];")
            if not varlay:
                # constant layers: one [bottom, top) row for every entity (set.py:342-345)
                src.append("  const int *__restrict__ lay = layers;")
                src += ["  " + b for b in bounds]
            # a direct (map-less) Dat written on an extruded set is addressed by the BASE entity
            # (parloop.py:494-497): all layers of a column hit the same row -> keep layers sequential.
            # Variable layers: every column has its own range -> one lane walks its column (builder.py:754-831)
            layer_parallel = not varlay and not any(i["kind"] == "dat" and "m" not in i and i["acc"] != READ for i in infos)
        src += ["  " + s for s in pre]
        if extruded and layer_parallel:
            src += ["  const long long nlay = lhi - llo;",
                    "  const long long total = (long long)(end - start) * nlay;",
                    "  for (long long it = blockIdx.x*(long long)blockDim.x + threadIdx.x; it < total; it += (long long)gridDim.x*blockDim.x) {",
                    "    const int n = start + (int)(it / nlay);",
                    "    const int layer = llo + (int)(it - (long long)(n - start)*nlay);"]
        else:
            src += ["  for (long long it = blockIdx.x*(long long)blockDim.x + threadIdx.x; it < (long long)(end - start); it += (long long)gridDim.x*blockDim.x) {",
                    "    const int n = start + (int)it;"]
        src.append("    const int e = " + ("subset_indices[n];" if gk._subset else "n;"))
        if varlay:
            # the layers array belongs to the superset: indexed by the entity, not by the position in a subset
            src.append("    const int *__restrict__ lay = layers + 2*(size_t)e;")
            src += ["    " + b for b in bounds]
        if extruded and not layer_parallel:
            src.append("    for (int layer = llo; layer < lhi; ++layer) {")
        src += ["      " + s for s in pack]
        src.append(f"      fdk::{lk.name}({', '.join(call_args)});")
        src += ["      " + s for s in unpack]
        if extruded and not layer_parallel:
            src.append("    }")
        src.append("  }")
        src += ["  " + s for s in post]
    src.append("}")
    if strides is not None:
        if len(strides) != len(staged_maps):
            raise ValueError("one compile-time stride per staged map")
        sig = next(i for i, l in enumerate(src) if l.startswith('extern "C" __global__'))
        for mi, S in zip(staged_maps, strides):
            pat = re.compile(r"\bp%d_maxnd\b" % mi)
            src[sig + 1:] = [pat.sub("((long long)%d)" % S, l) for l in src[sig + 1:]]
    return WrapperSource("\n".join(src) + "\n", sym, full_mode, layout, len(maps), staged_maps, lds_items,
                         layer_parallel, threads, kbytes, mat_staged,
                         (threads if (staged and configuration["lane_strided"] and not direct_maps) else 0),
                         ocr_lds_limit if ocr else 0)


def generate_sliced_wrapper(gk: GlobalKernel, mode: str) -> WrapperSource:
    """Row-sliced owner-computes-rows wrapper (modes "ocrs" / "ocrsp" [+ _k16] [+ _s<strides>]).

    Same contract as the owner-computes-rows wrapper -- complete CSR rows accumulated in LDS, stored without global atomics,
    MatSetValuesLocal semantics for negative indices (builder.py:573-625) -- but the unit of work is (entity, local row i)
    (fd_ocrplan_create_sliced): the local kernel is inlined once per i inside a ``switch`` on the wavefront's row index, and
    only row i of its output is read, so every instantiation keeps just the arithmetic that row needs.  The lgmaps do not
    appear: they are folded into the per-instance tables (slot = 0xffff / position = all-ones for dropped rows / columns),
    one table set per pair of lgmaps (fd_ocrplan_sliced_tables).  "ocrsp": the row blocks are ranges of row POSITIONS of a
    backend-derived row order, flushed row by row like "ocrp"."""
    lk = gk.local_kernel
    maps, map_index = _distinct_maps(gk)
    full_mode = mode
    sm_ = re.search(r"_s(\d+(?:x\d+)*)$", mode)
    strides = [int(v) for v in sm_.group(1).split("x")] if sm_ else None
    if sm_:
        mode = mode[:sm_.start()]
    # "_q<L0>x..k<K>e<S>": one bit-packed record per instance (fd_ocr_pack_records) instead of the uint16 / uint8 index rows and the
    # uint16 slot: local-map entries at L_m bits, column positions at K bits, the accumulator slot at S bits (dropped = all ones)
    rq_ = re.search(r"_q(\d+(?:x\d+)*)k(\d+)e(\d+)$", mode)
    rec = None
    if rq_:
        rec = {"lbits": [int(v) for v in rq_.group(1).split("x")], "kbits": int(rq_.group(2)), "sbits": int(rq_.group(3))}
        mode = mode[:rq_.start()]
    # "_g<pairs>": two rows per instance (fd_ocrplan_create_paired) -- the local rows in groups of two (one), two characters per group
    # (group_code); the chunk role is the group, one evaluation of the local kernel feeds both rows' accumulators
    gq_ = re.search(r"_g([0-9a-vz]+)$", mode)
    groups = None
    if gq_:
        groups = decode_groups(gq_.group(1))
        mode = mode[:gq_.start()]
        if rec is None:
            raise ValueError("paired instances come with instance records")
    ordered = mode.startswith("ocrsp")
    # "ocrspr": the flush of a derived row order through run-coded places (one byte per entry + one displacement per run of
    # CSR-consecutive rows in LDS, fd_ocr_row_runs) instead of row by row
    runflush = mode.startswith("ocrspr")
    ktype, kbytes = ("unsigned short", 2) if mode.endswith("_k16") else ("unsigned char", 1)
    skip = "0xffffu" if kbytes == 2 else "0xffu"
    threads = configuration["ocrs_block_threads"]
    params: List[str] = []
    layout: List[tuple] = []

    def P(decl, desc):
        params.append(decl)
        layout.append(desc)

    # virtual iteration spaces (subset / extruded): the instance's entity id is a position in the (subset x layer) space,
    # decoded only for direct arguments and the layer argument (every map row is a row of a derived map)
    extruded = bool(gk._extruded)
    varlay = bool(extruded and not gk._constant_layers)
    # interior facets of an extruded set: the loop runs on derived maps whose rows hold the nodes of BOTH stacked cells (Parloop.
    # _plan_map), in the order of the local kernel's packs (cell below, cell above: builder.py:94-124) -- to this wrapper simply maps
    # of twice the arity
    nf = 2 if (extruded and gk._iteration_region == ON_INTERIOR_FACETS) else 1
    if extruded:
        P("const int *__restrict__ layers", ("layers",))
    if gk._subset:
        P("const int *__restrict__ subset_indices", ("subset",))
    if varlay and (gk._pass_layer_arg or any(isinstance(a, DatKernelArg) and not a.is_indirect for a in gk.arguments)):
        P("const int *__restrict__ fd_vcol_", ("virt_col",))          # (position, layer) of every cell of the virtual space
        P("const int *__restrict__ fd_vlay_", ("virt_layer",))

    infos = []
    for k, (a, la) in enumerate(zip(gk.arguments, lk.arguments)):
        info = {"k": k, "arg": a, "acc": la.access, "ct": CTYPE[np.dtype(la.dtype)], "dtype": np.dtype(la.dtype)}
        if isinstance(a, MatKernelArg):
            info["kind"] = "mat"
            info["ar"], info["ac"] = a.maps[0].arity * nf, a.maps[1].arity * nf
            info["rbs"], info["cbs"] = int(np.prod(a.dims[0])), int(np.prod(a.dims[1]))
        elif isinstance(a, DatKernelArg):
            info["kind"] = "dat"
            info["c"] = int(np.prod(a.dim))
            if a.is_indirect:
                m = a.map_
                base = m.base_map if isinstance(m, PermutedMapKernelArg) else m
                info["m"], info["ar"] = map_index[id(base)], base.arity * nf
                info["perm"] = tuple(m.permutation) if isinstance(m, PermutedMapKernelArg) else None
                if nf > 1 and info["perm"] is not None:
                    raise ValueError("interior-facet matrix loops through PermutedMaps take the direct wrapper")
        elif isinstance(a, GlobalKernelArg):
            info["kind"] = "global"
        elif isinstance(a, PassthroughKernelArg):
            info["kind"] = "pass"
        else:
            raise TypeError(f"unsupported kernel argument {a!r}")
        infos.append(info)
    (mat,) = [i for i in infos if i["kind"] == "mat"]
    K, AR, AC, RB, CB = mat["k"], mat["ar"], mat["ac"], mat["rbs"], mat["cbs"]
    B = RB * CB            # scalars per (row node, column node) pair; node row r holds its rbs scalar rows back to back

    for info in infos:
        k, ct = info["k"], info["ct"]
        const = "const " if info["kind"] != "mat" else ""
        P(f"{const}{ct} *__restrict__ arg{k}" if info["kind"] != "pass" else f"void *arg{k}", ("arg", k))
    for mi in range(len(maps)):
        P(f"const int *__restrict__ map{mi}", ("map", mi))
    P("const int *__restrict__ bstart_", ("bstart",))
    P("const int *__restrict__ inst_ent_", ("ocr_inst_ent",))
    P("const unsigned char *__restrict__ chunk_role_", ("ocrs_chunk_role",))
    staged_maps = []
    for info in infos:
        if info["kind"] == "dat" and "m" in info and info["m"] not in staged_maps:
            staged_maps.append(info["m"])
    for mi in staged_maps:
        P(f"const int *__restrict__ p{mi}_blkoff", ("plan_blkoff", mi))
        P(f"const int *__restrict__ p{mi}_list", ("plan_list", mi))
        P(f"const unsigned short *__restrict__ p{mi}_lmap", ("plan_lmap", mi))
        P(f"long long p{mi}_maxnd", ("plan_maxnd", mi))
    P(f"const int *__restrict__ oc{K}_rblk", ("ocr_rblk", K))
    P(f"const fd_nnz_t *__restrict__ oc{K}_rowptr", ("ocr_prowptr" if ordered else "ocr_rowptr", K))
    if ordered:
        P(f"const fd_nnz_t *__restrict__ oc{K}_gstart", ("ocr_gstart", K))
    if runflush:
        if B != 1:
            raise ValueError("the run-coded flush serves scalar matrices")
        P(f"const unsigned char *__restrict__ oc{K}_grun", ("ocr_grun", K))
        P(f"const int *__restrict__ oc{K}_brun", ("ocr_brun", K))
        P(f"const fd_nnz_t *__restrict__ oc{K}_rdelta", ("ocr_rdelta", K))
    P(f"const unsigned short *__restrict__ oc{K}_slot", ("ocrs_slot", K))
    if B > 1:
        P(f"const unsigned short *__restrict__ oc{K}_rowlen", ("ocrs_rowlen", K))
    # ``unroll``: MatSetValuesLocal on dof indices with per-dof lgmaps (mat.py:700-716, parloop.py:279-314 -- component-wise
    # boundary conditions on vector spaces): which of the node's scalar rows / of the entity's scalar columns survive
    dofmask = bool(mat["arg"].unroll and mat["arg"].lgmaps)
    if dofmask:
        P(f"const unsigned char *__restrict__ oc{K}_rmask", ("ocrs_rmask", K))
        P(f"const unsigned long long *__restrict__ oc{K}_cmask", ("ocrs_cmask", K))
    P(f"const {ktype} *__restrict__ oc{K}_k", ("ocrs_kk", K))
    P(f"long long oc{K}_maxnnz", ("ocr_maxnnz", K))
    P(f"long long oc{K}_flags", ("ocr_flags", K))
    if rec:
        if B != 1 or dofmask:
            raise ValueError("instance records serve scalar matrices with node lgmaps")
        P(f"const unsigned int *__restrict__ oc{K}_rec", ("ocr_rec", K))
        skip = f"{(1 << rec['kbits']) - 1}u"
    slot_skip = f"{(1 << rec['sbits']) - 1}" if rec else "0xffff"

    ptimes = bool(configuration["phase_times"])         # profiling aid, see generate_wrapper
    if ptimes:
        P("long long *__restrict__ fd_times", ("phase_times",))
    lds_decl, lds_items, stage_nodes, pack, call_args = ["size_t fd_off = 0;"], [], {}, [], []
    for info in infos:
        k, ct = info["k"], info["ct"]
        if info["kind"] == "pass":
            call_args.append(f"arg{k}")
        elif info["kind"] == "global":
            call_args.append(f"const_cast<{ct} *>(arg{k})")
        elif info["kind"] == "mat":
            call_args.append(f"t{k}")
        elif "m" not in info:
            vi = info["arg"].index
            off = "" if vi is None else f" + {int(np.ravel_multi_index(tuple(vi), tuple(info['arg'].dim)))}"
            call_args.append(f"const_cast<{ct} *>(&arg{k}[(size_t)e*{info['c']}{off}])")
        else:
            c, ar, mi, perm = info["c"], info["ar"], info["m"], info["perm"]
            soa = c > 1            # component-major LDS layout for vector Dats (conflict-free lane strides)
            lds_items.append(("dat", mi, c, info["dtype"].itemsize, False))
            lds_decl.append(f"{ct} *s{k} = ({ct} *)(fd_lds + fd_off); fd_off += (((size_t)p{mi}_maxnd*{c}*sizeof({ct})) + 15) & ~(size_t)15;")
            # (a READ Dat unchanged since an earlier call is also kept in PLAN order and streamed, as in the whole-entity wrappers)
            P(f"const {ct} *__restrict__ pl{k}", ("plan_copy", k, mi))
            stage_nodes.setdefault(mi, []).append(
                ([f"{ct} v{k}[{c}];", f"if (pl{k}) {{ for (int j = 0; j < {c}; ++j) v{k}[j] = pl{k}[(size_t)(l0_{mi} + i)*{c} + j]; }} "
                                      f"else {{ for (int j = 0; j < {c}; ++j) v{k}[j] = arg{k}[(size_t)g*{c} + j]; }}"],
                 [f"for (int j = 0; j < {c}; ++j) s{k}[{'j*(int)p%d_maxnd + i' % mi if soa else 'i*%d + j' % c}] = v{k}[j];"]))
            idx = f"j*(int)p{mi}_maxnd + lm{mi}[{_permi(perm, 'i')}]" if soa else f"lm{mi}[{_permi(perm, 'i')}]*{c} + j"
            pack.append(f"{ct} t{k}[{ar * c}];")
            pack.append(f"for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) t{k}[i*{c}+j] = s{k}[{idx}];")
            call_args.append(f"t{k}")
    if gk._pass_layer_arg:
        call_args.append("layer")
    lds_items.append(("ocrs", K, B))
    if runflush:
        lds_decl.append(f"fd_nnz_t *srun{K} = (fd_nnz_t *)(fd_lds + fd_off); fd_off += 256*sizeof(fd_nnz_t);")
    lds_decl.append(f"double *sm{K} = (double *)(fd_lds + fd_off); fd_off += (((size_t)oc{K}_maxnnz*{B}*8) + 15) & ~(size_t)15;")

    includes, body = _hoist_includes(lk.code)
    sym = f"wrap_{lk.name}"
    src = ['#include "fd_wrapper.h"', "#include <math.h>", *includes, *[f"#include <{h}>" for h in lk.headers],
           "namespace fdk {", "#pragma clang force_cuda_host_device begin", body,
           "#pragma clang force_cuda_host_device end", "}  // namespace fdk", "",
           f'extern "C" __global__ __launch_bounds__({threads}) void {sym}(int start, int end, {", ".join(params)})', "{",
           "  extern __shared__ __align__(16) unsigned char fd_lds[];",
           "  const int tid = threadIdx.x, nthr = blockDim.x;",
           "  const int b = fdw::xcd_block(blockIdx.x, gridDim.x);",
           "  const int e0 = bstart_[b], e1 = bstart_[b+1];"]
    if ptimes:
        src.append("  if (tid == 0) { fd_times[5*(size_t)b] = wall_clock64(); fd_times[5*(size_t)b + 4] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)); }")
    src += ["  " + s for s in lds_decl]
    for mi in staged_maps:
        src.append(f"  const int l0_{mi} = p{mi}_blkoff[b], nd{mi} = p{mi}_blkoff[b+1] - l0_{mi};")
    # early: the first trip's index rows are requested ahead of the staging phase, the accumulators are zeroed by capacity and the
    # run displacements are staged after the main loop -- nothing before the main loop waits for the block's row starts or runs
    # (a second level of dependent scalar loads), and the block pays one memory round trip before its first trip instead of two
    early = bool(configuration["prefetch"])
    src += [f"  const int n0 = oc{K}_rblk[b], nown = oc{K}_rblk[b+1] - n0;",
            f"  const fd_nnz_t r0 = oc{K}_rowptr[n0]; const int nnzb = (int)(oc{K}_rowptr[n0 + nown] - r0)*{B};"]
    if runflush:
        src.append(f"  const int br0 = oc{K}_brun[b], nrun = oc{K}_brun[b+1] - br0;")
    srun_stage = (f'  _Pragma("clang loop unroll(disable) vectorize(disable)") for (int q = tid; q < nrun; q += nthr) srun{K}[q] = oc{K}_rdelta[br0 + q];'
                  if runflush else None)
    stage_src = [f"  for (int q = tid; q < {'(int)oc%d_maxnnz*%d' % (K, B) if early else 'nnzb'}; q += nthr) sm{K}[q] = 0;"]
    if runflush and not early:
        stage_src.append(srun_stage)
    for mi, acts in stage_nodes.items():
        stage_src.append(f"  for (int i = tid; i < nd{mi}; i += nthr) {{")
        stage_src.append(f"    const int g = p{mi}_list[l0_{mi} + i];")
        for part in (0, 1):
            for act in acts:
                stage_src += ["    " + l for l in act[part]]
        stage_src.append("  }")
    stage_src.append("  __syncthreads();")
    if ptimes:
        stage_src.append("  if (tid == 0) fd_times[5*(size_t)b + 1] = wall_clock64();")
    if not early:
        src += stage_src
    if extruded and not varlay:
        lo, hi = {ALL: ("layers[0]", "layers[1]-1"), ON_BOTTOM: ("layers[0]", "layers[0]+1"), ON_TOP: ("layers[1]-2", "layers[1]-1"),
                  ON_INTERIOR_FACETS: ("layers[0]", "layers[1]-1" if gk._extruded_periodic else "layers[1]-2")}[gk._iteration_region]
        src.append(f"  const int fd_llo = {lo}, fd_nlit = ({hi}) - fd_llo;")
    # every 64 consecutive slots hold one local row index, and e0 / nthr are multiples of 64: the index is wave-uniform.
    # Software pipeline like the unsliced wrappers: the index rows of the lane's NEXT instance are requested before the
    # current one's local kernel runs (a trip is only ~300 instructions, far less than an HBM round trip).
    need_e = any(i["kind"] == "dat" and "m" not in i for i in infos) or bool(gk._pass_layer_arg)
    rows = [(f"lm{mi}", maps[mi].arity * nf, f"fdw::load_lmap<{maps[mi].arity * nf}>(p{mi}_lmap + (size_t)(II - start)*{maps[mi].arity * nf}, DST);")
            for mi in staged_maps]
    rows.append(("kk", AC, f"fdw::load_packed<{ktype}, {AC}>(oc{K}_k + (size_t)(II - start)*{AC}, DST);"))
    scal = [("role", "(int)chunk_role_[(II - start) >> 6]"), ("slot", f"(int)oc{K}_slot[II - start]")]
    rec_decode = []
    if rec:
        if len(rec["lbits"]) != len(staged_maps):
            raise ValueError("one local-index width per staged map")
        off, fields = 0, []
        for mi, lb in zip(staged_maps, rec["lbits"]):
            for i in range(maps[mi].arity * nf):
                fields.append((f"lm{mi}[{i}]", off, lb))
                off += lb
        NRI = 2 if groups else 1                          # rows per instance: kk / slot of row s carry the suffix s
        for s_ in range(NRI):
            for j in range(AC):
                fields.append((f"kk{s_ if groups else ''}[{j}]", off, rec["kbits"]))
                off += rec["kbits"]
        for s_ in range(NRI):
            fields.append((f"const int slot{s_ if groups else ''}", off, rec["sbits"]))
            off += rec["sbits"]
        W = -(-off // 32)
        rec_decode = [f"int lm{mi}[{maps[mi].arity * nf}];" for mi in staged_maps] + \
            ([f"int kk0[{AC}], kk1[{AC}];"] if groups else [f"int kk[{AC}];"])
        rec_decode += [f"{name} = fdw::rec_field<{o}, {b}>(rc);" for name, o, b in fields]
        rows = [("rc", W, f"fdw::load_rec<{W}>(oc{K}_rec + (size_t)(II - start)*{W}, DST);")]
        scal = [("role", "(int)chunk_role_[(II - start) >> 6]")]
    if B > 1:
        scal.append(("rlen", f"(int)oc{K}_rowlen[II - start]"))
    if dofmask:
        scal.append(("rmask", f"(int)oc{K}_rmask[II - start]"))
    virt = extruded or bool(gk._subset)
    if need_e:
        scal.append(("fd_v" if virt else "e", "inst_ent_[II - start]"))
    # the local kernel once per local row index inside a switch on the wavefront's row index (each instantiation keeps just the
    # arithmetic its row needs)
    switch_src = ["    switch (fdw::wave_uniform(role)) {"]
    NT = AR * RB * AC * CB
    if groups:
        if B != 1 or dofmask or sorted(r for g in groups for r in g if r is not None) != list(range(AR)):
            raise ValueError("paired instances: scalar matrices, node lgmaps, the groups a partition of the local rows")
        def inst_(rows_):
            # one instantiation of the local kernel that keeps the rows ``rows_`` = [(s, local row), ...] of its output
            out_ = [f"        double t{K}[{NT}]; for (int q = 0; q < {NT}; ++q) t{K}[q] = 0;",
                    f"        fdk::{lk.name}({', '.join(call_args)});"]
            for s__, lr in rows_:
                out_.append(f"        if (slot{s__} != {slot_skip}) {{ for (int j = 0; j < {AC}; ++j) if (kk{s__}[j] != {skip}) "
                            f"atomicAdd(&sm{K}[slot{s__} + kk{s__}[j]], t{K}[{lr * AC} + j]); }}")
            return out_
        for gi, (ra, rb_) in enumerate(groups):
            if rb_ is None:
                switch_src += [f"    case {gi}: {{", "      {", *inst_([(0, ra)]), "      }", "    } break;"]
                continue
            # the instances of a group are sorted by ownership class (both rows / the first / the second only: fd_ocrplan_create_paired),
            # so nearly every wavefront needs ONE of three instantiations -- and the one-row instantiations carry no arithmetic of
            # the row nobody in the wavefront owns
            switch_src += [f"    case {gi}: {{",
                           f"      const bool fd_w0 = fdw::wave_any(slot0 != {slot_skip}), fd_w1 = fdw::wave_any(slot1 != {slot_skip});",
                           "      if (fd_w0 && fd_w1) {", *inst_([(0, ra), (1, rb_)]),
                           "      } else if (fd_w0) {", *inst_([(0, ra)]),
                           "      } else if (fd_w1) {", *inst_([(1, rb_)]),
                           "      }", "    } break;"]
    for r in ([] if groups else range(AR)):
        # element tensor t[(i*rbs + p)][(j*cbs + q)] (builder.py:573-625); CSR: scalar row (node, p) starts at
        # node_rowptr[node]*B + p*rowlen*cbs, column (k-th node of the row, q) sits at k*cbs + q
        rg = "if ((rmask >> p) & 1) " if dofmask else ""
        cg = f"if ((cmask >> (j*{CB} + q)) & 1ull) " if dofmask else ""
        if B == 1:
            scatter = [f"        for (int p = 0; p < 1; ++p) {rg}for (int j = 0; j < {AC}; ++j) if (kk[j] != {skip}) for (int q = 0; q < 1; ++q) {cg}"
                       f"atomicAdd(&sm{K}[slot + kk[j]], t{K}[{r * AC} + j]);"]
        else:
            scatter = [f"        for (int p = 0; p < {RB}; ++p) {rg}for (int j = 0; j < {AC}; ++j) if (kk[j] != {skip}) for (int q = 0; q < {CB}; ++q) {cg}",
                       f"          atomicAdd(&sm{K}[slot*{B} + (p*rlen + kk[j])*{CB} + q], t{K}[(({r * RB} + p)*{AC} + j)*{CB} + q]);"]
        switch_src += [f"    case {r}: {{",
                f"      double t{K}[{NT}]; for (int q = 0; q < {NT}; ++q) t{K}[q] = 0;",
                f"      fdk::{lk.name}({', '.join(call_args)});",
                # (dropped contributions branch around the ds_add_f64; sending them to per-lane dump words instead measured 5 % slower)
                f"      if (slot != {slot_skip}) {{", *scatter, "      }",
                "    } break;"]
    switch_src += ["    default: break;", "    }"]
    pf = bool(configuration["prefetch"])

    def loads(ii, prefix):
        out = [f"{prefix}{n} = {ex.replace('II', ii)};" for n, ex in scal]
        out += [ld.replace("II", ii).replace("DST", prefix + n) for n, _, ld in rows]
        return out
    # (prefetch distance one: requesting the index rows two trips ahead measured no faster, profiles/r4k_ab_p2_prefetch_plancopies.txt)
    for n, ln, _ in rows:
        ty = "unsigned" if rec else "int"
        src.append(f"  {ty} {n}[{ln}];" + (f" {ty} nx_{n}[{ln}];" if pf else ""))
    src.append("  int " + ", ".join(f"{n} = 0" + (f", nx_{n} = 0" if pf else "") for n, _ in scal) + ";")
    if dofmask:
        src.append("  unsigned long long cmask = 0" + (", nx_cmask = 0;" if pf else ";"))
    if pf:
        src.append("  if (e0 + tid < e1) {")
        src += ["    " + l for l in loads("(e0 + tid)", "")]
        if dofmask:
            src.append(f"    cmask = oc{K}_cmask[(e0 + tid) - start];")
        src.append("  }")
    if early:
        src += stage_src
    src.append("  for (int it = e0 + tid; it < e1; it += nthr) {")
    if pf:
        src.append("    const int itn = (it + nthr < e1) ? it + nthr : it;")
        src += ["    " + l for l in loads("itn", "nx_")]
        if dofmask:
            src.append(f"    nx_cmask = oc{K}_cmask[itn - start];")
    else:
        src += ["    " + l for l in loads("it", "")]
        if dofmask:
            src.append(f"    cmask = oc{K}_cmask[it - start];")
    if need_e and virt:
        if extruded and varlay:
            src += ["    const int fd_col = fd_vcol_[fd_v]; const int layer = fd_vlay_[fd_v];",
                    "    const int e = " + ("subset_indices[fd_col];" if gk._subset else "fd_col;")]
        elif extruded:
            src += ["    const int fd_col = fd_v / fd_nlit; const int layer = fd_llo + (fd_v - fd_col*fd_nlit);",
                    "    const int e = " + ("subset_indices[fd_col];" if gk._subset else "fd_col;")]
        else:
            src.append("    const int e = subset_indices[fd_v];")
    src += ["    " + s for s in rec_decode]
    src += ["    " + s for s in pack]
    src += switch_src
    NT = AR * RB * AC * CB
    if pf:
        for n, ln, _ in rows:
            src.append(f"    for (int q = 0; q < {ln}; ++q) {n}[q] = nx_{n}[q];")
        src.append("    " + " ".join(f"{n} = nx_{n};" for n, _ in scal) + (" cmask = nx_cmask;" if dofmask else ""))
    src.append("  }")
    # Flushes through tables (derived row orders): the table entries are global loads the stores depend on, so a trip of the flush
    # loop costs a memory round trip.  FU entries are requested together, and the first trip's (all of them for a block within
    # budget) BEFORE the barrier that ends the main loop: the flush itself then waits for no load (profiles/r5a_phase_times.txt:
    # the flush was 7.0 of a block's 17.7 microseconds on the CG2 share)
    preload = True
    post_flush = []
    if runflush:
        FU = 16
        ld = f"for (int f = 0; f < {FU}; ++f) {{ const int q = Q0 + f*nthr; g[f] = (int)oc{K}_grun[(size_t)r0 + (q < nnzb ? q : nnzb - 1)]; }}"
        if preload:
            src += [f"  int g[{FU}];", "  " + ld.replace("Q0", "tid")]
        post_flush.append(f"  for (int q0 = tid; q0 < nnzb; q0 += {FU}*nthr) {{ "
                          + (f"if (q0 >= {FU}*nthr) {{ {ld.replace('Q0', 'q0')} }} " if preload else f"int g[{FU}]; {ld.replace('Q0', 'q0')} ") +
                          f"size_t gp[{FU}]; for (int f = 0; f < {FU}; ++f) gp[f] = (size_t)(r0 + q0 + f*nthr + srun{K}[g[f]]); "
                          f"if (oc{K}_flags & 1) {{ for (int f = 0; f < {FU}; ++f) if (q0 + f*nthr < nnzb) arg{K}[gp[f]] = sm{K}[q0 + f*nthr]; }} "
                          f"else {{ double o[{FU}]; for (int f = 0; f < {FU}; ++f) o[f] = (q0 + f*nthr < nnzb) ? arg{K}[gp[f]] : 0.0; "
                          f"for (int f = 0; f < {FU}; ++f) if (q0 + f*nthr < nnzb) arg{K}[gp[f]] = o[f] + sm{K}[q0 + f*nthr]; }} }}")
    elif ordered:
        # row by row, 16 lanes per row: start / length / place of FU rows per lane group
        FU = 8
        ld = (f"for (int f = 0; f < {FU}; ++f) {{ const int fr = FR0 + f*(nthr >> 4); const int fp = n0 + (fr < nown ? fr : 0); "
              f"const fd_nnz_t a = oc{K}_rowptr[fp], b_ = oc{K}_rowptr[fp+1]; fs[f] = (int)(a - r0)*{B}; fl[f] = fr < nown ? (int)(b_ - a)*{B} : 0; "
              f"fd_[f] = (size_t)oc{K}_gstart[fp]*{B}; }}")
        decl = f"int fs[{FU}], fl[{FU}]; size_t fd_[{FU}];"
        if preload:
            src += ["  " + decl, "  " + ld.replace("FR0", "(tid >> 4)")]
        post_flush.append(f"  for (int fr0 = tid >> 4; fr0 < nown; fr0 += {FU}*(nthr >> 4)) {{ "
                          + (f"if (fr0 >= {FU}*(nthr >> 4)) {{ {ld.replace('FR0', 'fr0')} }} " if preload else f"{decl} {ld.replace('FR0', 'fr0')} ") +
                          f"if (oc{K}_flags & 1) {{ for (int f = 0; f < {FU}; ++f) for (int q = tid & 15; q < fl[f]; q += 16) arg{K}[fd_[f] + q] = sm{K}[fs[f] + q]; }} "
                          f"else {{ for (int f = 0; f < {FU}; ++f) for (int q = tid & 15; q < fl[f]; q += 16) arg{K}[fd_[f] + q] += sm{K}[fs[f] + q]; }} }}")
    if runflush and early:
        src.append(srun_stage)
    src.append("  __syncthreads();")
    if ptimes:
        src.append("  if (tid == 0) fd_times[5*(size_t)b + 2] = wall_clock64();")
    if post_flush:
        src += post_flush
    else:
        src.append(f"  if (oc{K}_flags & 1) {{ for (int q = tid; q < nnzb; q += nthr) arg{K}[(size_t)r0*{B} + q] = sm{K}[q]; }} "
                   f"else {{ for (int q = tid; q < nnzb; q += nthr) arg{K}[(size_t)r0*{B} + q] += sm{K}[q]; }}")
    if ptimes:
        src.append("  if (tid == 0) fd_times[5*(size_t)b + 3] = wall_clock64();")
    src.append("}")
    if strides is not None:
        if len(strides) != len(staged_maps):
            raise ValueError("one compile-time stride per staged map")
        sig = next(i for i, l in enumerate(src) if l.startswith('extern "C" __global__'))
        for mi, S in zip(staged_maps, strides):
            pat = re.compile(r"\bp%d_maxnd\b" % mi)
            src[sig + 1:] = [pat.sub("((long long)%d)" % S, l) for l in src[sig + 1:]]
    return WrapperSource("\n".join(src) + "\n", sym, full_mode, layout, len(maps), staged_maps, lds_items, True, threads, kbytes)


def lds_stride(max_nd: int, ocr: bool = False) -> int:
    """Node stride of a staged LDS array: the block maximum, rounded up to a multiple of configuration["lds_const_stride"] when the
    stride is compiled in (staged loops; owner-computes-rows loops keep run-time strides: compiled in they were 2 % slower,
    profiles/r1j_ab_lds_const_stride.txt); 0 = run-time stride, 1 = exact, 64 makes component offsets multiples of 512 bytes so
    that pairs of accesses fuse into ds_read2st64_b64 -- at the price of a larger LDS footprint, which costs a resident
    workgroup per CU on the P1 residual (measured: slower)."""
    g = 0 if ocr else int(configuration["lds_const_stride"])
    if g <= 0:
        return int(max_nd)
    return -(-int(max_nd) // g) * g


def record_layout(arities, max_nds, nr, nc, maxlen, same_map):
    """Field widths of the bit-packed instance records of a whole-entity owner-computes-rows loop (generate_wrapper, "_q"
    suffix): (lbits per staged map, kbits, diag, words per instance)."""
    lbits = [max(int(n - 1).bit_length(), 1) for n in max_nds]
    kbits = max(int(maxlen - 1).bit_length(), 1)
    diag = bool(same_map and nr == nc and maxlen <= 256 and configuration["ocr_records_diag"])
    bits = sum(a * b for a, b in zip(arities, lbits)) + (nr * nc - (nr if diag else 0)) * kbits
    return lbits, kbits, diag, -(-bits // 32)


def sliced_record_layout(arities, max_nds, nc, maxlen, max_nnz, rows=1):
    """Field widths of the bit-packed instance records of a row-sliced loop (generate_sliced_wrapper, "_q...e" suffix): (lbits per
    staged map, kbits, sbits, words per instance); the all-ones value of the position and slot fields means "dropped".  ``rows``
    = rows per instance (2: paired instances -- two rows of positions, two slots)."""
    lbits = [max(int(n - 1).bit_length(), 1) for n in max_nds]
    kbits = max(int(maxlen).bit_length(), 1)              # positions 0 .. maxlen-1 and the all-ones marker
    sbits = max(int(max_nnz).bit_length(), 1)
    bits = sum(a * b for a, b in zip(arities, lbits)) + rows * (nc * kbits + sbits)
    return lbits, kbits, sbits, -(-bits // 32)


_GROUP_CHARS = "0123456789abcdefghijklmnopqrstuv"


def group_code(groups) -> str:
    """Two characters per group of local rows (rows 0..31 as one base-32 digit, 'z' = no second row)."""
    return "".join(_GROUP_CHARS[a] + ("z" if b is None else _GROUP_CHARS[b]) for a, b in groups)


def decode_groups(code: str):
    if len(code) % 2:
        raise ValueError(f"bad group code {code!r}")
    return tuple((_GROUP_CHARS.index(code[i]), None if code[i + 1] == "z" else _GROUP_CHARS.index(code[i + 1])) for i in range(0, len(code), 2))


def mode_variant(base: str, kbytes: int, max_nds, rec=None, groups=None, direct_maps=()) -> str:
    """Name of the wrapper variant for a launch geometry: base mode [+ _k16] [+ _g<row groups>] [+ _q<record fields>] [+ _d<direct
    maps>] [+ _s<strides>] (``max_nds``: the maps that stay staged)."""
    m = base + ("_k16" if kbytes == 2 else "")
    if groups is not None:
        m += "_g" + group_code(groups)
    if rec is not None and base.startswith("ocrs"):
        lbits, kbits, sbits = rec[:3]
        m += "_q" + "x".join(str(b) for b in lbits) + f"k{kbits}e{sbits}"
    elif rec is not None:
        lbits, kbits, diag = rec[:3]
        m += "_q" + "x".join(str(b) for b in lbits) + f"k{kbits}" + ("d" if diag else "")
    ocr = base.startswith("ocr")
    if direct_maps:
        m += "_d" + "x".join(str(int(v)) for v in sorted(direct_maps))
    if not ocr and int(configuration["lds_const_stride"]) > 0 and max_nds:
        m += "_s" + "x".join(str(lds_stride(n, ocr)) for n in max_nds)
    return m


def _permi(perm, i):
    if perm is None:
        return i
    if len(perm) == 1:
        return "0"
    return "(" + " : ".join(f"{i} == {q} ? {p}" for q, p in enumerate(perm[:-1])) + f" : {perm[-1]})"
