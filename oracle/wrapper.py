"""oracle/wrapper.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the wrapper that PyOP2 generates and JIT-compiles for every
parloop: ``int wrap_<kernel>(start, end, [layers], [subset], dats..., maps...)``

Follows (file:line relative to /root/reference):
  * signature / arg order ........ pyop2/codegen/builder.py:962-981, pyop2/global_kernel.py:368-375
  * entity loop n in [start,end) . pyop2/codegen/builder.py:734-741
  * subset indirection ........... pyop2/codegen/builder.py:744-752
  * layer loop + iteration regions pyop2/codegen/builder.py:790-831
  * extruded node addressing ..... pyop2/codegen/builder.py:80-128  (map + offset*(layer-bottom+k))
  * Dat pack/unpack .............. pyop2/codegen/builder.py:352-429  (INC/WRITE zero-init, MIN/MAX too when the kernel requires zeroed outputs; READ/RW/MIN/MAX gather;
                                   unpack += / min / max / =)
  * Global pack/unpack ........... pyop2/codegen/builder.py:262-319
  * Mat pack/unpack .............. pyop2/codegen/builder.py:550-625  (zero-init, MatSetValues[Blocked]Local)
  * permuted map ................. pyop2/codegen/builder.py:144-176
  * periodic extrusion ........... pyop2/codegen/builder.py:101-123 (offset wraps with the ad hoc _Remainder,
                                   builder.py:26-29; offset_quotient), :806-809 (interior facets run one layer more)
  * mixed Dat / Mat packs ........ pyop2/codegen/builder.py:432-518 (MixedDatPack: parts concatenated at flat offsets),
                                   :628-699 (MixedMatPack: sub-blocks of one element tensor, one MatSetValues per block)
  * compile flags ................ pyop2/compilation.py:341-363 (gcc -O3 -march=native -ffast-math -fPIC -std=gnu11)

The generated C is compiled with the reference's own flags and driven through
ctypes exactly as pyop2/global_kernel.py:443-456 does.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
import re
import subprocess
from dataclasses import dataclass, field
from typing import Optional, Sequence, Tuple

import numpy as np

READ, WRITE, RW, INC, MIN, MAX = 1, 2, 3, 4, 5, 6      # pyop2/types/access.py:4-37
ALL, ON_BOTTOM, ON_TOP, ON_INTERIOR_FACETS = 1, 2, 3, 4

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")

# pyop2/compilation.py:345-349 (GNU flag set; gcc 11 here so no -O2 override)
REFERENCE_CFLAGS = ["-fPIC", "-Wall", "-std=gnu11", "-march=native", "-O3", "-ffast-math"]

_CTYPES = {np.dtype("float64"): "double", np.dtype("float32"): "float",
           np.dtype("int32"): "int32_t", np.dtype("uint32"): "uint32_t",
           np.dtype("int64"): "int64_t", np.dtype("uint64"): "uint64_t"}


@dataclass
class ODat:
    data: np.ndarray                      # (n, cdim) or (n,)
    access: int
    map: Optional[np.ndarray] = None      # (nent, arity) int32, or None = direct
    offset: Optional[Sequence[int]] = None  # extruded offsets per map entry
    perm: Optional[Sequence[int]] = None    # PermutedMap permutation
    offset_quotient: Optional[Sequence[int]] = None   # periodic extrusion (map.py:46-53)
    view_index: Optional[int] = None      # DatView: flat position of the one component the kernel sees (dat.py:714-805)

    @property
    def cdim(self):
        return int(np.prod(self.data.shape[1:])) if self.data.ndim > 1 else 1


@dataclass
class OGlobal:
    data: np.ndarray
    access: int


@dataclass
class OracleCSR:
    nrows: int
    ncols: int
    rbs: int
    cbs: int
    rowptr: np.ndarray
    colidx: np.ndarray
    values: np.ndarray

    def todense(self):
        A = np.zeros((self.nrows, self.ncols))
        for r in range(self.nrows):
            for q in range(self.rowptr[r], self.rowptr[r + 1]):
                A[r, self.colidx[q]] = self.values[q]
        return A

    def toscipy(self):
        import scipy.sparse as sp
        return sp.csr_matrix((self.values, self.colidx, self.rowptr), shape=(self.nrows, self.ncols))


@dataclass
class OMat:
    csr: OracleCSR
    access: int
    rmap: np.ndarray
    cmap: np.ndarray
    roffset: Optional[Sequence[int]] = None
    coffset: Optional[Sequence[int]] = None
    row_lgmap: Optional[np.ndarray] = None   # node -> node or -1 (BC rows masked)
    col_lgmap: Optional[np.ndarray] = None
    unroll: bool = False                      # MatSetValuesLocal with dof indices
    stats: dict = field(default_factory=dict)
    roffset_quotient: Optional[Sequence[int]] = None
    coffset_quotient: Optional[Sequence[int]] = None
    rperm: Optional[Sequence[int]] = None     # PermutedMap on the row / column side (builder.py:144-176: the MatPack
    cperm: Optional[Sequence[int]] = None     # indexes through the map's indexed(), which applies the permutation)


@dataclass
class OMixedDat:
    """MixedDat argument: ``parts`` are ODats (each with its own map); the local kernel sees ONE flat pack
    in which part p occupies [offset_p, offset_p + nf*arity_p*cdim_p) (builder.py:432-518)."""
    parts: Sequence[ODat]
    access: int


@dataclass
class OMixedMat:
    """MixedMat argument: ``blocks[i][j]`` are OMats; the local kernel sees one (sum rows) x (sum cols) element
    tensor whose sub-blocks are inserted block by block (builder.py:628-699)."""
    blocks: Sequence[Sequence[OMat]]
    access: int


class _CMat(ctypes.Structure):
    _fields_ = [("nrows", ctypes.c_int), ("ncols", ctypes.c_int),
                ("rbs", ctypes.c_int), ("cbs", ctypes.c_int),
                ("rowptr", ctypes.c_void_p), ("colidx", ctypes.c_void_p),
                ("vals", ctypes.c_void_p),
                ("row_lgmap", ctypes.c_void_p), ("col_lgmap", ctypes.c_void_p),
                ("dropped", ctypes.c_long), ("missing", ctypes.c_long)]


def compile_c(source: str, name: str, extra_sources=(), cflags=None, threads=False):
    """gcc JIT with on-disk cache keyed by (source, flags) -- pyop2/compilation.py:491-611."""
    os.makedirs(_BUILD, exist_ok=True)
    cflags = list(REFERENCE_CFLAGS if cflags is None else cflags)
    if threads:
        cflags.append("-fopenmp")
    key = hashlib.sha1((source + "".join(extra_sources) + " ".join(cflags)).encode()).hexdigest()[:16]
    so = os.path.join(_BUILD, f"{name}_{key}.so")
    if not os.path.exists(so):
        src = os.path.join(_BUILD, f"{name}_{key}.c")
        with open(src, "w") as f:
            f.write(source)
        cmd = ["gcc", *cflags, "-shared", "-o", so + ".tmp", src, *extra_sources, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("oracle compile failed:\n" + r.stderr + "\n" + source)
        os.replace(so + ".tmp", so)
    return ctypes.CDLL(so)


_PREAMBLE = """
#include <math.h>
#include <stdint.h>
#include <stdbool.h>
#include <string.h>
#include <complex.h>
typedef double PetscScalar;
typedef double PetscReal;
typedef int PetscInt;
typedef struct oracle_mat oracle_mat;
int oracle_MatSetValuesBlockedLocal(oracle_mat *A, int nr, const int *rows, int nc, const int *cols, const double *vals, int insert);
int oracle_MatSetValuesLocal(oracle_mat *A, int nr, const int *rows, int nc, const int *cols, const double *vals, int insert);
#include "%s"
""" % os.path.join(os.path.dirname(os.path.abspath(__file__)), "callables.h")


def generate_wrapper(kernel_src: str, kernel_name: str, args, *, subset=False,
                     extruded=False, iteration_region=ALL, pass_layer_arg=False, threads=False,
                     periodic=False, constant_layers=True, init_with_zero=False):
    """Emit the C wrapper (restating SURVEY.md Appendix A).  ``init_with_zero`` = the local kernel's
    ``requires_zeroed_output_arguments``: MIN/MAX packs (Dat and Global) then start from zero instead of the current
    values (builder.py:276-279, 368-371; passed down at builder.py:855, 871)."""
    also_zero = (MIN, MAX) if init_with_zero else ()
    # threads == "owner": the shared-memory analogue of N MPI ranks of the reference.  Thread t owns the node range
    # [trange[t], trange[t+1]) of the output's node set and executes the cells that touch it (tcells, the rank-local cell
    # set including its ghost-cell layer); contributions to nodes/rows outside the range are dropped, exactly like the
    # off-process rows a rank of the owner-computes partition never assembles.  No atomics, no private vectors.
    owner = threads == "owner"
    sig = ["int start", "int end"]
    if extruded:
        sig.append("const int *layers")
    if subset:
        sig.append("const int *subset_indices")
    body_pack, body_call, body_unpack = [], [], []
    maps = []     # (id(array), cname)

    def map_name(arr):
        for a, nm in maps:
            if a is arr:
                return nm
        nm = f"map{len(maps)}"
        maps.append((arr, nm))
        return nm

    ih = extruded and iteration_region == ON_INTERIOR_FACETS
    nf = 2 if ih else 1
    decls = []

    def declare(line):
        if line not in decls:
            decls.append(line)

    def table(name, vals):
        declare(f"static const int {name}[{len(vals)}] = {{{', '.join(str(int(o)) for o in vals)}}};")

    def node_expr(mname, arity, i, offset, perm, f="0", quotient=None):
        """map[e][perm[i]] + offset[i]*(layer - bottom + f), wrapped for periodic columns (builder.py:80-128)."""
        ii = i
        if perm is not None:                      # one table per distinct permutation of this map
            pname = f"{mname}_perm_" + "_".join(str(int(q)) for q in perm)
            table(pname, perm)
            ii = f"{pname}[{i}]"
        e = f"{mname}[(size_t)e*{arity} + {ii}]"
        if extruded and offset is not None:
            table(f"{mname}_off", offset)
            rel = f"(layer - lay[0] + {f})"
            if periodic:
                # builder.py:108-120: _Remainder(a, b) = a < b ? a : a - b (builder.py:26-29), num_layers = top - bottom
                if quotient is None:
                    rel = f"ORACLE_REM({rel}, (lay[1] - 1 - lay[0]))"
                else:
                    table(f"{mname}_quot", quotient)
                    rel = (f"(ORACLE_REM(({rel} + {mname}_quot[{ii}]), (lay[1] - 1 - lay[0])) - "
                           f"ORACLE_REM({mname}_quot[{ii}], (lay[1] - 1 - lay[0])))")
            # a permuted map permutes its offsets (and quotients) with its values (builder.py:160-169)
            e += f" + {mname}_off[{ii}]*{rel}"
        return e

    priv = []       # (arg name, ctype, length): INC Dats that get thread-private copies under OpenMP

    def emit_dat(an, a, access, tname, toff):
        """Pack/unpack of one (part of a) Dat into ``tname`` at flat offset ``toff``; returns the pack length."""
        ct = _CTYPES[a.data.dtype]
        c = a.cdim
        mn = map_name(a.map)
        ar = a.map.shape[1]
        nexpr = node_expr(mn, ar, 'i', a.offset, a.perm, 'f', a.offset_quotient)
        if a.view_index is not None:
            # builder.py:347-349, 365-367: a view packs one value per node, taken at the fixed component
            n = nf * ar
            loop = f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i)"
            tt = f"{tname}[{toff} + f*{ar}+i]"
            comp = str(int(a.view_index))
        else:
            n = nf * ar * c
            loop = f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j)"
            tt = f"{tname}[{toff} + (f*{ar}+i)*{c}+j]"
            comp = "j"
        if access in (INC, WRITE) + also_zero:
            body_pack.append(f"for (int q = 0; q < {n}; ++q) {tname}[{toff} + q] = 0;")
        else:
            body_pack.append(f"{loop} {tt} = {an}[(size_t)({nexpr})*{c} + {comp}];")
        if access != READ:
            lhs = f"{an}[(size_t)({nexpr})*{c} + {comp}]"
            if threads and not owner and access == INC:
                lhs = lhs.replace(f"{an}[", f"priv_{an}[", 1)
                priv.append((an, ct, a.data.size))
            op = {INC: f"{lhs} += {tt};",
                  MIN: f"{lhs} = {lhs} < {tt} ? {lhs} : {tt};",
                  MAX: f"{lhs} = {lhs} > {tt} ? {lhs} : {tt};",
                  WRITE: f"{lhs} = {tt};", RW: f"{lhs} = {tt};"}[access]
            if owner:
                if access != INC:
                    raise ValueError("owner-partitioned threading supports INC outputs only")
                op = f"{{ const int nd_ = {nexpr}; if (nd_ >= own_lo && nd_ < own_hi) {op} }}"
            body_unpack.append(f"{loop} {op}")
        return n

    def mat_shape(a):
        return nf * a.rmap.shape[1] * a.csr.rbs, nf * a.cmap.shape[1] * a.csr.cbs

    def emit_mat(an, a, access, tname):
        """MatSetValues[Blocked]Local of the dense pack ``tname`` (builder.py:573-625)."""
        rn, cn = map_name(a.rmap), map_name(a.cmap)
        ar, ac = a.rmap.shape[1], a.cmap.shape[1]
        rbs, cbs = a.csr.rbs, a.csr.cbs
        ins = 1 if access == WRITE else 0
        body_unpack.append(f"int r_{an}[{nf * ar}], c_{an}[{nf * ac}];")
        body_unpack.append(
            f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) r_{an}[f*{ar}+i] = "
            f"{node_expr(rn, ar, 'i', a.roffset, a.rperm, 'f', a.roffset_quotient)};")
        body_unpack.append(
            f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ac}; ++i) c_{an}[f*{ac}+i] = "
            f"{node_expr(cn, ac, 'i', a.coffset, a.cperm, 'f', a.coffset_quotient)};")
        if owner:
            body_unpack.append(f"for (int i = 0; i < {nf * ar}; ++i) if (r_{an}[i] < own_lo || r_{an}[i] >= own_hi) r_{an}[i] = -1;")
        if a.unroll:
            body_unpack.append(f"int ru_{an}[{nf * ar * rbs}], cu_{an}[{nf * ac * cbs}];")
            body_unpack.append(f"for (int i = 0; i < {nf * ar}; ++i) for (int p = 0; p < {rbs}; ++p) "
                               f"ru_{an}[i*{rbs}+p] = r_{an}[i] < 0 ? -1 : r_{an}[i]*{rbs}+p;")
            body_unpack.append(f"for (int i = 0; i < {nf * ac}; ++i) for (int p = 0; p < {cbs}; ++p) "
                               f"cu_{an}[i*{cbs}+p] = c_{an}[i] < 0 ? -1 : c_{an}[i]*{cbs}+p;")
            body_unpack.append(f"oracle_MatSetValuesLocal({an}, {nf * ar * rbs}, ru_{an}, {nf * ac * cbs}, cu_{an}, {tname}, {ins});")
        else:
            body_unpack.append(f"oracle_MatSetValuesBlockedLocal({an}, {nf * ar}, r_{an}, {nf * ac}, c_{an}, {tname}, {ins});")

    for k, a in enumerate(args):
        if isinstance(a, ODat):
            ct = _CTYPES[a.data.dtype]
            sig.append(f"{ct} *arg{k}")
            c = a.cdim
            if a.map is None:
                # direct: pointer straight into the Dat (builder.py:387-396)
                body_call.append(f"&arg{k}[(size_t)e*{c}" + (f" + {a.view_index}" if a.view_index is not None else "") + "]")
                continue
            body_pack.append(f"{ct} t{k}[{nf * a.map.shape[1] * c}];")
            emit_dat(f"arg{k}", a, a.access, f"t{k}", 0)
            body_call.append(f"t{k}")
        elif isinstance(a, OMixedDat):
            # one flat pack, the parts at consecutive offsets (builder.py:439-475); one pointer per part
            ct = _CTYPES[a.parts[0].data.dtype]
            total = sum(nf * p.map.shape[1] * p.cdim for p in a.parts)
            body_pack.append(f"{ct} t{k}[{total}];")
            off = 0
            for pi, p in enumerate(a.parts):
                sig.append(f"{ct} *arg{k}_{pi}")
                off += emit_dat(f"arg{k}_{pi}", p, a.access, f"t{k}", off)
            body_call.append(f"t{k}")
        elif isinstance(a, OGlobal):
            ct = _CTYPES[a.data.dtype]
            sig.append(f"{ct} *arg{k}")
            n = a.data.size
            if a.access == READ:
                body_call.append(f"arg{k}")
            else:
                # private accumulator then combine (builder.py:292-319)
                init = {INC: "0", MIN: f"arg{k}[q]", MAX: f"arg{k}[q]", WRITE: "0", RW: f"arg{k}[q]"}[a.access]
                if a.access in also_zero:
                    init = "0"
                body_pack.append(f"{ct} t{k}[{n}]; for (int q = 0; q < {n}; ++q) t{k}[q] = {init};")
                body_call.append(f"t{k}")
                op = {INC: f"arg{k}[q] += t{k}[q];",
                      MIN: f"arg{k}[q] = arg{k}[q] < t{k}[q] ? arg{k}[q] : t{k}[q];",
                      MAX: f"arg{k}[q] = arg{k}[q] > t{k}[q] ? arg{k}[q] : t{k}[q];",
                      WRITE: f"arg{k}[q] = t{k}[q];", RW: f"arg{k}[q] = t{k}[q];"}[a.access]
                body_unpack.append(f"for (int q = 0; q < {n}; ++q) {op}")
        elif isinstance(a, OMat):
            sig.append(f"oracle_mat *arg{k}")
            nr, nc = mat_shape(a)
            body_pack.append(f"double t{k}[{nr * nc}]; for (int q = 0; q < {nr * nc}; ++q) t{k}[q] = 0;")
            body_call.append(f"t{k}")
            emit_mat(f"arg{k}", a, a.access, f"t{k}")
        elif isinstance(a, OMixedMat):
            # one (R x C) element tensor; block (i, j) is copied out of it and inserted on its own (builder.py:667-699)
            R = sum(mat_shape(row[0])[0] for row in a.blocks)
            C = sum(mat_shape(b)[1] for b in a.blocks[0])
            body_pack.append(f"double t{k}[{R * C}]; for (int q = 0; q < {R * C}; ++q) t{k}[q] = 0;")
            body_call.append(f"t{k}")
            ro = 0
            for bi, row in enumerate(a.blocks):
                co = 0
                for bj, blk in enumerate(row):
                    an = f"arg{k}_{bi}_{bj}"
                    sig.append(f"oracle_mat *{an}")
                    nr, nc = mat_shape(blk)
                    body_unpack.append(f"double t_{an}[{nr * nc}];")
                    body_unpack.append(f"for (int i = 0; i < {nr}; ++i) for (int j = 0; j < {nc}; ++j) "
                                       f"t_{an}[i*{nc}+j] = t{k}[({ro}+i)*{C} + {co}+j];")
                    emit_mat(an, blk, a.access, f"t_{an}")
                    co += nc
                ro += mat_shape(row[0])[0]
        else:
            raise TypeError(a)
    for _, nm in maps:
        sig.append(f"const int *{nm}")
    if owner:
        sig += ["int nthreads_", "const int *trange", "const int *tcell_off", "const int *tcells"]
    if pass_layer_arg:
        body_call.append("layer")

    # PETSc's header is not available to the oracle; the three typedefs the wrappers use are in _PREAMBLE
    kernel_src = re.sub(r'^[ \t]*#[ \t]*include[ \t]*[<"]petsc[a-z]*\.h[>"][ \t]*$', "", kernel_src, flags=re.M)
    lines = [_PREAMBLE, "#include <stdlib.h>", "#define ORACLE_REM(a, b) ((a) < (b) ? (a) : (a) - (b))",
             kernel_src, *decls, f"int wrap_{kernel_name}({', '.join(sig)})", "{"]
    if owner:
        lines += ['  _Pragma("omp parallel for schedule(static, 1) num_threads(nthreads_)")',
                  "  for (int th_ = 0; th_ < nthreads_; ++th_) {",
                  "  const int own_lo = trange[th_], own_hi = trange[th_ + 1];",
                  "  for (int n = tcell_off[th_]; n < tcell_off[th_ + 1]; ++n) {",
                  "    int e = tcells[n];"]
    elif threads:
        # shared-memory analogue of rank-local assembly + local_to_global SUM (pyop2/types/dat.py:659-678):
        # contiguous entity ranges per thread, private output vectors summed afterwards
        lines.append('  _Pragma("omp parallel")')
        lines.append("  {")
        for an, ct, n in priv:
            lines.append(f"  {ct} *priv_{an} = ({ct} *)calloc({n}, sizeof({ct}));")
        lines.append('  _Pragma("omp for schedule(static)")')
    if not owner:
        lines += ["  for (int n = start; n < end; ++n) {",
                  "    int e = " + ("subset_indices[n];" if subset else "n;")]
    if extruded:
        # builder.py:790-812; periodic columns have one more interior facet (between the top and the bottom cell)
        lo, hi = {ALL: ("lay[0]", "lay[1]-1"),
                  ON_BOTTOM: ("lay[0]", "lay[0]+1"),
                  ON_TOP: ("lay[1]-2", "lay[1]-1"),
                  ON_INTERIOR_FACETS: ("lay[0]", "lay[1]-1" if periodic else "lay[1]-2")}[iteration_region]
        # builder.py:754-760, 834-838: one [bottom, top) row for all entities (constant layers) or one per entity,
        # indexed by the entity itself (for a Subset: the superset's entity, set.py:434-436)
        lines.append("    const int *lay = layers" + ("" if constant_layers else " + 2*(size_t)e") + ";")
        lines.append(f"    for (int layer = {lo}; layer < {hi}; ++layer) {{")
    lines += ["      " + s for s in body_pack]
    lines.append(f"      {kernel_name}({', '.join(body_call)});")
    lines += ["      " + s for s in body_unpack]
    if extruded:
        lines.append("    }")
    lines.append("  }")
    if owner:
        lines.append("  }")
    elif threads:
        for an, ct, n in priv:
            lines.append('  _Pragma("omp critical")')
            lines.append(f"  for (long q = 0; q < {n}; ++q) {an}[q] += priv_{an}[q];")
            lines.append(f"  free(priv_{an});")
        lines.append("  }")
    lines += ["  return 0;", "}"]
    return "\n".join(lines), [m for m, _ in maps]


def par_loop(kernel_src: str, kernel_name: str, start: int, end: int, args, *,
             subset: Optional[np.ndarray] = None, layers: Optional[Tuple[int, int]] = None,
             iteration_region=ALL, pass_layer_arg=False, cflags=None, return_fn=False, threads=False,
             periodic=False, init_with_zero=False, owner_partition=None):
    """Generate + compile + run the wrapper over [start, end).  Arrays are modified in place.
    ``layers``: (bottom, top) for constant layers or an (nentities, 2) array for variable layers.
    ``threads="owner"`` with ``owner_partition=(trange, tcell_off, tcells)`` (see make_owner_partition()): node-partitioned
    OpenMP execution, one thread per node range."""
    constant_layers = layers is None or np.ndim(layers) == 1
    code, maps = generate_wrapper(kernel_src, kernel_name, args, subset=subset is not None,
                                  extruded=layers is not None, iteration_region=iteration_region,
                                  pass_layer_arg=pass_layer_arg, threads=threads, periodic=periodic,
                                  constant_layers=constant_layers, init_with_zero=init_with_zero)
    lib = compile_c(code, "wrap_" + kernel_name + ("_own" if threads == "owner" else "_omp" if threads else ""), extra_sources=[os.path.join(_HERE, "csr.c")],
                    cflags=cflags, threads=threads)
    fn = getattr(lib, "wrap_" + kernel_name)
    cargs = [ctypes.c_int(start), ctypes.c_int(end)]
    keep = []
    if layers is not None:
        la = np.ascontiguousarray(layers, dtype=np.int32)
        keep.append(la)
        cargs.append(la.ctypes.data_as(ctypes.c_void_p))
    if subset is not None:
        sa = np.ascontiguousarray(subset, dtype=np.int32)
        keep.append(sa)
        cargs.append(sa.ctypes.data_as(ctypes.c_void_p))
    cmats = []

    def add_mat(a):
        cm = _CMat(a.csr.nrows, a.csr.ncols, a.csr.rbs, a.csr.cbs,
                   a.csr.rowptr.ctypes.data, a.csr.colidx.ctypes.data, a.csr.values.ctypes.data,
                   a.row_lgmap.ctypes.data if a.row_lgmap is not None else None,
                   a.col_lgmap.ctypes.data if a.col_lgmap is not None else None, 0, 0)
        cmats.append((a, cm))
        cargs.append(ctypes.byref(cm))

    for a in args:
        if isinstance(a, (ODat, OGlobal)):
            assert a.data.flags.c_contiguous
            cargs.append(a.data.ctypes.data_as(ctypes.c_void_p))
        elif isinstance(a, OMixedDat):
            for p in a.parts:
                assert p.data.flags.c_contiguous
                cargs.append(p.data.ctypes.data_as(ctypes.c_void_p))
        elif isinstance(a, OMixedMat):
            for row in a.blocks:
                for blk in row:
                    add_mat(blk)
        else:
            add_mat(a)
    for m in maps:
        assert m.dtype == np.int32 and m.flags.c_contiguous, "maps must be contiguous int32"
        cargs.append(m.ctypes.data_as(ctypes.c_void_p))
    if threads == "owner":
        trange, tcell_off, tcells = (np.ascontiguousarray(x, dtype=np.int32) for x in owner_partition)
        keep += [trange, tcell_off, tcells]
        cargs.append(ctypes.c_int(len(trange) - 1))
        cargs += [x.ctypes.data_as(ctypes.c_void_p) for x in (trange, tcell_off, tcells)]
    fn.restype = ctypes.c_int
    if return_fn:
        return fn, cargs, keep, cmats
    fn(*cargs)
    for a, cm in cmats:
        a.stats = {"dropped": cm.dropped, "missing": cm.missing}
        if cm.missing:
            raise RuntimeError(f"{cm.missing} matrix entries outside the sparsity")
    return None


def make_owner_partition(mapv: np.ndarray, ncell: int, nnode: int, nthreads: int):
    """(trange, tcell_off, tcells) for ``threads="owner"``: thread t owns nodes [trange[t], trange[t+1]) and runs the
    cells of [0, ncell) that touch one of them -- the analogue of one MPI rank's owned + ghost cells
    (firedrake/mesh.py:1131-1179 with the vertex-adjacent overlap of SURVEY.md 8e option 1)."""
    trange = (np.arange(nthreads + 1, dtype=np.int64) * nnode // nthreads).astype(np.int32)
    owner_of = np.searchsorted(trange, mapv[:ncell], side="right") - 1            # (ncell, arity)
    lo, hi = owner_of.min(axis=1), owner_of.max(axis=1)
    lists = []
    if (hi - lo).max() <= 1:
        # cells straddle at most two neighbouring ranges (contiguous partitions of a locality-preserving numbering)
        ids = np.arange(ncell, dtype=np.int32)
        both = hi > lo
        owner_t = np.concatenate([lo, hi[both]])
        cell = np.concatenate([ids, ids[both]])
    else:
        pairs = np.unique(np.stack([owner_of.reshape(-1), np.repeat(np.arange(ncell), mapv.shape[1])], axis=1), axis=0)
        owner_t, cell = pairs[:, 0], pairs[:, 1].astype(np.int32)
    order = np.argsort(owner_t, kind="stable")
    tcells = cell[order].astype(np.int32)
    tcell_off = np.concatenate([[0], np.cumsum(np.bincount(owner_t, minlength=nthreads))]).astype(np.int32)
    del lists
    return trange, tcell_off, tcells


_csr_lib = None


def _csrlib():
    global _csr_lib
    if _csr_lib is None:
        with open(os.path.join(_HERE, "csr.c")) as f:
            src = f.read()
        _csr_lib = compile_c(src, "oracle_csr")
    return _csr_lib


def build_sparsity(nrow_nodes: int, ncol_nodes: int, pairs, rbs=1, cbs=1, set_diag=True) -> OracleCSR:
    """pairs: list of (rmap, cmap) or (rmap, cmap, nlayers, roffset, coffset[, rquotient, cquotient, periodic[, region]])
    -- ``region`` in the oracle's numbering (ALL = 1 ...); one entry per (map pair, iteration region).  ``nlayers`` is
    the number of cell layers (constant layers) or the (nentities, 2) array of [bottom, top) node levels."""
    lib = _csrlib()
    n = len(pairs)
    P = ctypes.POINTER(ctypes.c_int)
    rm = (P * n)(); cmm = (P * n)(); ro = (P * n)(); co = (P * n)(); rq = (P * n)(); cq = (P * n)(); lay = (P * n)()
    nent = (ctypes.c_int * n)(); ra = (ctypes.c_int * n)(); ca = (ctypes.c_int * n)(); nl = (ctypes.c_int * n)()
    reg = (ctypes.c_int * n)(); per = (ctypes.c_int * n)()
    keep = []

    def arr(x):
        x = np.ascontiguousarray(x, dtype=np.int32)
        keep.append(x)
        return x.ctypes.data_as(P)

    for k, p in enumerate(pairs):
        r, c = np.ascontiguousarray(p[0], dtype=np.int32), np.ascontiguousarray(p[1], dtype=np.int32)
        keep += [r, c]
        rm[k] = r.ctypes.data_as(P); cmm[k] = c.ctypes.data_as(P)
        nent[k] = r.shape[0]; ra[k] = r.shape[1]; ca[k] = c.shape[1]
        reg[k], per[k] = ALL, 0
        if len(p) > 2 and p[2] is not None and np.size(p[2]) and np.any(p[2]):
            if np.ndim(p[2]) == 2:                      # variable layers
                la = np.ascontiguousarray(p[2], dtype=np.int32)
                lay[k] = arr(la)
                nl[k] = int((la[:, 1] - 1 - la[:, 0]).max())
            else:
                nl[k] = int(p[2])
            ro[k] = arr(p[3]); co[k] = arr(p[4])
            if len(p) > 5 and p[5] is not None:
                rq[k] = arr(p[5])
            if len(p) > 6 and p[6] is not None:
                cq[k] = arr(p[6])
            if len(p) > 7:
                per[k] = int(bool(p[7]))
            if len(p) > 8 and p[8] is not None:
                reg[k] = int(p[8])
        else:
            nl[k] = 0
    rp = P(); ci = P()
    lib.oracle_build_node_sparsity_ex.restype = ctypes.c_long
    nnz = lib.oracle_build_node_sparsity_ex(nrow_nodes, ncol_nodes, int(bool(set_diag)),
                                            n, rm, cmm, nent, ra, ca, nl, ro, co, reg, per, rq, cq, lay,
                                            ctypes.byref(rp), ctypes.byref(ci))
    nrowptr = np.ctypeslib.as_array(rp, shape=(nrow_nodes + 1,)).copy()
    ncolidx = np.ctypeslib.as_array(ci, shape=(max(nnz, 1),))[:nnz].copy()
    lib.oracle_free(rp); lib.oracle_free(ci)
    if rbs == 1 and cbs == 1:
        rowptr, colidx = nrowptr, ncolidx
    else:
        rowptr = np.empty(nrow_nodes * rbs + 1, dtype=np.int32)
        colidx = np.empty(nnz * rbs * cbs, dtype=np.int32)
        lib.oracle_expand_blocks(nrow_nodes, nrowptr.ctypes.data_as(P), ncolidx.ctypes.data_as(P),
                                 rbs, cbs, rowptr.ctypes.data_as(P), colidx.ctypes.data_as(P))
    return OracleCSR(nrow_nodes * rbs, ncol_nodes * cbs, rbs, cbs, rowptr, colidx,
                     np.zeros(len(colidx), dtype=np.float64))
