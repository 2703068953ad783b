"""oracle/wrapper.py -- TEST INFRASTRUCTURE ONLY.

CPU restatement of the wrapper that PyOP2 generates and JIT-compiles for every
parloop: ``int wrap_<kernel>(start, end, [layers], [subset], dats..., maps...)``

Follows (file:line relative to /root/reference):
  * signature / arg order ........ pyop2/codegen/builder.py:962-981, pyop2/global_kernel.py:368-375
  * entity loop n in [start,end) . pyop2/codegen/builder.py:734-741
  * subset indirection ........... pyop2/codegen/builder.py:744-752
  * layer loop + iteration regions pyop2/codegen/builder.py:790-831
  * extruded node addressing ..... pyop2/codegen/builder.py:80-128  (map + offset*(layer-bottom+k))
  * Dat pack/unpack .............. pyop2/codegen/builder.py:352-429  (INC/WRITE zero-init; READ/RW/MIN/MAX gather;
                                   unpack += / min / max / =)
  * Global pack/unpack ........... pyop2/codegen/builder.py:262-319
  * Mat pack/unpack .............. pyop2/codegen/builder.py:550-625  (zero-init, MatSetValues[Blocked]Local)
  * permuted map ................. pyop2/codegen/builder.py:144-176
  * compile flags ................ pyop2/compilation.py:341-363 (gcc -O3 -march=native -ffast-math -fPIC -std=gnu11)

The generated C is compiled with the reference's own flags and driven through
ctypes exactly as pyop2/global_kernel.py:443-456 does.
"""
from __future__ import annotations

import ctypes
import hashlib
import os
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
"""


def generate_wrapper(kernel_src: str, kernel_name: str, args, *, subset=False,
                     extruded=False, iteration_region=ALL, pass_layer_arg=False, threads=False):
    """Emit the C wrapper (restating SURVEY.md Appendix A)."""
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

    def node_expr(mname, arity, i, offset, perm, f="0"):
        ii = f"{mname}_perm[{i}]" if perm is not None else i
        e = f"{mname}[(size_t)e*{arity} + {ii}]"
        if extruded and offset is not None:
            e += f" + {mname}_off[{i}]*(layer - layers[0] + {f})"
        return e

    decls = []
    priv = []       # (arg index, ctype, length): INC Dats that get thread-private copies under OpenMP
    for k, a in enumerate(args):
        if isinstance(a, ODat):
            ct = _CTYPES[a.data.dtype]
            sig.append(f"{ct} *arg{k}")
            c = a.cdim
            if a.map is None:
                # direct: pointer straight into the Dat (builder.py:387-396)
                body_call.append(f"&arg{k}[(size_t)e*{c}]")
                continue
            mn = map_name(a.map)
            ar = a.map.shape[1]
            if a.offset is not None:
                decls.append(f"static const int {mn}_off[{ar}] = {{{', '.join(str(int(o)) for o in a.offset)}}};")
            if a.perm is not None:
                decls.append(f"static const int {mn}_perm[{ar}] = {{{', '.join(str(int(o)) for o in a.perm)}}};")
            body_pack.append(f"{ct} t{k}[{nf * ar * c}];")
            if a.access in (INC, WRITE):
                body_pack.append(f"for (int q = 0; q < {nf * ar * c}; ++q) t{k}[q] = 0;")
            else:
                body_pack.append(
                    f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) "
                    f"t{k}[(f*{ar}+i)*{c}+j] = arg{k}[(size_t)({node_expr(mn, ar, 'i', a.offset, a.perm, 'f')})*{c} + j];")
            body_call.append(f"t{k}")
            if a.access != READ:
                lhs = f"arg{k}[(size_t)({node_expr(mn, ar, 'i', a.offset, a.perm, 'f')})*{c} + j]"
                rhs = f"t{k}[(f*{ar}+i)*{c}+j]"
                if threads and a.access == INC:
                    lhs = lhs.replace(f"arg{k}[", f"priv{k}[", 1)
                    priv.append((k, ct, a.data.size))
                op = {INC: f"{lhs} += {rhs};",
                      MIN: f"{lhs} = {lhs} < {rhs} ? {lhs} : {rhs};",
                      MAX: f"{lhs} = {lhs} > {rhs} ? {lhs} : {rhs};",
                      WRITE: f"{lhs} = {rhs};", RW: f"{lhs} = {rhs};"}[a.access]
                body_unpack.append(
                    f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) for (int j = 0; j < {c}; ++j) {op}")
        elif isinstance(a, OGlobal):
            ct = _CTYPES[a.data.dtype]
            sig.append(f"{ct} *arg{k}")
            n = a.data.size
            if a.access == READ:
                body_call.append(f"arg{k}")
            else:
                # private accumulator then combine (builder.py:292-319)
                init = {INC: "0", MIN: f"arg{k}[q]", MAX: f"arg{k}[q]", WRITE: "0", RW: f"arg{k}[q]"}[a.access]
                body_pack.append(f"{ct} t{k}[{n}]; for (int q = 0; q < {n}; ++q) t{k}[q] = {init};")
                body_call.append(f"t{k}")
                op = {INC: f"arg{k}[q] += t{k}[q];",
                      MIN: f"arg{k}[q] = arg{k}[q] < t{k}[q] ? arg{k}[q] : t{k}[q];",
                      MAX: f"arg{k}[q] = arg{k}[q] > t{k}[q] ? arg{k}[q] : t{k}[q];",
                      WRITE: f"arg{k}[q] = t{k}[q];", RW: f"arg{k}[q] = t{k}[q];"}[a.access]
                body_unpack.append(f"for (int q = 0; q < {n}; ++q) {op}")
        elif isinstance(a, OMat):
            sig.append(f"oracle_mat *arg{k}")
            rn, cn = map_name(a.rmap), map_name(a.cmap)
            ar, ac = a.rmap.shape[1], a.cmap.shape[1]
            rbs, cbs = a.csr.rbs, a.csr.cbs
            if a.roffset is not None:
                decls.append(f"static const int {rn}_off[{ar}] = {{{', '.join(str(int(o)) for o in a.roffset)}}};")
            if a.coffset is not None and cn != rn:
                decls.append(f"static const int {cn}_off[{ac}] = {{{', '.join(str(int(o)) for o in a.coffset)}}};")
            size = nf * ar * rbs * nf * ac * cbs
            body_pack.append(f"double t{k}[{size}]; for (int q = 0; q < {size}; ++q) t{k}[q] = 0;")
            body_call.append(f"t{k}")
            ins = 1 if a.access == WRITE else 0
            body_unpack.append(f"int r{k}[{nf * ar}], c{k}[{nf * ac}];")
            body_unpack.append(
                f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ar}; ++i) r{k}[f*{ar}+i] = {node_expr(rn, ar, 'i', a.roffset, None, 'f')};")
            body_unpack.append(
                f"for (int f = 0; f < {nf}; ++f) for (int i = 0; i < {ac}; ++i) c{k}[f*{ac}+i] = {node_expr(cn, ac, 'i', a.coffset, None, 'f')};")
            if a.unroll:
                body_unpack.append(f"int ru{k}[{nf * ar * rbs}], cu{k}[{nf * ac * cbs}];")
                body_unpack.append(f"for (int i = 0; i < {nf * ar}; ++i) for (int p = 0; p < {rbs}; ++p) "
                                   f"ru{k}[i*{rbs}+p] = r{k}[i] < 0 ? -1 : r{k}[i]*{rbs}+p;")
                body_unpack.append(f"for (int i = 0; i < {nf * ac}; ++i) for (int p = 0; p < {cbs}; ++p) "
                                   f"cu{k}[i*{cbs}+p] = c{k}[i] < 0 ? -1 : c{k}[i]*{cbs}+p;")
                body_unpack.append(f"oracle_MatSetValuesLocal(arg{k}, {nf * ar * rbs}, ru{k}, {nf * ac * cbs}, cu{k}, t{k}, {ins});")
            else:
                body_unpack.append(f"oracle_MatSetValuesBlockedLocal(arg{k}, {nf * ar}, r{k}, {nf * ac}, c{k}, t{k}, {ins});")
        else:
            raise TypeError(a)
    for _, nm in maps:
        sig.append(f"const int *{nm}")
    if pass_layer_arg:
        body_call.append("layer")

    lines = [_PREAMBLE, "#include <stdlib.h>", kernel_src, *decls, f"int wrap_{kernel_name}({', '.join(sig)})", "{"]
    if threads:
        # shared-memory analogue of rank-local assembly + local_to_global SUM (pyop2/types/dat.py:659-678):
        # contiguous entity ranges per thread, private output vectors summed afterwards
        lines.append('  _Pragma("omp parallel")')
        lines.append("  {")
        for k, ct, n in priv:
            lines.append(f"  {ct} *priv{k} = ({ct} *)calloc({n}, sizeof({ct}));")
        lines.append('  _Pragma("omp for schedule(static)")')
    lines += ["  for (int n = start; n < end; ++n) {",
              "    int e = " + ("subset_indices[n];" if subset else "n;")]
    if extruded:
        lo, hi = {ALL: ("layers[0]", "layers[1]-1"),
                  ON_BOTTOM: ("layers[0]", "layers[0]+1"),
                  ON_TOP: ("layers[1]-2", "layers[1]-1"),
                  ON_INTERIOR_FACETS: ("layers[0]", "layers[1]-2")}[iteration_region]
        lines.append(f"    for (int layer = {lo}; layer < {hi}; ++layer) {{")
    lines += ["      " + s for s in body_pack]
    lines.append(f"      {kernel_name}({', '.join(body_call)});")
    lines += ["      " + s for s in body_unpack]
    if extruded:
        lines.append("    }")
    lines.append("  }")
    if threads:
        for k, ct, n in priv:
            lines.append('  _Pragma("omp critical")')
            lines.append(f"  for (long q = 0; q < {n}; ++q) arg{k}[q] += priv{k}[q];")
            lines.append(f"  free(priv{k});")
        lines.append("  }")
    lines += ["  return 0;", "}"]
    return "\n".join(lines), [m for m, _ in maps]


def par_loop(kernel_src: str, kernel_name: str, start: int, end: int, args, *,
             subset: Optional[np.ndarray] = None, layers: Optional[Tuple[int, int]] = None,
             iteration_region=ALL, pass_layer_arg=False, cflags=None, return_fn=False, threads=False):
    """Generate + compile + run the wrapper over [start, end).  Arrays are modified in place."""
    code, maps = generate_wrapper(kernel_src, kernel_name, args, subset=subset is not None,
                                  extruded=layers is not None, iteration_region=iteration_region,
                                  pass_layer_arg=pass_layer_arg, threads=threads)
    lib = compile_c(code, "wrap_" + kernel_name + ("_omp" if threads else ""), extra_sources=[os.path.join(_HERE, "csr.c")],
                    cflags=cflags, threads=threads)
    fn = getattr(lib, "wrap_" + kernel_name)
    cargs = [ctypes.c_int(start), ctypes.c_int(end)]
    keep = []
    if layers is not None:
        la = np.asarray(layers, dtype=np.int32)
        keep.append(la)
        cargs.append(la.ctypes.data_as(ctypes.c_void_p))
    if subset is not None:
        sa = np.ascontiguousarray(subset, dtype=np.int32)
        keep.append(sa)
        cargs.append(sa.ctypes.data_as(ctypes.c_void_p))
    cmats = []
    for a in args:
        if isinstance(a, (ODat, OGlobal)):
            assert a.data.flags.c_contiguous
            cargs.append(a.data.ctypes.data_as(ctypes.c_void_p))
        else:
            cm = _CMat(a.csr.nrows, a.csr.ncols, a.csr.rbs, a.csr.cbs,
                       a.csr.rowptr.ctypes.data, a.csr.colidx.ctypes.data, a.csr.values.ctypes.data,
                       a.row_lgmap.ctypes.data if a.row_lgmap is not None else None,
                       a.col_lgmap.ctypes.data if a.col_lgmap is not None else None, 0, 0)
            cmats.append((a, cm))
            cargs.append(ctypes.byref(cm))
    for m in maps:
        assert m.dtype == np.int32 and m.flags.c_contiguous, "maps must be contiguous int32"
        cargs.append(m.ctypes.data_as(ctypes.c_void_p))
    fn.restype = ctypes.c_int
    if return_fn:
        return fn, cargs, keep, cmats
    fn(*cargs)
    for a, cm in cmats:
        a.stats = {"dropped": cm.dropped, "missing": cm.missing}
        if cm.missing:
            raise RuntimeError(f"{cm.missing} matrix entries outside the sparsity")
    return None


_csr_lib = None


def _csrlib():
    global _csr_lib
    if _csr_lib is None:
        with open(os.path.join(_HERE, "csr.c")) as f:
            src = f.read()
        _csr_lib = compile_c(src, "oracle_csr")
    return _csr_lib


def build_sparsity(nrow_nodes: int, ncol_nodes: int, pairs, rbs=1, cbs=1, set_diag=True) -> OracleCSR:
    """pairs: list of (rmap, cmap) or (rmap, cmap, nlayers, roffset, coffset)."""
    lib = _csrlib()
    n = len(pairs)
    P = ctypes.POINTER(ctypes.c_int)
    rm = (P * n)(); cmm = (P * n)(); ro = (P * n)(); co = (P * n)()
    nent = (ctypes.c_int * n)(); ra = (ctypes.c_int * n)(); ca = (ctypes.c_int * n)(); nl = (ctypes.c_int * n)()
    keep = []
    for k, p in enumerate(pairs):
        r, c = np.ascontiguousarray(p[0], dtype=np.int32), np.ascontiguousarray(p[1], dtype=np.int32)
        keep += [r, c]
        rm[k] = r.ctypes.data_as(P); cmm[k] = c.ctypes.data_as(P)
        nent[k] = r.shape[0]; ra[k] = r.shape[1]; ca[k] = c.shape[1]
        if len(p) > 2 and p[2]:
            nl[k] = int(p[2])
            o1 = np.asarray(p[3], dtype=np.int32); o2 = np.asarray(p[4], dtype=np.int32)
            keep += [o1, o2]
            ro[k] = o1.ctypes.data_as(P); co[k] = o2.ctypes.data_as(P)
        else:
            nl[k] = 0
    rp = P(); ci = P()
    lib.oracle_build_node_sparsity.restype = ctypes.c_long
    nnz = lib.oracle_build_node_sparsity(nrow_nodes, ncol_nodes, int(set_diag and nrow_nodes == ncol_nodes or set_diag),
                                         n, rm, cmm, nent, ra, ca, nl, ro, co,
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
