"""Local kernels, global-kernel argument descriptors and the GlobalKernel itself.

Mirror of pyop2/local_kernel.py:20-227 (CStringLocalKernel) and
pyop2/global_kernel.py:27-456 (MapKernelArg ... GlobalKernel, compile_global_kernel).
``GlobalKernel.__call__(comm, start, end, *args)`` keeps the reference's calling
convention (global_kernel.py:327-335); what changes is what it compiles to: a HIP
wrapper kernel for gfx950 (codegen.py) loaded through the C ABI (fd_kernel_load).
"""
from __future__ import annotations

import ctypes
import hashlib
import re
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np

from . import _lib
from .op2types import Access, IterationRegion, ALL, READ

_C_TYPE = {np.dtype("float64"): "double", np.dtype("float32"): "float", np.dtype("int32"): "int",
           np.dtype("uint32"): "unsigned int", np.dtype("int64"): "int64_t", np.dtype("uint64"): "uint64_t"}


# ---- local kernel ---------------------------------------------------------------------------
@dataclass(frozen=True)
class LocalKernelArg:            # pyop2/local_kernel.py:20-40
    access: Access
    dtype: np.dtype


class CStringLocalKernel:
    """A local kernel given as a C string (pyop2/local_kernel.py:177-207).  This is the
    generic input of the backend: the same text is compiled by gcc for the CPU path of the
    reference and by hipcc (as a __device__ function) here."""

    def __init__(self, code, name, accesses=None, dtypes=None, *, flop_count=None, headers=(),
                 requires_zeroed_output_arguments=False, cpp=False, **_ignored):
        if not isinstance(code, str):
            raise TypeError("C-string local kernels need `code` to be a str (loopy kernels: LoopyLocalKernel)")
        if not isinstance(name, str):
            from .exceptions import NameTypeError
            raise NameTypeError("Kernel name must be a string")        # local_kernel.py:99 (validate_type)
        self.code = code
        self.name = name
        self.accesses = None if accesses is None else tuple(Access(a) for a in accesses)
        self.dtypes = None if dtypes is None else tuple(np.dtype(d) for d in dtypes)
        self.flop_count = flop_count
        self.headers = tuple(headers)
        self.requires_zeroed_output_arguments = requires_zeroed_output_arguments
        self.cpp = cpp

    @property
    def arguments(self):
        return tuple(LocalKernelArg(a, d) for a, d in zip(self.accesses, self.dtypes))

    def __str__(self):                     # local_kernel.py:172-176
        return f"OP2 Kernel: {self.name}"

    def __repr__(self):
        return 'Kernel("""%s""", %r)' % (self.code, self.name)

    @property
    def cache_key(self):
        return hashlib.md5((self.code + self.name + repr(self.accesses) + repr(self.dtypes) + repr(self.headers)
                            + repr(self.cpp) + repr(self.requires_zeroed_output_arguments)).encode()).hexdigest()

    def with_signature(self, accesses, dtypes):
        return CStringLocalKernel(self.code, self.name, accesses, dtypes, flop_count=self.flop_count,
                                  headers=self.headers,
                                  requires_zeroed_output_arguments=self.requires_zeroed_output_arguments, cpp=self.cpp)


# Includes that only make sense on the host side of the reference: PETSc's umbrella header (the types the wrappers
# use -- PetscScalar, PetscInt -- are typedef'd in csrc/fd_wrapper.h) and C99 <complex.h>, which loopy's C target
# emits unconditionally (pyop2/codegen/rep2loopy.py:565) but real-valued kernels never use.
_HOST_ONLY_INCLUDE = re.compile(r"^[ \t]*#[ \t]*include[ \t]*[<\"](petsc[a-z]*\.h|complex\.h)[>\"][ \t]*$", flags=re.M)


def strip_host_only_includes(code: str) -> str:
    return _HOST_ONLY_INCLUDE.sub("", code)


class LoopyLocalKernel(CStringLocalKernel):
    """pyop2/local_kernel.py:210-227: a local kernel given as a loopy ``LoopKernel``/``TranslationUnit`` -- what TSFC
    hands to PyOP2 (tsfc/loopy.py:216-283).  It is lowered ONCE on the host to C with ``lp.generate_code_v2``, exactly
    the text the reference compiles with gcc (global_kernel.py:408-423), and that text is compiled as a ``__device__``
    function: loopy's C dialect (``__restrict__`` pointers, ``int32_t`` loop counters, ``double const t0[...] = {...}``
    tabulation tables, ``static inline`` helper preambles) is valid HIP device code as it stands; only the host-only
    includes are dropped.  Register allocation of the temporaries and constant-table placement are hipcc's job.
    Needs the ``loopy`` package (absent from this image -- see ``loopy_c_kernel`` for text produced elsewhere)."""

    def __init__(self, code, name, accesses=None, dtypes=None, **kwargs):
        try:
            import loopy as lp
        except ImportError as exc:                         # pragma: no cover - loopy is not installed here
            raise ImportError("LoopyLocalKernel needs the 'loopy' package to lower the kernel to C; pass the generated "
                              "C text to loopy_c_kernel() instead") from exc
        self.loopy_code = code
        if dtypes is None:                                 # local_kernel.py:217-227
            knl = code.callables_table[name].subkernel if hasattr(code, "callables_table") else code
            dtypes = tuple(a.dtype.numpy_dtype if hasattr(a.dtype, "numpy_dtype") else a.dtype
                           for a in knl.args if isinstance(a, lp.ArrayArg))
        text = lp.generate_code_v2(code).device_code()
        super().__init__(text, name, accesses, dtypes, **kwargs)


def loopy_c_kernel(code: str, name: str, accesses=None, dtypes=None, **kwargs):
    """A local kernel from C text that loopy generated elsewhere (``lp.generate_code_v2(knl).device_code()``), e.g.
    captured from a Firedrake installation.  TSFC kernels accumulate into zeroed output tensors
    (tsfc/loopy.py:335-348, tsfc_interface.py:330-331), hence ``requires_zeroed_output_arguments`` defaults to True."""
    kwargs.setdefault("requires_zeroed_output_arguments", True)
    return CStringLocalKernel(code, name, accesses, dtypes, **kwargs)


class TensorProductLocalKernel(CStringLocalKernel):
    """A local kernel on a tensor-product (hexahedral Q_k) element whose element tensor is a quadrature contraction with a
    per-point 4 x 4 weight:

        matrix:  A_e = sum_q Phi_q^T W_q Phi_q              action:  y_e = sum_q Phi_q^T W_q (Phi_q u_e)

    ``Phi_q`` = (d/dxi_1, d/dxi_2, d/dxi_3, value) of the (degree+1)^3 basis functions at the nq^3 Gauss points, ``W_q``
    what ``weights_code`` computes from the cell geometry.  TSFC hands PyOP2 a sum-factorised scalar kernel for such forms
    (tsfc/spectral.py:157-191); ``code`` plays that role here -- an ordinary C kernel with the TSFC argument order
    (A, coords[, u]) that the oracle and the direct wrapper execute -- and the descriptor fields let the backend build
    the element tensor where it belongs on this chip: the matrix as a dense contraction on the fp64 matrix cores, the action
    sum-factorised out of LDS (csrc/fd_tensor.h).  ``weights_code`` defines

        static inline void <name>_weights(const double J[3][3], const double X[3], double wq, double W[16])

    (J[r][s] = dx_r/dxi_s at the point, X the physical point, wq the quadrature weight; W row-major, W[l*4+k] couples
    component l of the test side with component k of the trial side).  Instantiated for degree 1..8 and up to 11 Gauss points
    per axis (codegen.tensor_geometry); other descriptors take the ordinary wrappers on the C text.

    ``ncoef`` > 0: the form has coefficient arguments -- ``ncoef`` scalar READ Dats on the Q_k map after the standard arguments
    (A, coords, w_0 ... / y, coords, u, w_0 ...), the ``w_k`` TSFC passes to a variable-coefficient or linearised nonlinear form
    (tsfc/kernel_interface/firedrake_loopy.py:432-522).  The templates evaluate them at the Gauss points (sum-factorised) and the
    callback becomes ``<name>_weights(J, X, wq, const double *C, W)`` with C[m] = value of coefficient m at the point.

    ``coef_gradients``: the callback also receives the coefficients' REFERENCE gradients, ``<name>_weights(J, X, wq, C, DC, W)`` with
    DC[3*m + a] = d C[m] / d xi_a (physical gradient = K^T DC, K = J^-1) -- the linearisation of a form nonlinear in grad(u0)
    (tsfc/fem.py:742-805 tabulates the derivative tables for exactly this).

    ``vdim`` = D > 1: a vector-valued space (Q_k)^D -- the unknown (and y, u of the action) are Dats of dim D, the Mat has dims (D, D)
    and the element tensor is t[(i*D + p)][(j*D + r)] (MatSetValuesBlockedLocal, builder.py:573-625); the callback fills the 4D x 4D
    block weight W[((p*4 + l) * 4D) + r*4 + k] (reference component l of test component p against k of trial component r)."""

    def __init__(self, code, name, accesses=None, dtypes=None, *, kind, degree, nq, weights_code, ncoef=0, coef_gradients=False, vdim=1,
                 **kwargs):
        if kind not in ("matrix", "action"):
            raise ValueError("TensorProductLocalKernel kind must be 'matrix' or 'action'")
        kwargs.setdefault("requires_zeroed_output_arguments", True)
        super().__init__(code, name, accesses, dtypes, **kwargs)
        if not 1 <= int(vdim) <= 3:
            raise ValueError("TensorProductLocalKernel vdim must be 1, 2 or 3")
        self.tp = {"kind": kind, "degree": int(degree), "nq": int(nq), "weights_code": weights_code, "ncoef": int(ncoef),
                   "coef_gradients": bool(coef_gradients) and int(ncoef) > 0, "vdim": int(vdim)}

    @property
    def cache_key(self):
        return hashlib.md5((CStringLocalKernel.cache_key.fget(self) + repr(sorted(self.tp.items()))).encode()).hexdigest()

    def with_signature(self, accesses, dtypes):
        return TensorProductLocalKernel(self.code, self.name, accesses, dtypes, flop_count=self.flop_count, headers=self.headers,
                                        requires_zeroed_output_arguments=self.requires_zeroed_output_arguments, cpp=self.cpp, **self.tp)


def Kernel(code, name, **kwargs):
    """pyop2/local_kernel.py:54-83 ``Kernel`` factory: C strings and loopy kernels."""
    if isinstance(code, str):
        return CStringLocalKernel(code, name, **kwargs)
    if type(code).__name__ in ("LoopKernel", "TranslationUnit") or hasattr(code, "callables_table"):
        return LoopyLocalKernel(code, name, **kwargs)
    raise TypeError("code argument is the wrong type: expected a C string or a loopy kernel")


# ---- global kernel argument descriptors -----------------------------------------------------
@dataclass(eq=False, frozen=True)
class MapKernelArg:              # pyop2/global_kernel.py:27-52
    arity: int
    offset: Optional[Tuple[int, ...]] = None
    offset_quotient: Optional[Tuple[int, ...]] = None

    @property
    def cache_key(self):
        return type(self), self.arity, self.offset, self.offset_quotient


@dataclass(eq=False, frozen=True)
class PermutedMapKernelArg:      # pyop2/global_kernel.py:55-70
    base_map: MapKernelArg
    permutation: Tuple[int, ...]

    @property
    def arity(self):
        return self.base_map.arity

    @property
    def offset(self):
        return self.base_map.offset

    @property
    def cache_key(self):
        return type(self), self.base_map.cache_key, tuple(self.permutation)


@dataclass(frozen=True)
class GlobalKernelArg:           # pyop2/global_kernel.py:92-108
    dim: Tuple[int, ...]
    double: bool = False

    @property
    def cache_key(self):
        return type(self), self.dim

    @property
    def maps(self):
        return ()


@dataclass(frozen=True)
class DatKernelArg:              # pyop2/global_kernel.py:111-152
    dim: Tuple[int, ...]
    map_: object = None
    index: Optional[Tuple[int, ...]] = None

    @property
    def is_direct(self):
        return self.map_ is None

    @property
    def is_indirect(self):
        return not self.is_direct

    @property
    def cache_key(self):
        return type(self), self.dim, None if self.map_ is None else self.map_.cache_key, self.index

    @property
    def maps(self):
        return () if self.map_ is None else (self.map_,)


@dataclass(frozen=True)
class MatKernelArg:              # pyop2/global_kernel.py:155-180
    dims: Tuple[Tuple[int, ...], Tuple[int, ...]]
    maps: Tuple[object, object]
    unroll: bool = False
    lgmaps: bool = False          # backend extension: BC-masked lgmaps are passed (parloop.py:279-302)

    @property
    def cache_key(self):
        return type(self), self.dims, tuple(m.cache_key for m in self.maps), self.unroll, self.lgmaps


@dataclass(frozen=True)
class MixedDatKernelArg:         # pyop2/global_kernel.py:183-219
    arguments: Tuple[DatKernelArg, ...]

    def __iter__(self):
        return iter(self.arguments)

    def __len__(self):
        return len(self.arguments)

    @property
    def cache_key(self):
        return (type(self),) + tuple(a.cache_key for a in self.arguments)

    @property
    def maps(self):
        return tuple(m for a in self.arguments for m in a.maps)


@dataclass(frozen=True)
class MixedMatKernelArg:         # pyop2/global_kernel.py:222-252
    arguments: Tuple[MatKernelArg, ...]
    shape: Tuple[int, int]

    def __iter__(self):
        return iter(self.arguments)

    def __len__(self):
        return len(self.arguments)

    @property
    def cache_key(self):
        return (type(self), self.shape) + tuple(a.cache_key for a in self.arguments)

    @property
    def maps(self):
        return tuple(m for a in self.arguments for m in a.maps)


@dataclass(frozen=True)
class PassthroughKernelArg:      # pyop2/global_kernel.py:245-252
    @property
    def cache_key(self):
        return type(self)

    @property
    def maps(self):
        return ()


class GlobalKernel:
    """pyop2/global_kernel.py:255-405."""

    _cache = {}

    def __init__(self, local_kernel, arguments, *, extruded=False, extruded_periodic=False,
                 constant_layers=False, subset=False, iteration_region=None, pass_layer_arg=False):
        if local_kernel.accesses is None or len(local_kernel.accesses) != len(arguments):
            raise ValueError("Number of arguments passed to the local and global kernels do not match")
        if pass_layer_arg and not extruded:
            raise ValueError("Cannot request layer argument for non-extruded iteration")
        if constant_layers and not extruded:
            raise ValueError("Cannot request constant_layers argument for non-extruded iteration")
        if extruded_periodic and not extruded:
            raise ValueError("Cannot request extruded_periodic for non-extruded iteration")
        self.local_kernel = local_kernel
        self.arguments = tuple(arguments)
        self._extruded = extruded
        self._extruded_periodic = bool(extruded_periodic)
        self._constant_layers = constant_layers
        self._subset = subset
        self._iteration_region = IterationRegion(iteration_region) if iteration_region is not None else ALL
        self._pass_layer_arg = pass_layer_arg
        seen = {}
        map_ids = []
        for a in self.arguments:
            for m in a.maps:
                base = m.base_map if isinstance(m, PermutedMapKernelArg) else m
                map_ids.append(seen.setdefault(id(base), len(seen)))
        self.cache_key = (local_kernel.cache_key, *[a.cache_key for a in self.arguments], *map_ids,
                          extruded, constant_layers, subset, int(self._iteration_region), pass_layer_arg,
                          *((True,) if extruded_periodic else ()))
        self._compiled = {}

    @property
    def name(self):
        return f"wrap_{self.local_kernel.name}"

    @property
    def is_mixed(self):
        return any(isinstance(a, (MixedDatKernelArg, MixedMatKernelArg)) for a in self.arguments)

    def flattened(self):
        """The same loop with every Mixed argument split into its parts -- what the reference's builder does when it
        emits one pointer per part (builder.py:872-893, 904-916).  The local kernel keeps seeing ONE concatenated pack
        per mixed argument: an adaptor with the original name gathers the parts' packs into it before the call and
        copies them back afterwards (MixedDatPack, builder.py:432-518), resp. cuts the mixed element tensor into its
        blocks (MixedMatPack, builder.py:628-699).  The adaptor is inlined by hipcc; the packs stay in registers."""
        if not self.is_mixed:
            return self
        if getattr(self, "_flat", None) is None:
            nf = 2 if (self._extruded and self._iteration_region == IterationRegion.ON_INTERIOR_FACETS) else 1
            lk = self.local_kernel
            inner = lk.name + "__mixed"
            flat_args, accesses, dtypes, params, pre, call, post = [], [], [], [], [], [], []
            for k, (a, la) in enumerate(zip(self.arguments, lk.arguments)):
                ct = _C_TYPE[np.dtype(la.dtype)]
                if isinstance(a, MixedDatKernelArg):
                    sizes = []
                    for pa in a:
                        if pa.map_ is None:
                            raise NotImplementedError("direct (map-less) MixedDat arguments")        # as builder.py:441
                        sizes.append(nf * pa.map_.arity * int(np.prod(pa.dim)))
                    pre.append(f"{ct} m{k}[{sum(sizes)}];")
                    off = 0
                    for p_, (pa, n) in enumerate(zip(a, sizes)):
                        flat_args.append(pa); accesses.append(la.access); dtypes.append(la.dtype)
                        params.append(f"{ct} *a{k}_{p_}")
                        pre.append(f"for (int q = 0; q < {n}; ++q) m{k}[{off} + q] = a{k}_{p_}[q];")
                        if la.access != READ:
                            post.append(f"for (int q = 0; q < {n}; ++q) a{k}_{p_}[q] = m{k}[{off} + q];")
                        off += n
                    call.append(f"m{k}")
                elif isinstance(a, MixedMatKernelArg):
                    nr, nc = a.shape
                    blocks = [a.arguments[i * nc:(i + 1) * nc] for i in range(nr)]
                    rows = [nf * row[0].maps[0].arity * int(np.prod(row[0].dims[0])) for row in blocks]
                    cols = [nf * b.maps[1].arity * int(np.prod(b.dims[1])) for b in blocks[0]]
                    R, C = sum(rows), sum(cols)
                    pre.append(f"{ct} m{k}[{R * C}]; for (int q = 0; q < {R * C}; ++q) m{k}[q] = 0;")
                    ro = 0
                    for i, row in enumerate(blocks):
                        co = 0
                        for j, b in enumerate(row):
                            flat_args.append(b); accesses.append(la.access); dtypes.append(la.dtype)
                            params.append(f"{ct} *a{k}_{i}_{j}")
                            post.append(f"for (int i = 0; i < {rows[i]}; ++i) for (int j = 0; j < {cols[j]}; ++j) "
                                        f"a{k}_{i}_{j}[i*{cols[j]} + j] = m{k}[({ro} + i)*{C} + {co} + j];")
                            co += cols[j]
                        ro += rows[i]
                    call.append(f"m{k}")
                else:
                    flat_args.append(a); accesses.append(la.access); dtypes.append(la.dtype)
                    params.append(f"{ct} *a{k}")
                    call.append(f"a{k}")
            if self._pass_layer_arg:
                params.append("int layer")
                call.append("layer")
            body = "\n  ".join(pre + [f"{inner}({', '.join(call)});"] + post)
            code = (f"#define {lk.name} {inner}\n{lk.code}\n#undef {lk.name}\n"
                    f"static inline void {lk.name}({', '.join(params)})\n{{\n  {body}\n}}\n")
            flk = CStringLocalKernel(code, lk.name, accesses, dtypes, flop_count=lk.flop_count, headers=lk.headers,
                                     requires_zeroed_output_arguments=lk.requires_zeroed_output_arguments, cpp=lk.cpp)
            self._flat = GlobalKernel(flk, flat_args, extruded=self._extruded, extruded_periodic=self._extruded_periodic,
                                      constant_layers=self._constant_layers, subset=self._subset,
                                      iteration_region=self._iteration_region, pass_layer_arg=self._pass_layer_arg)
        return self._flat

    def compile(self, mode=None):
        """compile_global_kernel (global_kernel.py:426-456): codegen -> hipcc -> code object ->
        fd_kernel_load.  Returns a :class:`CompiledWrapper`."""
        if self.is_mixed:
            return self.flattened().compile(mode)
        from .codegen import generate_wrapper, select_mode
        from .compilation import compile_hip
        mode = mode or select_mode(self)
        if mode == "auto":
            mode = select_mode(self)
        cw = self._compiled.get(mode)
        if cw is None:
            ck = (self.cache_key, mode)
            cw = GlobalKernel._cache.get(ck)
            if cw is None:
                src = generate_wrapper(self, mode)

                def build(src=src, mode=mode):
                    # hipcc runs when the code object is first needed: a wrapper that only lends its argument layout to the
                    # geometry-specific variant a loop ends up launching (parloop._staged_geometry / _ocr_geometry) is never compiled
                    path = compile_hip(src.source, self.name)
                    path = self._unrolled_variant(src, path)
                    return self._occupancy_variant(mode, src, path)

                cw = CompiledWrapper(src, builder=build)
                GlobalKernel._cache[ck] = cw
            self._compiled[mode] = cw
        return cw

    def _unrolled_variant(self, src, path):
        """The wrappers keep the element tensor and the packs in registers, which needs every loop of the local kernel fully
        unrolled (constant indices); LLVM's default threshold gives up on larger nests -- a 12x12 vector-P1 element matrix
        written as four nested loops already stays in scratch memory and runs 20x slower.  So when hipcc reports scratch,
        the wrapper is compiled once more with a high ``-unroll-threshold`` and that code object is kept if its scratch is
        smaller (genuine register spills are not cured by unrolling; then the first one stays).  The flag is recorded on
        the source so that later variants of this wrapper are built the same way."""
        from .compilation import compile_hip, kernel_resources
        from .configuration import configuration
        thr = configuration["unroll_retry_threshold"]
        res = kernel_resources(path, src.symbol)
        if thr <= 0 or not res or not (0 < res.get("scratch", 0) <= configuration["unroll_retry_max_scratch"]):
            return path                 # (a register file holds 2 KB per lane: far larger tensors stay where they are)
        extra = ("-mllvm", f"-unroll-threshold={thr}")
        path2 = compile_hip(src.source, self.name, extra)
        res2 = kernel_resources(path2, src.symbol)
        if res2 and res2.get("scratch", 1 << 30) < res["scratch"]:
            src.extra_flags = extra
            return path2
        return path

    def _occupancy_variant(self, mode, src, path):
        """Workgroups of T lanes put T/256 wavefronts on every SIMD, so the resident wavefronts per SIMD go up in steps
        of T/256.  If the wrapper's register count is what stops the next step, recompile it with
        ``__launch_bounds__(T, next step)`` and keep the variant when the registers it gives up cost (almost) no
        scratch; otherwise keep the original.  Decided from hipcc's own resource report, once per JIT compilation."""
        from .codegen import generate_wrapper
        from .compilation import compile_hip, kernel_resources
        from .configuration import configuration
        limit = configuration["auto_occupancy_scratch"]
        if limit < 0 or configuration["min_waves"] or not (mode.startswith("staged") or mode.startswith("ocr")) or mode.startswith("ocrs"):
            return src, path            # (row-sliced loops are LDS-limited: ~90 registers per lane)
        res = kernel_resources(path, src.symbol)
        if not res or "occupancy" not in res:
            return src, path
        step = max(1, src.block_threads // 256)
        target = (res["occupancy"] // step + 1) * step
        if target > 8 or res.get("vgprs", 0) <= 512 // target:
            return src, path            # already at the hardware limit / not limited by registers
        src2 = generate_wrapper(self, mode, min_waves=target)
        src2.extra_flags = src.extra_flags
        path2 = compile_hip(src2.source, self.name, src.extra_flags)
        res2 = kernel_resources(path2, src2.symbol)
        if res2 and res2.get("occupancy", 0) >= target and res2.get("scratch", 1 << 30) <= limit:
            return src2, path2
        return src, path

    def __call__(self, comm, start, end, *args, **launch):
        """func(start, end, *arglist) -- global_kernel.py:327-335."""
        cw = self.compile(launch.pop("mode", None))
        cw.launch(start, end, args, **launch)



# symbol -> (work-items, workgroup size) of the most recent launch of a wrapper of that name (diagnostics; see CompiledWrapper.launch)
last_launch = {}

class CompiledWrapper:
    """A loaded wrapper kernel + the layout of its argument list."""

    def __init__(self, src, hsaco_path=None, builder=None):
        """``builder`` () -> (final WrapperSource, code-object path): the compilation, deferred until ``path`` / ``handle`` is first
        read.  ``src`` before that is the generated wrapper (argument layout, staged maps, LDS items, block size: everything
        the plan builders read); afterwards the variant that was kept (unroll / occupancy retries change flags and launch
        bounds only)."""
        self._src = src
        self._path = hsaco_path
        self._builder = builder
        self._handle = None

    def _build(self):
        if self._path is None:
            self._src, self._path = self._builder()
            self._builder = None

    @property
    def src(self):
        return self._src

    @property
    def path(self):
        self._build()
        return self._path

    @property
    def handle(self):
        if self._handle is None:
            _lib.require_gpu()
            h = ctypes.c_void_p()
            _lib.call("fd_kernel_load", self.path.encode(), self.src.symbol.encode(), ctypes.byref(h))
            self._handle = h.value
        return self._handle

    def launch(self, start, end, args, *, block_threads=256, ents_per_block=256, nblocks=-1, lds_bytes=0, stream=None):
        n = len(args)
        if int(end) > int(start):
            # work-items of this launch (what a profiler reports as the dispatch's grid size): lets a measurement script tell the
            # launches of one wrapper on different problem sizes apart (bench.py keys its PMC results by (kernel, grid))
            nb_ = int(nblocks) if int(nblocks) > 0 else -(-(int(end) - int(start)) // int(ents_per_block))
            last_launch[self.src.symbol] = (nb_ * int(block_threads), int(block_threads))
        arr = (ctypes.c_void_p * max(n, 1))(*[int(a) if a is not None else None for a in args])       # (plain ints: no c_void_p object per slot)
        _lib.call("fd_kernel_launch", self.handle, int(start), int(end), arr, n, int(block_threads),
                  int(ents_per_block), int(nblocks), int(lds_bytes), stream)
