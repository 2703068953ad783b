"""The PyOP2 side of the seam: turn the reference's own kernel descriptors into this
backend's and hand back a callable with the reference's ``func(start, end, *arglist)`` signature.

PyOP2 cannot be imported here (``loopy``, ``petsc4py`` and ``mpi4py`` are absent), so nothing in this module imports
it: the descriptors are recognised by *class name and attributes* -- exactly the fields ``pyop2/global_kernel.py:27-325``
and ``pyop2/local_kernel.py:86-227`` define -- which is also what makes the translation testable with stand-in objects
(tests/test_bridge.py).  Two levels of integration exist:

* **function level** (this module): ``compile_global_kernel_hip(kernel, comm)`` replaces
  ``pyop2.global_kernel.compile_global_kernel`` (global_kernel.py:426-456).  The returned callable takes the reference's
  positional list -- ``start, end, [layers], [subset], one pointer per Dat/Global/Mat, one per distinct Map`` -- with
  DEVICE pointers (the carriers' ``_kernel_args_`` patched as INTEGRATION.md shows; a Mat slot carries the handle of a
  ``DeviceMat``, the device CSR that stands where the PETSc ``Mat`` handle stood) and runs the same staged /
  owner-computes-rows / direct wrappers the native Parloop picks: the backend-private plan tables are resolved inside
  ``func``, keyed on the Map pointers, the Mat handle and the iteration range, and cached across calls.
* **Parloop level**: ``firedrake_amd.parloop.Parloop`` is the worked replacement of ``pyop2.parloop.Parloop`` (same
  protocol, halo exchanges included); ``as_fd_global_kernel`` is what it needs from PyOP2's side.
"""
from __future__ import annotations

import numpy as np

from . import kernel as K
from .op2types import Access, IterationRegion


def _name(obj):
    return type(obj).__name__


def _tuple(x):
    return None if x is None else tuple(int(v) for v in x)


def _map_arg(m, memo):
    """MapKernelArg / PermutedMapKernelArg (global_kernel.py:27-70).  Identity matters: the reference de-duplicates
    map pointers by object identity (global_kernel.py:309-314), so one source object gives one target object."""
    if m is None:
        return None
    hit = memo.get(id(m))
    if hit is not None:
        return hit
    n = _name(m)
    if n == "MapKernelArg":
        out = K.MapKernelArg(int(m.arity), _tuple(getattr(m, "offset", None)), _tuple(getattr(m, "offset_quotient", None)))
    elif n == "PermutedMapKernelArg":
        out = K.PermutedMapKernelArg(_map_arg(m.base_map, memo), tuple(int(p) for p in m.permutation))
    elif n == "ComposedMapKernelArg":
        # global_kernel.py:73-90, types/map.py:206-270: local[i] = global[maps_[0][maps_[1][...]]][i], inner maps of arity 1.
        # The reference indexes through the chain inside the wrapper (builder.py:179-198) and passes one pointer per
        # constituent; here the chain is composed ONCE on the device into an ordinary table (``_composed_table``): the
        # wrapper sees a plain map, ``func`` consumes the constituents' pointers.
        chain = []
        for b in m.base_maps:
            chain.extend(getattr(b, "base_maps", None) or [b]) if _name(b) == "ComposedMapKernelArg" else chain.append(b)
        if any(_name(b) != "MapKernelArg" for b in chain):
            raise NotImplementedError("ComposedMapKernelArg over permuted maps")
        if any(int(b.arity) != 1 for b in chain[1:]):
            raise ValueError("ComposedMapKernelArg: every map but the first must have arity 1 (types/map.py:246-247)")
        first = chain[0]
        out = K.MapKernelArg(int(first.arity), _tuple(getattr(first, "offset", None)), _tuple(getattr(first, "offset_quotient", None)))
        memo.setdefault("composed", {})[id(out)] = tuple(chain)
        for b in chain:
            memo.setdefault("leaves", {}).setdefault(id(b), b)
    else:
        raise TypeError(f"unknown map kernel argument {n}")
    if n == "MapKernelArg":
        memo.setdefault("leaves", {}).setdefault(id(m), m)
        memo.setdefault("leaf_of", {})[id(out)] = m
    memo[id(m)] = out
    return out


def _dim(d):
    if d is None or d == ():
        return (1,)
    return (int(d),) if np.isscalar(d) else tuple(int(v) for v in d)


def _kernel_arg(a, memo):
    n = _name(a)
    if n == "DatKernelArg":
        return K.DatKernelArg(_dim(a.dim), _map_arg(getattr(a, "map_", None), memo), _tuple(getattr(a, "index", None)))
    if n == "GlobalKernelArg":
        return K.GlobalKernelArg(_dim(a.dim), bool(getattr(a, "double", False)))
    if n == "MatKernelArg":
        (rd, cd), = a.dims if len(a.dims) == 1 else (a.dims,)          # ((rdim, cdim),) in the reference (mat.py:203-212)
        return K.MatKernelArg((rd, cd), tuple(_map_arg(m, memo) for m in a.maps), unroll=bool(getattr(a, "unroll", False)))
    if n == "MixedDatKernelArg":
        return K.MixedDatKernelArg(tuple(_kernel_arg(x, memo) for x in a.arguments))
    if n == "MixedMatKernelArg":
        return K.MixedMatKernelArg(tuple(_kernel_arg(x, memo) for x in a.arguments), tuple(int(v) for v in a.shape))
    if n == "PassthroughKernelArg":
        return K.PassthroughKernelArg()
    raise TypeError(f"unknown global kernel argument {n}")


def as_fd_local_kernel(lk):
    """CStringLocalKernel / LoopyLocalKernel (local_kernel.py:86-227) -> this backend's local kernel."""
    kw = dict(flop_count=getattr(lk, "flop_count", None), headers=tuple(getattr(lk, "headers", ()) or ()),
              requires_zeroed_output_arguments=bool(getattr(lk, "requires_zeroed_output_arguments", False)),
              cpp=bool(getattr(lk, "cpp", False)))
    accesses = tuple(Access(int(a)) for a in lk.accesses)
    dtypes = tuple(np.dtype(d) for d in lk.dtypes)
    tp = getattr(lk, "fdhip_tensor", None)
    if tp is not None:
        # a tensor-product form the Firedrake-side patch recognised (tsfc_interface.py attaches the
        # descriptor next to the TSFC kernel it wraps): same C text, plus what the matrix-core wrappers need
        code = lk.code if isinstance(lk.code, str) else K.LoopyLocalKernel(lk.code, lk.name, accesses, dtypes, **kw).code
        return tensor_product_local_kernel(code, lk.name, accesses, dtypes, tp, **kw)
    if isinstance(lk.code, str):
        return K.CStringLocalKernel(lk.code, lk.name, accesses, dtypes, **kw)
    return K.LoopyLocalKernel(lk.code, lk.name, accesses, dtypes, **kw)      # lowered with lp.generate_code_v2


def tensor_form_info(cell, family, degree, quadrature_degree, terms, kind, value_size=1):
    """The descriptor a Firedrake-side patch attaches to the pyop2 local kernel of a tensor-product form
    (``kernel.fdhip_tensor = tensor_form_info(...)``).  Everything in it is form data Firedrake holds
    when it builds the kernel (firedrake/tsfc_interface.py:84-140):

    ``cell``, ``family``, ``degree``   ``V.ufl_element()``: cellname, family and degree of the (test = trial) space
    ``quadrature_degree``              the integral's ``metadata["quadrature_degree"]`` / TSFC's estimate (tsfc/driver.py)
    ``terms``                          the integrand as a sum of second-order terms with constant coefficients:
                                       ``{"stiffness": alpha, "mass": beta, "advection": (bx, by, bz)}`` for
                                       alpha*inner(grad u, grad v) + inner(dot(b, grad u), v) + beta*inner(u, v)
                                       -- or, for a variable-coefficient / linearised nonlinear form,
                                       ``{"stiffness": "<C expr>", "mass": "<C expr>", "coefficients": n}``: the two factors as
                                       C expressions in ``C[0..n)`` -- the values at the point of the form's n coefficient
                                       Functions, each on the SAME space or on the Q1 space of the coordinates (the backend tells
                                       them apart by the Map of the argument), in the order TSFC passes them (``w_0 ...``,
                                       firedrake_loopy.py:432-522) -- and ``X[0..2]`` (the physical point); what the UFL
                                       integrand ``kappa(w0)*inner(grad(du), grad(v)) + c(u0)*du*v`` prints as
                                       -- or a form family the backend holds the point weight of:
                                       ``{"elasticity": (mu, lam), "mass": rho}`` for 2 mu eps(u):eps(v) + lam div(u) div(v) + rho u.v on a
                                       VectorFunctionSpace (``value_size`` 3), ``{"nonlinear_diffusion": 1}`` for the Newton Jacobian of
                                       (1 + |grad u|^2) grad(u).grad(v) at the form's one coefficient (its GRADIENT at the points)
                                       -- or, in general, ``{"weights_code": "<C>", "coefficients": n, "coefficient_gradients": bool}``:
                                       the C text of ``<NAME>_weights`` (kernel.TensorProductLocalKernel documents its signatures; the
                                       literal NAME is replaced by the kernel's name) filling the 4 value_size x 4 value_size point weight
    ``kind``                           "matrix" for a 2-form, "action" for action(a, u) / a 1-form linear in one coefficient
    ``value_size``                     ``V.value_size``: 1 for a scalar space, 2 or 3 for a VectorFunctionSpace on the same element

    Returns None when the form is not one the tensor wrappers cover (the loop then takes the ordinary wrappers)."""
    nq = int(quadrature_degree) // 2 + 1                   # Gauss-Legendre points per axis exact for that degree
    from .codegen import tensor_geometry
    ok = (cell in ("hexahedron", "quadrilateral * interval", "TensorProductCell(quadrilateral, interval)") and family in ("Q", "CG", "Lagrange")
          and tensor_geometry(int(degree), nq) is not None and kind in ("matrix", "action")
          and set(terms) <= {"stiffness", "mass", "advection", "coefficients", "elasticity", "nonlinear_diffusion", "weights_code",
                             "coefficient_gradients"}
          and int(value_size) in (1, 2, 3))
    if not ok:
        return None
    base = {"kind": kind, "degree": int(degree), "nq": nq}
    if "weights_code" in terms:
        if set(terms) - {"weights_code", "coefficients", "coefficient_gradients"}:
            return None
        return dict(base, weights_code=str(terms["weights_code"]), ncoef=int(terms.get("coefficients", 0)),
                    coef_gradients=bool(terms.get("coefficient_gradients", False)), vdim=int(value_size))
    if "elasticity" in terms:
        if set(terms) - {"elasticity", "mass"} or int(value_size) != 3:
            return None
        mu, lam = terms["elasticity"]
        return dict(base, family="elasticity", mu=float(mu), lam=float(lam), rho=float(terms.get("mass", 0.0)), vdim=3)
    if "nonlinear_diffusion" in terms:
        if set(terms) - {"nonlinear_diffusion"} or int(value_size) != 1:
            return None
        return dict(base, family="nonlinear_diffusion", ncoef=1, coef_gradients=True)
    if int(value_size) != 1:
        return None
    if "coefficients" in terms:
        if "advection" in terms or int(terms["coefficients"]) < 0:
            return None
        return {"kind": kind, "degree": int(degree), "nq": nq, "ncoef": int(terms["coefficients"]),
                "kappa": str(terms.get("stiffness", "0.0")), "react": str(terms.get("mass", "0.0"))}
    info = {"kind": kind, "degree": int(degree), "nq": nq, "alpha": float(terms.get("stiffness", 0.0)), "beta": float(terms.get("mass", 0.0))}
    if "advection" in terms:
        info["velocity"] = tuple(float(v) for v in terms["advection"])
    return info


def tensor_product_local_kernel(code, name, accesses, dtypes, info, **kw):
    """TensorProductLocalKernel from the TSFC kernel text and a ``tensor_form_info`` descriptor: the per-point weight
    callback of alpha*inner(grad u, grad v) + beta*inner(u, v) is W = w|J| [alpha K K^T, 0; 0, beta] (tensor.py)."""
    from .tensor import coefficient_weights, elasticity_weights, nonlinear_diffusion_weights, second_order_weights
    kw.setdefault("requires_zeroed_output_arguments", True)
    common = dict(kind=info["kind"], degree=info["degree"], nq=info["nq"])
    if "weights_code" in info:
        return K.TensorProductLocalKernel(code, name, accesses, dtypes, ncoef=info["ncoef"], coef_gradients=info["coef_gradients"],
                                          vdim=info["vdim"], weights_code=info["weights_code"].replace("NAME", name), **common, **kw)
    if info.get("family") == "elasticity":
        return K.TensorProductLocalKernel(code, name, accesses, dtypes, vdim=3,
                                          weights_code=elasticity_weights(name, info["mu"], info["lam"], info["rho"]), **common, **kw)
    if info.get("family") == "nonlinear_diffusion":
        return K.TensorProductLocalKernel(code, name, accesses, dtypes, ncoef=1, coef_gradients=True,
                                          weights_code=nonlinear_diffusion_weights(name), **common, **kw)
    if "ncoef" in info:
        return K.TensorProductLocalKernel(code, name, accesses, dtypes, kind=info["kind"], degree=info["degree"], nq=info["nq"],
                                          ncoef=info["ncoef"], weights_code=coefficient_weights(name, info["kappa"], info["react"]), **kw)
    return K.TensorProductLocalKernel(code, name, accesses, dtypes, kind=info["kind"], degree=info["degree"], nq=info["nq"],
                                      weights_code=second_order_weights(name, info["alpha"], info["beta"], info.get("velocity", (0.0, 0.0, 0.0))),
                                      **kw)


def as_fd_global_kernel(gk):
    """pyop2 ``GlobalKernel`` (global_kernel.py:255-325) -> ``firedrake_amd.kernel.GlobalKernel`` with the same local
    kernel, argument descriptors (map identity preserved) and iteration flags."""
    memo = {}
    args = [_kernel_arg(a, memo) for a in gk.arguments]
    region = getattr(gk, "_iteration_region", None)
    fd = _global_kernel(gk, args, region)
    # how the reference's map pointers (one per distinct LEAF map, parloop.py:203-212) feed this kernel's maps
    fd._seam_maps = {"leaves": list(memo.get("leaves", {}).values()), "leaf_of": memo.get("leaf_of", {}),
                     "composed": memo.get("composed", {})}
    return fd


def _global_kernel(gk, args, region):
    return K.GlobalKernel(as_fd_local_kernel(gk.local_kernel), args,
                          extruded=bool(getattr(gk, "_extruded", False)),
                          extruded_periodic=bool(getattr(gk, "_extruded_periodic", False)),
                          constant_layers=bool(getattr(gk, "_constant_layers", False)),
                          subset=bool(getattr(gk, "_subset", False)),
                          iteration_region=None if region is None else IterationRegion(int(region)),
                          pass_layer_arg=bool(getattr(gk, "_pass_layer_arg", False)))


# ---- the function-level seam ------------------------------------------------------------------------------------------
# The reference's call is ``func(start, end, *arglist)`` with raw pointers taken from the carriers' ``_kernel_args_``
# (pyop2/parloop.py:203-232, global_kernel.py:327-335): Dat -> data pointer (dat.py:94-96), Map -> values pointer
# (map.py:57-59), Global -> data pointer (glob.py:32-33), ExtrudedSet -> layers pointer (set.py:351-353), Subset ->
# indices pointer (set.py:434-436), Mat -> the PETSc ``Mat`` handle (mat.py:621-623).  The replacement receives DEVICE
# pointers in the same slots; in a Mat slot it receives the handle of a ``DeviceMat`` (below): the device CSR the
# carrier-side patch created for that matrix.  Everything backend-private -- block-localisation plans keyed on the Map
# pointer and the iteration range, owner-computes-rows plans keyed on (Mat handle, Map pointers, range), element->nonzero
# tables -- is resolved INSIDE ``func`` and cached, so the reference's Parloop needs no change beyond the carrier patches.

_device_mats = {}
_device_maps = {}


class DeviceMat:
    """Device CSR behind a PyOP2 ``Mat`` (the ``handle`` is what the patched ``Mat._kernel_args_`` returns).

    ``rowptr/colidx/values`` are device pointers of the scalar ("aij") pattern; ``nrows_owned`` = rows this rank
    assembles (the row Set's ``size``; the CSR holds ``nrows`` >= that).  Vector-valued spaces (``rbs * cbs > 1``,
    MatSetValuesBlockedLocal, builder.py:573-625): ``rowptr/colidx`` describe the NODE pattern (``nrows`` node rows, ``nnz``
    node entries) and ``values`` holds the scalar CSR of the block-expanded matrix (``nnz * rbs * cbs`` doubles, scalar row
    ``node * rbs + p`` = the node row's entries expanded by ``cbs``, columns ascending) -- what a PETSc (S)AIJ matrix on a
    vector space stores; the expanded index arrays are derived here (fd_csr_expand_blocks).  ``set_lgmaps`` mirrors the lgmap swap the
    reference performs around the wrapper call for boundary conditions (pyop2/parloop.py:279-314): device int32 arrays
    with -1 for dropped rows/columns, or None.  ``zero()`` is ``Mat.zero()`` (mat.py:851-855): deferred, an
    owner-computes-rows assembly that follows overwrites complete rows instead of memset + add."""

    _next = [1]

    def __init__(self, rowptr, colidx, values, nrows, nnz, nrows_owned=None, ncols=None, rbs=1, cbs=1, rowptr_bytes=4):
        """``rowptr_bytes``: size of the row starts behind ``rowptr`` -- PetscInt of the PETSc build, 4 or 8 (pyop2/datatypes.py:6-10).
        The backend's row starts are ``fd_nnz_t`` (64-bit); a 32-bit array is widened once into a buffer this object owns."""
        from . import _lib
        from .device import DeviceBuffer
        self.handle = (DeviceMat._next[0] << 4) | 0xD            # never a plausible device address
        DeviceMat._next[0] += 1
        self.nrows, self.ncols, self.nnz = int(nrows), int(ncols if ncols is not None else nrows), int(nnz)
        self.nrows_owned = int(nrows if nrows_owned is None else nrows_owned)
        self.rbs, self.cbs = int(rbs), int(cbs)
        wrap = lambda p, n: DeviceBuffer.wrap(int(p), int(n), owned=False)       # noqa: E731
        self.colidx = wrap(colidx, max(nnz, 1) * 4)
        if int(rowptr_bytes) == _lib.NNZ_BYTES:
            self.rowptr = wrap(rowptr, (self.nrows + 1) * _lib.NNZ_BYTES)
        elif int(rowptr_bytes) == 4:
            narrow = wrap(rowptr, (self.nrows + 1) * 4).download(np.int32, (self.nrows + 1,))
            self.rowptr = DeviceBuffer.from_numpy(narrow.astype(_lib.NNZ_DTYPE))
        else:
            raise ValueError("DeviceMat: rowptr_bytes must be 4 or 8")
        self.values = wrap(values, max(nnz, 1) * 8 * self.rbs * self.cbs)
        self._scalar = None
        self.lgmaps = None
        self._carriers = {}
        self._zero_requested = False
        _device_mats[self.handle] = self

    def set_lgmaps(self, row_ptr, col_ptr, token=None):
        """Install the (row, column) lgmap pair of the next calls (pyop2/parloop.py:279-314 swaps them per assemble).  The
        row-sliced wrapper folds a pair into per-instance tables cached by (pointer, ``token``): lgmaps are immutable in the
        reference (PETSc LGMaps); if a device array is ever REWRITTEN in place, pass a new ``token`` with it."""
        if (row_ptr is None) != (col_ptr is None):
            raise ValueError("set_lgmaps: give both the row and the column lgmap (identity = arange), or neither")
        self.lgmaps = None if row_ptr is None else (_RawIntArray(row_ptr, token), _RawIntArray(col_ptr, token))

    def zero(self):
        self._zero_requested = True

    def scalar_pattern(self):
        """(rowptr, colidx, nnz) of the scalar CSR the values are laid out in."""
        if self.rbs * self.cbs == 1:
            return self.rowptr, self.colidx, self.nnz
        if self._scalar is None:
            import ctypes
            from . import _lib
            from .device import DeviceBuffer
            rp2, ci2 = ctypes.c_void_p(), ctypes.c_void_p()
            _lib.call("fd_csr_expand_blocks", self.nrows, self.rowptr.ptr, self.colidx.ptr, self.rbs, self.cbs,
                      ctypes.byref(rp2), ctypes.byref(ci2), None)
            n2 = self.nnz * self.rbs * self.cbs
            self._scalar = (DeviceBuffer.wrap(rp2.value, (self.nrows * self.rbs + 1) * _lib.NNZ_BYTES), DeviceBuffer.wrap(ci2.value, max(n2, 1) * 4), n2)
        return self._scalar

    def free(self):
        _device_mats.pop(self.handle, None)


def register_map(ptr, nent_total, arity, toset_sizes, iterset_sizes=None, preferred_blocks=None, preferred_node_blocks=None,
                 values=None):
    """Optional: tell the backend what it cannot read off a bare Map pointer -- the (core, owned, total) sizes of the Set
    the Map points into (needed when a Mat is assembled through it: only owned rows are assembled here), the number of
    map rows, the producer's block hints, and -- for loops over subsets / extruded sets, whose plans live on DERIVED maps
    (one row per subset entity, resp. per (column, layer) cell) -- the host values the PyOP2 Map carries anyway
    (``Map.values_with_halo``; without them the table is downloaded once).  Unregistered maps work for Dat-only loops
    over plain sets."""
    _device_maps[int(ptr)] = {"nent": int(nent_total), "arity": int(arity), "toset": tuple(int(v) for v in np.atleast_1d(toset_sizes)),
                              "iterset": None if iterset_sizes is None else tuple(int(v) for v in np.atleast_1d(iterset_sizes)),
                              "pb": preferred_blocks, "pnb": preferred_node_blocks,
                              "values": None if values is None else np.ascontiguousarray(values, dtype=np.int32)}


def forget(ptr):
    """Forget a device address (a Map, Subset or layers buffer the carrier is about to free): drops the registration and
    every cached loop, plan, host copy and composed table keyed on it, so a later allocation at the same address starts
    clean.  The carrier-side patch calls this from the carriers' finalisers: everything the seam caches
    is keyed on addresses because addresses are all ``func(start, end, *arglist)`` receives."""
    ptr = int(ptr)
    _device_maps.pop(ptr, None)
    for k in [k for k in _host_ints_cache if k[0] == ptr]:
        _host_ints_cache.pop(k)
    for k in [k for k in _composed if ptr in k]:
        _composed.pop(k)
    for f in list(_seam_funcs):
        for k in [k for k in f.loops if ptr in k]:
            f.loops.pop(k)


unregister_map = forget


def reset():
    """Drop every address-keyed cache of the seam (all registrations, host copies, composed tables, cached loops)."""
    _device_maps.clear()
    _host_ints_cache.clear()
    _composed.clear()
    for f in _seam_funcs:
        f.loops.clear()


_host_ints_cache = {}
_composed = {}
_seam_funcs = []


def _host_ints(ptr, n, shape=None):
    """int32 values behind a device pointer the host needs ONCE per pointer (layers, subset indices, map rows of loops over
    virtual iteration spaces): in PyOP2 these live in host arrays of the carriers; the seam only sees their addresses."""
    from . import _lib
    key = (int(ptr), int(n))
    a = _host_ints_cache.get(key)
    if a is None:
        a = np.empty(int(n), dtype=np.int32)
        if n:
            _lib.call("fd_memcpy_d2h", a.ctypes.data, int(ptr), a.nbytes, None)
            _lib.call("fd_stream_sync", None)
        _host_ints_cache[key] = a
    return a if shape is None else a.reshape(shape)


def _composed_table(ptrs, arity0, n):
    """Device table of a ComposedMap (types/map.py:206-270) from its constituents' pointers: the arity-1 chain is followed
    with row gathers (negative = undefined entries propagate), then the first map's rows are gathered."""
    from . import _lib
    from .device import DeviceBuffer
    key = tuple(int(p) for p in ptrs) + (int(n),)
    t = _composed.get(key)
    if t is None:
        cur, keep = int(ptrs[-1]), []
        for p in reversed(ptrs[1:-1]):
            nxt = DeviceBuffer(max(n, 1) * 4)
            _lib.call("fd_gather_rows", int(p), 1, cur, n, nxt.ptr, None)
            keep.append(nxt)
            cur = nxt.ptr
        t = DeviceBuffer(max(n, 1) * arity0 * 4)
        _lib.call("fd_gather_rows", int(ptrs[0]), arity0, cur, n, t.ptr, None)
        _lib.call("fd_stream_sync", None)
        _composed[key] = t
    return t


class _RawIntArray:
    """A device int32 array known only by its pointer (Parloop._lgmap accepts objects with ``_fd_dev_ptr``)."""

    def __init__(self, ptr, token=None):
        self._fd_dev_ptr = int(ptr) if ptr is not None else 0
        self._fd_token = token


class _Shape:
    """Stands in for a host array nobody may read: only ``shape``/``len`` are known."""

    def __init__(self, shape):
        self.shape = tuple(shape)

    def __len__(self):
        return self.shape[0]


def _borrowed_carriers(fd, arglist, start, end):
    """Carriers of this backend (Set/Map/Dat/Global/Sparsity/Mat) that BORROW the device memory behind ``arglist`` (this
    kernel's positional list: composed maps already resolved to one table each)."""
    from . import op2
    from .device import DeviceBuffer
    from .parloop import DatParloopArg, GlobalParloopArg, MatParloopArg
    from .op2types import Map, Mat, Sparsity
    it = iter(arglist)
    layers_ptr = next(it) if fd._extruded else None
    subset_ptr = next(it) if fd._subset else None
    arg_ptrs = [next(it) for _ in fd.arguments]
    from .codegen import _distinct_maps
    mkas, mindex = _distinct_maps(fd)
    map_ptrs = [next(it) for _ in mkas]
    virtual = bool(fd._extruded or fd._subset)
    infos = [_device_maps.get(int(p)) for p in map_ptrs]
    if virtual and any(i is None for i in infos):
        raise ValueError("function-level seam: the maps of a loop over a subset / an extruded set must be registered "
                         "(bridge.register_map: the row count of the table is needed)")
    iter_total = max([i["nent"] for i in infos if i] + ([0] if virtual else [int(end)]))
    sizes = next((i["iterset"] for i in infos if i and i["iterset"]), None)
    if sizes is None:
        sizes = (iter_total,) * 3 if virtual else ((int(end), int(end), iter_total) if iter_total > end else (int(end),) * 3)
    base = op2.Set(tuple(sizes), "seam_iterset")
    iterset = base
    if fd._extruded:
        # the layers argument (set.py:351-353): [[bottom, top)] once, or one row per base entity
        lay = _host_ints(layers_ptr, 2) if fd._constant_layers else _host_ints(layers_ptr, 2 * base.total_size, (base.total_size, 2))
        if fd._constant_layers and int(lay[0]) != 0:
            raise NotImplementedError("function-level seam: constant layers start at 0 (set.py:342-345)")
        iterset = op2.ExtrudedSet(base, int(lay[1]) if fd._constant_layers else lay, extruded_periodic=fd._extruded_periodic)
    if fd._subset:
        iterset = op2.Subset(iterset, _host_ints(subset_ptr, int(end)))
    tosets, maps = {}, []
    for mka, ptr, info in zip(mkas, map_ptrs, infos):
        sizes = info["toset"] if info else None
        m = Map.__new__(Map)
        m._iterset, m._arity, m.name = (iterset.superset if fd._subset else iterset), mka.arity, f"seam_map_{int(ptr):x}"
        m._toset = tosets.setdefault(sizes, op2.Set(sizes if sizes and len(sizes) == 3 else (sizes[0] if sizes else 1), "seam_toset")) if sizes else None
        if virtual:
            # plans of loops over virtual spaces are built on derived maps: the host needs the rows (Parloop._plan_map)
            vals = info["values"] if info["values"] is not None else _host_ints(ptr, info["nent"] * mka.arity)
            m._values = np.asarray(vals, dtype=np.int32).reshape(info["nent"], mka.arity)
        else:
            m._values = _Shape((iter_total, mka.arity))
        m._offset, m._offset_quotient, m._plans = mka.offset, mka.offset_quotient, {}
        nrows = info["nent"] if info else iter_total
        m._dev = DeviceBuffer.wrap(int(ptr), nrows * mka.arity * 4, owned=False)
        if info and info["pb"] is not None:
            m.preferred_blocks = info["pb"]
        if info and info["pnb"] is not None:
            m.preferred_node_blocks = info["pnb"]
        maps.append(m)

    def map_of(mka):
        if mka is None:
            return None
        base = mka.base_map if isinstance(mka, K.PermutedMapKernelArg) else mka
        m = maps[mindex[id(base)]]
        return op2.PermutedMap(m, mka.permutation) if isinstance(mka, K.PermutedMapKernelArg) else m

    pargs = []
    for a, la, ptr in zip(fd.arguments, fd.local_kernel.arguments, arg_ptrs):
        if isinstance(a, K.DatKernelArg):
            m = map_of(a.map_)
            d = _BorrowedDat(int(ptr), a.dim, la.dtype, None if m is None else m._base()._toset)
            pargs.append(DatParloopArg(d, m))
        elif isinstance(a, K.GlobalKernelArg):
            pargs.append(GlobalParloopArg(_BorrowedDat(int(ptr), a.dim, la.dtype)))
        elif isinstance(a, K.MatKernelArg):
            dm = _device_mats.get(int(ptr))
            if dm is None:
                raise ValueError("a Mat slot must hold the handle of a bridge.DeviceMat")
            rm, cm = (map_of(x) for x in a.maps)
            for m in (rm, cm):
                if m._base()._toset is None:
                    raise ValueError("maps a Mat is assembled through must be registered (bridge.register_map): the owned row count is needed")
            key = (id(rm._base()), id(cm._base()))
            mat = dm._carriers.get(key)
            if mat is None:
                sp = Sparsity.__new__(Sparsity)
                sp._nested, sp._blocks, sp._built, sp._elem_tables = False, [[sp]], True, {}
                sp._dsets = (op2.DataSet(rm.toset, dm.rbs), op2.DataSet(cm.toset, dm.cbs))
                sp._pairs, sp._has_diagonal, sp.name = [], True, "seam_sparsity"
                sp._node_rowptr, sp._node_colidx, sp._node_nnz = dm.rowptr, dm.colidx, dm.nnz
                sp._rowptr, sp._colidx, sp._nnz = dm.scalar_pattern()
                mat = Mat.__new__(Mat)
                mat._sparsity, mat._dtype, mat.name, mat._vals = sp, np.dtype("float64"), "seam_mat", dm.values
                mat._zero_pending, mat.dat_version, mat._blocks = False, 0, [[mat]]
                mat._fd_device_mat = dm
                dm._carriers[key] = mat
            pargs.append(MatParloopArg(mat, (rm, cm), dm.lgmaps))
        else:
            raise NotImplementedError(f"function-level seam: {type(a).__name__}")
    return iterset, pargs


class _BorrowedDat:
    """A Dat/Global whose storage is somebody else's device memory: the wrapper only ever asks for the pointer.  ``toset`` =
    the Set the accessing Map points into when it is known (registered maps): its size is what the backend-derived locality
    order partitions."""

    class _NoHalo:
        halo = None
        size = total_size = None

    def __init__(self, ptr, dim, dtype, toset=None):
        self._ptr, self.dim, self.dtype = ptr, tuple(dim), np.dtype(dtype)
        self.cdim = int(np.prod(dim))
        # dat_version None: the owner (PyOP2 / PETSc / another seam func) may rewrite the buffer between calls without this
        # carrier hearing of it, so nothing may be cached on its VALUES (Parloop._plan_copy keeps the in-kernel gather)
        self._halo_frozen, self.halo_valid, self.dat_version = False, True, None
        s = toset if toset is not None else _BorrowedDat._NoHalo()
        self.dataset = type("DS", (), {"set": s, "size": getattr(s, "size", 0) or 0, "total_size": getattr(s, "total_size", 0) or 0,
                                       "cdim": self.cdim, "dim": self.dim})()

    def _dev_ptr(self, write):
        return self._ptr

    def global_to_local_begin(self, *_):      # the reference's Parloop performs the exchanges around func
        pass

    global_to_local_end = local_to_global_begin = local_to_global_end = global_to_local_begin


def compile_global_kernel_hip(kernel, comm=None):
    """Replacement for ``pyop2.global_kernel.compile_global_kernel`` (global_kernel.py:426-456): same cache role, same
    returned signature ``func(start, end, *arglist)``, and the SAME wrapper shapes the native Parloop uses -- staged
    (LDS gather / reduction) for Dat loops, owner-computes-rows for matrix loops, direct otherwise.  ``arglist`` holds
    device pointers in the reference's positional order; a Mat slot holds a ``DeviceMat.handle``."""
    from .codegen import select_mode
    from .parloop import Parloop
    fd = as_fd_global_kernel(kernel)
    seam = fd._seam_maps
    if fd.is_mixed:
        fd = fd.flattened()
    mode = select_mode(fd)
    nlead = (1 if fd._extruded else 0) + (1 if fd._subset else 0)
    from .codegen import _distinct_maps
    fd_maps = _distinct_maps(fd)[0]
    nhead = nlead + len(fd.arguments)
    # the reference passes one pointer per distinct LEAF map (parloop.py:203-212; a ComposedMap contributes its
    # constituents); this kernel's maps are those leaves or composed tables built from them
    leaves = seam["leaves"]
    leaf_pos = {id(l): i for i, l in enumerate(leaves)}
    nref = nhead + len(leaves)
    virtual = bool(fd._extruded or fd._subset)

    def own_arglist(arglist):
        """the reference's positional list -> this kernel's (composed maps resolved to their device tables)"""
        if not seam["composed"]:
            return list(arglist)
        ptrs = arglist[nhead:]
        out = list(arglist[:nhead])
        for mk in fd_maps:
            chain = seam["composed"].get(id(mk))
            if chain is None:
                out.append(ptrs[leaf_pos[id(seam["leaf_of"][id(mk)])]])
                continue
            cp = [ptrs[leaf_pos[id(b)]] for b in chain]
            info = _device_maps.get(int(cp[-1]))
            if info is None:
                raise ValueError("function-level seam: the innermost map of a ComposedMap must be registered (its row count is needed)")
            t = _composed_table(cp, mk.arity, info["nent"])
            if int(t.ptr) not in _device_maps:
                first = _device_maps.get(int(cp[0]))
                register_map(t.ptr, info["nent"], mk.arity, first["toset"] if first else (1,), info["iterset"])
            out.append(t.ptr)
        return out
    loops = {}
    has_mat = any(isinstance(a, K.MatKernelArg) for a in fd.arguments)
    # the descriptor does not say whether BC-masked lgmaps will be swapped in around a call (pyop2/parloop.py:279-314
    # does that on the PETSc Mat): the wrapper variant with lgmap parameters is built on first need
    variants = {False: fd}

    def variant(with_lgmaps):
        if with_lgmaps not in variants:
            import dataclasses
            args = [dataclasses.replace(a, lgmaps=True) if isinstance(a, K.MatKernelArg) else a for a in fd.arguments]
            variants[True] = K.GlobalKernel(fd.local_kernel, args, extruded=fd._extruded, extruded_periodic=fd._extruded_periodic,
                                            constant_layers=fd._constant_layers, subset=fd._subset,
                                            iteration_region=fd._iteration_region, pass_layer_arg=fd._pass_layer_arg)
        return variants[with_lgmaps]

    def func(start, end, *arglist):
        if len(arglist) != nref:
            raise ValueError(f"{fd.name}: expected {nref} arguments after (start, end), got {len(arglist)}")
        start, end = int(start), int(end)
        ref_arglist = arglist                        # cache keys use the reference's pointers (unregister_map finds them)
        arglist = own_arglist(arglist)
        if mode == "direct" and not has_mat and not virtual:
            # needs nothing but the reference's own list
            cw = fd.compile("direct")
            args = list(arglist) + [0] * (len(cw.src.layout) - len(arglist))
            threads, n = cw.src.block_threads, max(end - start, 0)
            cw.launch(start, end, args, block_threads=threads, ents_per_block=threads,
                      nblocks=max(1, min((n + threads - 1) // threads, 256 * 32)))
            return 0
        with_lg = False
        if has_mat:
            for a, ptr in zip(fd.arguments, arglist[nlead:]):
                if isinstance(a, K.MatKernelArg):
                    dm = _device_mats.get(int(ptr))
                    if dm is None:
                        raise ValueError("a Mat slot must hold the handle of a bridge.DeviceMat")
                    with_lg = with_lg or dm.lgmaps is not None
        key = (with_lg,) + tuple(int(a) if a is not None else 0 for a in ref_arglist)
        pl = loops.get(key)
        if pl is None:
            from .configuration import configuration
            gk = variant(with_lg)
            iterset, pargs = _borrowed_carriers(gk, arglist, start, end)
            old = configuration["type_check"]
            configuration["type_check"] = 0          # the carriers are pointer shells: nothing to check against
            try:
                pl = Parloop(gk, iterset, pargs)
            finally:
                configuration["type_check"] = old
            loops[key] = pl
        if end <= start:
            return 0
        for pa in pl.arguments:
            dm = getattr(getattr(pa, "data", None), "_fd_device_mat", None)
            if dm is not None:
                pa.lgmaps = dm.lgmaps                # the lgmap swap of pyop2/parloop.py:279-314 happens per call
                if dm._zero_requested:               # Mat.zero() since the last assembly: consumed by this launch
                    pa.data._zero_pending = True
                    dm._zero_requested = False
        if pl._prepare()["cw"].src.mode.startswith("ocr"):
            pl._compute_ocr(start, end)
        else:
            pl._compute((start, end - start))
        return 0
    func.global_kernel = fd
    func.mode = mode
    func.loops = loops
    _seam_funcs.append(func)
    return func
