"""The PyOP2 side of the seam: turn the reference's own kernel descriptors into this
backend's and hand back a callable with the reference's ``func(start, end, *arglist)`` signature.

PyOP2 cannot be imported here (``loopy``, ``petsc4py`` and ``mpi4py`` are absent), so nothing in this module imports
it: the descriptors are recognised by *class name and attributes* -- exactly the fields ``pyop2/global_kernel.py:27-325``
and ``pyop2/local_kernel.py:86-227`` define -- which is also what makes the translation testable with stand-in objects
(tests/test_bridge.py).  Two levels of integration exist:

* **function level** (this module): ``compile_global_kernel_hip(kernel, comm)`` replaces
  ``pyop2.global_kernel.compile_global_kernel`` (global_kernel.py:426-456).  The returned callable takes the reference's
  positional list -- ``start, end, [layers], [subset], one pointer per Dat/Global, one per distinct Map`` -- with DEVICE
  pointers (the carriers' ``_kernel_args_`` patched as INTEGRATION.md shows) and launches the *direct* wrapper, which
  needs nothing but that list.  Matrix arguments are PETSc ``Mat`` handles in the reference's list and have no meaning
  on the device: loops with Mat arguments, and the staged / owner-computes-rows fast paths (which need the
  backend-private plan tables), go through
* **Parloop level**: ``firedrake_amd.parloop.Parloop`` is the worked replacement of ``pyop2.parloop.Parloop`` (same
  protocol, ``_arglist`` appends the private tables); ``as_fd_global_kernel`` is what it needs from PyOP2's side.
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
        raise NotImplementedError("ComposedMapKernelArg: compose the maps on the host (op2.ComposedMap does) and pass "
                                  "the composed table as an ordinary map")
    else:
        raise TypeError(f"unknown map kernel argument {n}")
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
    if isinstance(lk.code, str):
        return K.CStringLocalKernel(lk.code, lk.name, accesses, dtypes, **kw)
    return K.LoopyLocalKernel(lk.code, lk.name, accesses, dtypes, **kw)      # lowered with lp.generate_code_v2


def as_fd_global_kernel(gk):
    """pyop2 ``GlobalKernel`` (global_kernel.py:255-325) -> ``firedrake_amd.kernel.GlobalKernel`` with the same local
    kernel, argument descriptors (map identity preserved) and iteration flags."""
    memo = {}
    args = [_kernel_arg(a, memo) for a in gk.arguments]
    region = getattr(gk, "_iteration_region", None)
    return K.GlobalKernel(as_fd_local_kernel(gk.local_kernel), args,
                          extruded=bool(getattr(gk, "_extruded", False)),
                          extruded_periodic=bool(getattr(gk, "_extruded_periodic", False)),
                          constant_layers=bool(getattr(gk, "_constant_layers", False)),
                          subset=bool(getattr(gk, "_subset", False)),
                          iteration_region=None if region is None else IterationRegion(int(region)),
                          pass_layer_arg=bool(getattr(gk, "_pass_layer_arg", False)))


def compile_global_kernel_hip(kernel, comm=None):
    """Replacement for ``pyop2.global_kernel.compile_global_kernel``: same cache role, same returned signature
    ``func(start, end, *arglist)``.  See the module docstring for what the function-level seam covers."""
    fd = as_fd_global_kernel(kernel)
    if fd.is_mixed:
        fd = fd.flattened()
    if any(isinstance(a, K.MatKernelArg) for a in fd.arguments):
        raise NotImplementedError("loops with Mat arguments need the Parloop-level integration "
                                  "(firedrake_amd.parloop.Parloop): a PETSc Mat handle means nothing on the device")
    cw = fd.compile("direct")
    nref = sum(1 for d in cw.src.layout if d[0] in ("layers", "subset", "arg", "map"))
    threads = cw.src.block_threads

    def func(start, end, *arglist):
        if len(arglist) != nref:
            raise ValueError(f"{fd.name}: expected {nref} arguments after (start, end), got {len(arglist)}")
        args = list(arglist) + [0] * (len(cw.src.layout) - nref)           # backend-private slots unused by `direct`
        n = max(int(end) - int(start), 0)
        cw.launch(start, end, args, block_threads=threads, ents_per_block=threads,
                  nblocks=max(1, min((n + threads - 1) // threads, 256 * 32)))
    func.wrapper = cw
    return func
