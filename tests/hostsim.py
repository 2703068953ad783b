"""TEST INFRASTRUCTURE ONLY: run a DIRECT-mode wrapper emitted by firedrake_amd/codegen.py on the host.

The generated HIP source is compiled by g++ against tests/hostsim/fd_wrapper.h (a sequential one-lane
stand-in for csrc/fd_wrapper.h) and called with host pointers in the kernel's own parameter order
(``WrapperSource.layout``).  This checks the code generator's indexing logic -- maps, extruded offsets,
layer bounds, subsets, lgmap masking, CSR search -- against the oracle where no GPU is available.  It is
not a fallback: nothing under firedrake_amd/ can reach it, and the product path still raises without a GPU.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

from helpers import oracle_pattern
from firedrake_amd.codegen import generate_wrapper
from firedrake_amd.configuration import configuration
from firedrake_amd.parloop import DatParloopArg, GlobalParloopArg, MatParloopArg

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "hostsim", "_build")


def _compile(source, name):
    os.makedirs(_BUILD, exist_ok=True)
    key = hashlib.sha1(source.encode()).hexdigest()[:16]
    so = os.path.join(_BUILD, f"{name}_{key}.so")
    if not os.path.exists(so):
        src = os.path.join(_BUILD, f"{name}_{key}.cpp")
        with open(src, "w") as f:
            f.write(source)
        cmd = ["g++", "-O1", "-fPIC", "-shared", "-std=gnu++17", "-w", "-I", os.path.join(_HERE, "hostsim"),
               "-o", so + ".tmp", src, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hostsim compile failed:\n" + r.stderr[-4000:])
        os.replace(so + ".tmp", so)
    return ctypes.CDLL(so)


def run_direct(pl, part=None):
    """Execute Parloop ``pl`` over ``part`` = (offset, size) (default: all owned entities) with the direct
    wrapper on the host.  Returns one array per argument: COPIES of Dat/Global data after the loop, or an
    OracleCSR holding the assembled values for a Mat.  The carriers are not modified."""
    old = configuration["mat_scatter"]
    configuration["mat_scatter"] = "search"          # the element->nz table is built on the device
    try:
        src = generate_wrapper(pl.global_kernel, "direct")
    finally:
        configuration["mat_scatter"] = old
    lib = _compile(src.source, src.symbol)
    fn = getattr(lib, src.symbol)
    fn.restype = None
    offset, size = part if part is not None else (0, pl.iterset.size)
    maps = []
    for pa in pl.arguments:
        for m in getattr(pa, "maps", ()):
            if all(m._base() is not q for q in maps):
                maps.append(m._base())
    outs, keep = {}, []

    def ptr(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return ctypes.c_void_p(a.ctypes.data)

    cargs = [ctypes.c_int(offset), ctypes.c_int(offset + size)]
    for desc in src.layout:
        kind = desc[0]
        if kind == "layers":
            cargs.append(ptr(np.asarray(pl.iterset.layers_array, dtype=np.int32)))
        elif kind == "subset":
            cargs.append(ptr(np.asarray(pl.iterset.indices, dtype=np.int32)))
        elif kind == "arg":
            pa = pl.arguments[desc[1]]
            if isinstance(pa, MatParloopArg):
                csr = oracle_pattern(pa.data.sparsity)
                outs[desc[1]] = csr
                cargs.append(ctypes.c_void_p(csr.values.ctypes.data))
            elif isinstance(pa, (DatParloopArg, GlobalParloopArg)):
                host = pa.data._host if pa.data._host_valid else pa.data._to_host()     # a DatView: its parent's rows
                a = np.array(host, copy=True)
                outs[desc[1]] = a
                cargs.append(ctypes.c_void_p(a.ctypes.data))
            else:
                cargs.append(ctypes.c_void_p(0))
        elif kind == "map":
            cargs.append(ptr(np.asarray(maps[desc[1]].values_with_halo, dtype=np.int32)))
        elif kind == "bstart":
            cargs.append(ctypes.c_void_p(0))
        elif kind == "mat_rowptr":
            cargs.append(ctypes.c_void_p(outs[desc[1]].rowptr.ctypes.data))
        elif kind == "mat_colidx":
            cargs.append(ctypes.c_void_p(outs[desc[1]].colidx.ctypes.data))
        elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
            pa = pl.arguments[desc[1]]
            cargs.append(ptr(np.asarray(pa.lgmaps[0 if kind == "mat_row_lgmap" else 1], dtype=np.int32)))
        else:
            raise AssertionError(f"hostsim cannot provide {kind}")
    fn(*cargs)
    return [outs.get(k) for k in range(len(pl.arguments))]
