"""The reference's own local-kernel TEXT (tests/golden/reference_kernels.json, lifted unchanged from its PyOP2 tests by
tests/golden/make_reference_kernels.py) as ``op2.Kernel`` objects, plus two executors with one interface so that every
re-enacted reference test runs twice:

* ``GpuBackend``  -- ``op2.par_loop``: the HIP wrappers through the C ABI (tests marked ``gpu``);
* ``HostBackend`` -- the wrapper text ``codegen.py`` generates around the SAME kernel text, compiled with g++ against the
  stand-in header and run on the CPU (tests/hostsim.py, direct mode), results written back into the carriers; mixed arguments
  (which the parloop splits before any wrapper is generated) go through the oracle instead.  Says nothing about the HIP
  build -- it shows on a machine without a GPU that reference-authored C compiles unchanged inside the wrapper.
"""
import json
import os

import numpy as np

from firedrake_amd import op2
from firedrake_amd.parloop import (DatParloopArg, GlobalParloopArg, MatParloopArg, MixedDatLegacyArg,
                                   MixedMatLegacyArg)

_REF = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "reference_kernels.json")))["kernels"]


def ref_kernel(key, **kw):
    """``op2.Kernel`` around the reference's text for ``key`` = "<test file>::<fixture or Class.test>::<kernel name>"."""
    ent = _REF[key]
    return op2.Kernel(ent["code"], ent["name"], **kw)


def ref_source(key):
    return _REF[key]["source"]


class GpuBackend:
    name = "gpu"

    def par_loop(self, kernel, iterset, *args, **kw):
        op2.par_loop(kernel, iterset, *args, **kw)

    def values(self, mat):
        return np.array(mat.values)

    def zero(self, mat):
        mat.zero()

    def zero_rows(self, mat, rows, diag):
        mat.zero_rows(rows, diag)


class HostBackend:
    name = "host"

    def __init__(self):
        self._dense = {}

    def _get(self, mat):
        return self._dense.setdefault(id(mat), np.zeros((mat.nrows, mat.ncols)))

    def par_loop(self, kernel, iterset, *args, **kw):
        from helpers import oracle_run
        if any(isinstance(a, (MixedDatLegacyArg, MixedMatLegacyArg)) for a in args):
            outs = oracle_run(kernel, iterset, *args, **kw)
            for a, out in zip(args, outs):
                if isinstance(a, MixedDatLegacyArg) and a.access != op2.READ:
                    for d, o in zip(a.data.split, out):
                        d._host_rw()[...] = np.asarray(o).reshape(d._host.shape)
                elif isinstance(a, MixedMatLegacyArg):
                    for i, row in enumerate(out):
                        for j, csr in enumerate(row):
                            blk = a.data[i, j]
                            self._dense[id(blk)] = self._get(blk) + csr.todense()
            return
        from hostsim import run_direct
        pl = op2.LegacyParloop(kernel, iterset, *args, **kw)
        outs = run_direct(pl)
        for pa, acc, out in zip(pl.arguments, pl.accesses, outs):
            if isinstance(pa, (DatParloopArg, GlobalParloopArg)) and acc != op2.READ:
                pa.data._host_rw()[...] = np.asarray(out).reshape(pa.data._host.shape)
            elif isinstance(pa, MatParloopArg):
                # the host-sim assembles into a zeroed copy of the pattern: INC adds to what the Mat held, WRITE replaces the
                # entries the loop touched (the reference tests that use WRITE touch every entry of the pattern)
                d = out.todense()
                self._dense[id(pa.data)] = d if acc == op2.WRITE else self._get(pa.data) + d

    def values(self, mat):
        return np.array(self._get(mat))

    def zero(self, mat):
        self._dense[id(mat)] = np.zeros((mat.nrows, mat.ncols))

    def zero_rows(self, mat, rows, diag):
        d = self._get(mat)
        rows = np.asarray(getattr(rows, "indices", rows), dtype=np.int64)
        d[rows, :] = 0.0
        d[rows, rows] = diag
