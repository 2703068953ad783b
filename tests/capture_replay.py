"""Replay of a ``tools/firedrake_capture.py`` capture (``tests/golden/firedrake_<cfg>.npz``) through this repository's PyOP2-level API.

File format (version 1).  An ``.npz`` without pickles: arrays + ``meta``, a JSON string:

    meta = {"format": 1, "config": "c1", "source": "firedrake" | "synthetic", "tolerances": {"vector": 1e-12, "matrix": 1e-12},
            "parloops": [loop, ...],                      # in execution order, each tagged with the form it belongs to ("F" | "J")
            "outputs": {"F": {"data": key}, "J": {"indptr": key, "indices": key, "data": key}}}
    loop = {"kernel_name", "kernel_c", "headers", "requires_zeroed_output_arguments", "form",
            "iterset": {"sizes": [core, owned, total], ["subset_indices": key, "superset": {...}], ["layers_array": key, ...]},
            "iteration_region", "pass_layer_arg", "subset", "extruded",
            "args": [{"kind": "dat", "access", "dtype", "dim", "map": map | null, "index", "dataset_sizes", "before": key, ["after": key]},
                     {"kind": "global", "access", "dtype", "dim", "before": key, ["after": key]},
                     {"kind": "mat", "access", "dtype", "dims", "maps": [map, map], "unroll", "lgmaps": [key, key] | null,
                      "row_sizes", "col_sizes"}]}
    map  = {"type": "map", "values": key, "arity", "iterset_sizes", "toset_sizes", "offset", "offset_quotient"}
         | {"type": "permuted", "permutation", "map": map} | {"type": "composed", "maps": [map, ...]}

``key`` names an array of the file.  The replay rebuilds Sets / Maps / Dats / Globals / Mats (one object per distinct key, so carriers
shared between loops stay shared), wraps the kernel text with ``op2.Kernel`` and hands every loop to a ``runner`` -- the oracle or the
device -- then compares written carriers with their captured ``after`` arrays and the assembled tensors with ``outputs``."""
import json

import numpy as np

from firedrake_amd import op2
from firedrake_amd.kernel import loopy_c_kernel

ACCESS = {"READ": op2.READ, "WRITE": op2.WRITE, "RW": op2.RW, "INC": op2.INC, "MIN": op2.MIN, "MAX": op2.MAX}


def load(path):
    z = np.load(path, allow_pickle=False)
    meta = json.loads(str(z["meta"]))
    if meta.get("format") != 1:
        raise ValueError(f"{path}: unknown capture format {meta.get('format')!r}")
    return meta, z


class Replay:
    """Objects of one capture, built lazily and shared by key."""

    def __init__(self, meta, arrays):
        self.meta, self.z = meta, arrays
        self.sets, self.maps, self.dats = {}, {}, {}

    def set_of(self, sizes):
        sizes = tuple(int(s) for s in sizes)
        if sizes not in self.sets:                            # (sets are identified by their sizes: enough for single-mesh captures)
            self.sets[sizes] = op2.Set(sizes if len(set(sizes)) > 1 else sizes[0])
        return self.sets[sizes]

    def iterset(self, d):
        base = self.set_of(d["superset"]["sizes"] if "subset_indices" in d else d["sizes"])
        src = d.get("superset", d)
        if "layers_array" in src:
            la = np.asarray(self.z[src["layers_array"]])
            key = ("ext", src["layers_array"])
            if key not in self.sets:
                layers = int(la[0, 1]) if src.get("constant_layers", True) else la
                self.sets[key] = op2.ExtrudedSet(base, layers, extruded_periodic=bool(src.get("extruded_periodic", False)))
            base = self.sets[key]
        if "subset_indices" in d:
            return op2.Subset(base, np.asarray(self.z[d["subset_indices"]]))
        return base

    def map_of(self, d, iterset):
        if d is None:
            return None
        if d["type"] == "permuted":
            return op2.PermutedMap(self.map_of(d["map"], iterset), d["permutation"])
        if d["type"] == "composed":
            return op2.ComposedMap(*[self.map_of(q, None) for q in d["maps"]])
        key = d["values"]
        if key not in self.maps:
            its = iterset if (iterset is not None and tuple(iterset.sizes) == tuple(d["iterset_sizes"])) else self.set_of(d["iterset_sizes"])
            base_it = getattr(its, "superset", its)
            self.maps[key] = op2.Map(base_it, self.set_of(d["toset_sizes"]), int(d["arity"]), np.asarray(self.z[key], dtype=np.int32),
                                     offset=d.get("offset"), offset_quotient=d.get("offset_quotient"))
        return self.maps[key]

    def run(self, runner, form=None, check=None):
        """Replay the loops of ``form`` (all when None) in order through ``runner(kernel, iterset, args, loop)`` -> list of post-loop
        arrays per argument (None for READ ones; an object with ``rowptr / colidx / values`` for Mats).  Written Dats / Globals are
        compared with the captured ``after`` arrays by ``check(got, expected, what)``; returns {form: assembled Mat result or None}."""
        mats = {}
        for loop in self.meta["parloops"]:
            if form is not None and loop["form"] != form:
                continue
            its = self.iterset(loop["iterset"])
            accesses = [ACCESS[a["access"]] for a in loop["args"]]
            dtypes = [np.dtype(a["dtype"]) for a in loop["args"]]
            mk = loopy_c_kernel if loop.get("requires_zeroed_output_arguments", True) else op2.Kernel
            kernel = mk(loop["kernel_c"], loop["kernel_name"], accesses=accesses, dtypes=dtypes, headers=tuple(loop.get("headers", ())))
            args = []
            for a, acc in zip(loop["args"], accesses):
                if a["kind"] == "dat":
                    ds = self.set_of(a["dataset_sizes"])
                    dim = tuple(a["dim"])
                    d = op2.Dat(ds ** (dim if len(dim) > 1 else dim[0]), np.array(self.z[a["before"]], copy=True), dtype=np.dtype(a["dtype"]))
                    m = self.map_of(a["map"], its)
                    if a.get("index") is not None:
                        d = op2.DatView(d, tuple(a["index"]))               # a kernel that sees one component (dat.py:714-805)
                    args.append(d(acc, m) if m is not None else d(acc))
                elif a["kind"] == "global":
                    g = op2.Global(tuple(a["dim"]), np.array(self.z[a["before"]], copy=True), dtype=np.dtype(a["dtype"]))
                    args.append(g(acc))
                else:
                    rm, cm = (self.map_of(q, its) for q in a["maps"])
                    rdim, cdim = a["dims"]
                    rs, cs = self.set_of(a["row_sizes"]), self.set_of(a["col_sizes"])
                    sp = op2.Sparsity((rs ** (rdim[0] if len(rdim) == 1 else tuple(rdim)), cs ** (cdim[0] if len(cdim) == 1 else tuple(cdim))),
                                      [(rm, cm, None)])
                    mat = op2.Mat(sp)
                    lg = None if a.get("lgmaps") is None else tuple(np.asarray(self.z[k], dtype=np.int32) for k in a["lgmaps"])
                    kw = {"lgmaps": lg}
                    if a.get("unroll"):
                        kw["unroll_map"] = True
                    args.append(mat(acc, (rm, cm), **kw))
            outs = runner(kernel, its, args, loop)
            for a, out in zip(loop["args"], outs):
                if a["kind"] == "mat":
                    mats[loop["form"]] = out
                elif "after" in a and check is not None:
                    check(np.asarray(out).reshape(-1), np.asarray(self.z[a["after"]]).reshape(-1), f"{loop['kernel_name']}: {a['kind']} after the loop")
        return mats


def csr_rows_equal(got_rowptr, got_colidx, got_values, ref_indptr, ref_indices, ref_data, rows=None):
    """max |difference| between two CSR matrices that may hold different (super)patterns: compared entry by entry over the union
    pattern of the selected rows (PETSc keeps explicit zeros of the preallocation; a BC row holds its diagonal only)."""
    worst = 0.0
    nrows = min(len(got_rowptr), len(ref_indptr)) - 1
    for r in (range(nrows) if rows is None else rows):
        a = dict(zip(got_colidx[got_rowptr[r]:got_rowptr[r + 1]].tolist(), got_values[got_rowptr[r]:got_rowptr[r + 1]].tolist()))
        b = dict(zip(ref_indices[ref_indptr[r]:ref_indptr[r + 1]].tolist(), ref_data[ref_indptr[r]:ref_indptr[r + 1]].tolist()))
        for c in set(a) | set(b):
            worst = max(worst, abs(a.get(c, 0.0) - b.get(c, 0.0)))
    return worst
