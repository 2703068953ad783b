#!/usr/bin/env python
"""tools/firedrake_capture.py [--configs c1,c2,c3,c4,c5] [--full] [--out tests/golden] -- runs ONLY where ``import firedrake`` works.

The day an environment has Firedrake, this one command closes the three things this repository cannot close on its own
("missing" 1-2): TSFC-generated kernel text has never met the HIP wrappers, the oracle is pinned to the reference's
PyOP2-level goldens but not to ``assemble()`` itself, and the function-level seam has never been bound to a real PyOP2.

What it does.  For each of the five BASELINE.json configs (reduced sizes unless --full) it builds the form in Firedrake and hooks the
ONE boundary this backend replaces -- ``pyop2.parloop.Parloop.__call__`` (pyop2/parloop.py:243-260), whose ``_compute`` ends in
``func(start, end, *arglist)`` (pyop2/global_kernel.py:327-335).  Every parloop that ``assemble(F)`` / ``assemble(J)`` executes is
recorded AT that boundary:

  * the local kernel as C text -- ``lp.generate_code_v2(local_kernel.code).device_code()`` for TSFC's loopy kernels
    (pyop2/local_kernel.py:210-227), the string itself for C-string kernels -- with name, access descriptors and dtypes;
  * the argument descriptors of the GlobalKernel (Dat dims, map arities / offsets, Mat dims / unroll, iteration region, subset,
    extrusion: pyop2/global_kernel.py:28-330);
  * the data that crosses: ``Map.values_with_halo``, every Dat / Global BEFORE the loop, every written Dat / Global AFTER it, the
    Mat's lgmaps (parloop.py:279-302), set sizes (core, owned, total);
  * and, per form, what ``assemble()`` returned: ``r.dat.data_ro`` and the PETSc matrix as CSR (``getValuesCSR``).

Each config becomes ``<out>/firedrake_<cfg>.npz`` (arrays + one JSON string ``meta``; no pickles) in the format documented in
tests/capture_replay.py, which replays it: ``tests/test_firedrake_capture.py`` runs every captured loop through the oracle (CPU) and
through the HIP wrappers (GPU) and compares with the captured outputs at the SURVEY.md Appendix D tolerances.  A synthetic capture in
the same format (tests/golden/make_synthetic_capture.py, written from this repository's own kernels) keeps that test exercised until
a real one exists.

Nothing here is imported by the product or by the tests; the file only talks to Firedrake's public API and to the PyOP2 objects a
Parloop holds."""
import argparse
import json
import os
import sys

import numpy as np

FORMAT = 1


# ------------------------------------------------------------------------------------------------------------------------------------
# recording at the Parloop boundary
# ------------------------------------------------------------------------------------------------------------------------------------
class Recorder:
    """Collects arrays (deduplicated by object identity of their carrier) and one record per executed parloop."""

    def __init__(self):
        self.arrays, self.ids, self.loops = {}, {}, []

    def put(self, prefix, carrier, array):
        key = self.ids.get((prefix, id(carrier)))
        if key is None:
            key = f"{prefix}{len([k for k in self.ids if k[0] == prefix])}"
            self.ids[(prefix, id(carrier))] = key
            self.arrays[key] = np.ascontiguousarray(array)
        return key

    def snapshot(self, tag, array):
        key = f"{tag}_{len(self.arrays)}"
        self.arrays[key] = np.array(array, copy=True)
        return key


def kernel_text(local_kernel):
    """C text of a PyOP2 local kernel (pyop2/local_kernel.py:182-227)."""
    code = local_kernel.code
    if isinstance(code, str):
        return code
    import loopy as lp
    return lp.generate_code_v2(code).device_code()


def describe_map(rec, m):
    """Map / PermutedMap / ComposedMap -> JSON (values stored once per Map object)."""
    from pyop2.types import ComposedMap, PermutedMap
    if m is None:
        return None
    if isinstance(m, PermutedMap):
        return {"type": "permuted", "permutation": [int(p) for p in m.permutation], "map": describe_map(rec, m.map_)}
    if isinstance(m, ComposedMap):
        return {"type": "composed", "maps": [describe_map(rec, q) for q in m.maps_]}
    return {"type": "map", "values": rec.put("map", m, np.asarray(m.values_with_halo, dtype=np.int32)), "arity": int(m.arity),
            "iterset_sizes": [int(s) for s in m.iterset.sizes], "toset_sizes": [int(s) for s in m.toset.sizes],
            "offset": None if m.offset is None else [int(o) for o in m.offset],
            "offset_quotient": None if getattr(m, "offset_quotient", None) is None else [int(o) for o in m.offset_quotient]}


def describe_set(rec, s):
    from pyop2.types import ExtrudedSet, Subset
    d = {"sizes": [int(v) for v in s.sizes], "name": getattr(s, "name", None)}
    if isinstance(s, Subset):
        d["subset_indices"] = rec.snapshot("subset", np.asarray(s.indices, dtype=np.int32))
        d["superset"] = describe_set(rec, s.superset)
    base = s.superset if isinstance(s, Subset) else s
    if isinstance(base, ExtrudedSet):
        d["layers_array"] = rec.snapshot("layers", np.asarray(base.layers_array, dtype=np.int32))
        d["constant_layers"] = bool(base.constant_layers)
        d["extruded_periodic"] = bool(getattr(base, "_extruded_periodic", False))
    return d


def record_parloop(rec, pl, post):
    """One record: descriptors + data before the loop; ``post`` fills in the written carriers afterwards."""
    from pyop2 import op2
    from pyop2.parloop import DatParloopArg, GlobalParloopArg, MatParloopArg
    gk = pl.global_kernel
    lk = gk.local_kernel
    names = {op2.READ: "READ", op2.WRITE: "WRITE", op2.RW: "RW", op2.INC: "INC", op2.MIN: "MIN", op2.MAX: "MAX"}
    args = []
    for pa, ka, acc, dt in zip(pl.arguments, gk.arguments, lk.accesses, lk.dtypes):
        a = {"access": names[acc], "dtype": np.dtype(dt).name}
        if isinstance(pa, DatParloopArg):
            d = pa.data
            a.update(kind="dat", dim=[int(v) for v in d.dim], map=describe_map(rec, pa.map_),
                     index=None if getattr(ka, "index", None) is None else [int(v) for v in ka.index],
                     dataset_sizes=[int(s) for s in d.dataset.set.sizes],
                     before=rec.snapshot("dat_before", d.data_ro_with_halos))
            if acc != op2.READ:
                post.append((a, "after", lambda d=d: d.data_ro_with_halos))
        elif isinstance(pa, GlobalParloopArg):
            g = pa.data
            a.update(kind="global", dim=[int(v) for v in g.dim], before=rec.snapshot("glob_before", g.data_ro))
            if acc != op2.READ:
                post.append((a, "after", lambda g=g: g.data_ro))
        elif isinstance(pa, MatParloopArg):
            sp = pa.data.sparsity
            lg = None
            if pa.lgmaps is not None:
                # (a list of (row lgmap, column lgmap) per block: parloop.py:279-302; the non-mixed case has one pair)
                (rl, cl), = pa.lgmaps
                lg = [rec.snapshot("lgmap", np.asarray(rl.indices, dtype=np.int32)), rec.snapshot("lgmap", np.asarray(cl.indices, dtype=np.int32))]
            a.update(kind="mat", dims=[[int(v) for v in dm] for dm in sp.dims[0][0]], maps=[describe_map(rec, m) for m in pa.maps],
                     unroll=bool(getattr(ka, "unroll", False)), lgmaps=lg,
                     row_sizes=[int(s) for s in sp.dsets[0].set.sizes], col_sizes=[int(s) for s in sp.dsets[1].set.sizes])
        else:
            raise NotImplementedError(f"capture of {type(pa).__name__} (mixed spaces are outside the five configs)")
        args.append(a)
    return {"kernel_name": lk.name, "kernel_c": kernel_text(lk), "headers": list(getattr(lk, "headers", ()) or ()),
            "requires_zeroed_output_arguments": bool(getattr(lk, "requires_zeroed_output_arguments", False)),
            "iterset": describe_set(rec, pl.iterset), "iteration_region": str(getattr(gk, "_iteration_region", None)),
            "pass_layer_arg": bool(getattr(gk, "_pass_layer_arg", False)), "subset": bool(getattr(gk, "_subset", False)),
            "extruded": bool(getattr(gk, "_extruded", False)), "args": args}


class capture_parloops:
    """Context manager: every ``Parloop.__call__`` inside it is recorded into ``rec``."""

    def __init__(self, rec, tag):
        self.rec, self.tag = rec, tag

    def __enter__(self):
        from pyop2 import parloop as pyop2_parloop
        self.cls = pyop2_parloop.Parloop
        self.orig = orig = self.cls.__call__
        rec, tag = self.rec, self.tag

        def call(pl):
            post = []
            entry = record_parloop(rec, pl, post)
            entry["form"] = tag
            orig(pl)
            for a, key, get in post:
                a[key] = rec.snapshot("dat_after", get())
            rec.loops.append(entry)
        self.cls.__call__ = call
        return self

    def __exit__(self, *exc):
        self.cls.__call__ = self.orig
        return False


# ------------------------------------------------------------------------------------------------------------------------------------
# the five configs (BASELINE.json "configs"), reduced unless --full
# ------------------------------------------------------------------------------------------------------------------------------------
def config_forms(cfg, full):
    import firedrake as fd
    if cfg in ("c1", "c2", "c5"):
        if cfg == "c1":
            mesh, degree = fd.UnitSquareMesh(64, 64), 1
        elif cfg == "c2":
            n = 215 if full else 12
            mesh, degree = fd.UnitCubeMesh(n, n, n), 1
        else:
            n = 215 if full else 8
            mesh, degree = fd.UnitCubeMesh(n, n, n), 2
        V = fd.FunctionSpace(mesh, "CG", degree)
        u, v = fd.Function(V), fd.TestFunction(V)
        xs = fd.SpatialCoordinate(mesh)
        u.interpolate(fd.sin(3 * xs[0]) * fd.cos(2 * xs[1]) + (0.3 * xs[2] if mesh.geometric_dimension() == 3 else 0))
        f = fd.Function(V).interpolate((1 + 8 * fd.pi ** 2) * fd.cos(2 * fd.pi * xs[0]) * fd.cos(2 * fd.pi * xs[1]))
        F = fd.inner(fd.grad(u), fd.grad(v)) * fd.dx - f * v * fd.dx
        J = fd.derivative(F, u)
        bcs = [fd.DirichletBC(V, 0, "on_boundary")]
        return {"F": (F, bcs), "J": (J, bcs)}, mesh, V
    if cfg == "c3":
        n = 32 if full else 3
        base = fd.UnitSquareMesh(n, n, quadrilateral=True)
        mesh = fd.ExtrudedMesh(base, n)
        V = fd.FunctionSpace(mesh, "Q", 4)
        u, v = fd.TrialFunction(V), fd.TestFunction(V)
        a = (fd.inner(fd.grad(u), fd.grad(v)) + fd.inner(u, v)) * fd.dx
        w = fd.Function(V).interpolate(fd.SpatialCoordinate(mesh)[0])
        return {"J": (a, []), "F": (fd.action(a, w), [])}, mesh, V
    if cfg == "c4":
        # demos/DG_advection: DQ1 upwind; the right-hand side L1 assembled matrix-free
        n = 2048 if full else 16
        mesh = fd.UnitSquareMesh(n, n, quadrilateral=True)
        V = fd.FunctionSpace(mesh, "DQ", 1)
        W = fd.VectorFunctionSpace(mesh, "CG", 1)
        x, y = fd.SpatialCoordinate(mesh)
        vel = fd.Function(W).interpolate(fd.as_vector((0.5 - y, x - 0.5)))
        q = fd.Function(V).interpolate(fd.exp(-((x - 0.5) ** 2 + (y - 0.75) ** 2) / 0.01))
        q_in = fd.Constant(1.0)
        phi = fd.TestFunction(V)
        dtc = fd.Constant(1e-3)
        nrm = fd.FacetNormal(mesh)
        un = 0.5 * (fd.dot(vel, nrm) + abs(fd.dot(vel, nrm)))
        L1 = dtc * (q * fd.div(phi * vel) * fd.dx - fd.conditional(fd.dot(vel, nrm) < 0, phi * fd.dot(vel, nrm) * q_in, 0.0) * fd.ds
                    - fd.conditional(fd.dot(vel, nrm) > 0, phi * fd.dot(vel, nrm) * q, 0.0) * fd.ds
                    - (phi('+') - phi('-')) * (un('+') * q('+') - un('-') * q('-')) * fd.dS)
        return {"F": (L1, [])}, mesh, V
    raise ValueError(cfg)


def capture(cfg, full, outdir):
    import firedrake as fd
    forms, mesh, V = config_forms(cfg, full)
    rec = Recorder()
    outputs = {}
    for tag, (form, bcs) in forms.items():
        fd.assemble(form, bcs=bcs)                         # warm-up: JIT outside the recording
        with capture_parloops(rec, tag):
            t = fd.assemble(form, bcs=bcs)
        if tag == "J":
            indptr, indices, data = t.petscmat.getValuesCSR()
            outputs[tag] = {"indptr": rec.snapshot("J_indptr", indptr), "indices": rec.snapshot("J_indices", indices),
                            "data": rec.snapshot("J_data", data)}
        else:
            outputs[tag] = {"data": rec.snapshot("F_data", t.dat.data_ro)}
    meta = {"format": FORMAT, "config": cfg, "full_size": bool(full), "source": "firedrake", "firedrake_version": getattr(fd, "__version__", "unknown"),
            "scalar_type": str(fd.utils.ScalarType), "int_type": str(fd.utils.IntType), "dofs": int(V.dim()),
            "tolerances": {"vector": 1e-12, "matrix": 1e-12 if cfg != "c3" else 1e-11},
            "parloops": rec.loops, "outputs": outputs}
    path = os.path.join(outdir, f"firedrake_{cfg}.npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **rec.arrays)
    print(f"{cfg}: {len(rec.loops)} parloops, {len(rec.arrays)} arrays -> {path}")


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--configs", default="c1,c2,c3,c4,c5")
    ap.add_argument("--full", action="store_true", help="the sizes BASELINE.json names (GBs per file) instead of the reduced parity sizes")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
    args = ap.parse_args()
    try:
        import firedrake  # noqa: F401
    except ImportError as exc:
        sys.exit(f"firedrake_capture.py needs an environment with Firedrake ({exc}); see the module docstring")
    os.makedirs(args.out, exist_ok=True)
    for cfg in args.configs.split(","):
        capture(cfg.strip(), args.full, args.out)


if __name__ == "__main__":
    main()
