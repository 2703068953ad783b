#!/usr/bin/env python
"""Writes tests/golden/firedrake_synthetic_c1.npz: a capture in the format of tools/firedrake_capture.py (tests/capture_replay.py
documents it) produced WITHOUT Firedrake -- Poisson CG1 on a 12 x 12 triangulated square with Dirichlet conditions, the kernels
this repository restates for config C1, the loop outputs computed by the oracle -- so that the replay path (kernel text -> wrappers ->
comparison with captured arrays) is exercised until a real capture exists.  ``"source": "synthetic"`` marks it: it pins nothing to
the reference; it only keeps the one-command path of the day Firedrake is available from rotting.

    python tests/golden/make_synthetic_capture.py          # rewrites the file; deterministic"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

from firedrake_amd import forms, mesh as fmesh, op2  # noqa: E402
from helpers import oracle_run  # noqa: E402


def main():
    m = fmesh.UnitSquareMesh(12, 12, degrees=(1,), perturb=0.1)
    prob = forms.PoissonProblem(m, 1, bcs=True)
    V = prob.V
    cm = V.cell_node_map
    arrays, loops = {}, []

    def put(key, a):
        arrays[key] = np.ascontiguousarray(a)
        return key

    sizes = lambda s: [int(v) for v in s.sizes]           # noqa: E731
    mapd = {"type": "map", "values": put("map0", np.asarray(cm.values_with_halo, dtype=np.int32)), "arity": int(cm.arity),
            "iterset_sizes": sizes(m.cell_set), "toset_sizes": sizes(V.node_set), "offset": None, "offset_quotient": None}
    coords = np.array(m.coordinates.data_ro_with_halos)
    u, f = np.array(prob.u.data_ro_with_halos), np.array(prob.f.data_ro_with_halos)
    # residual loop
    r0 = np.zeros(V.node_set.total_size)
    r_after = oracle_run(prob.kres, m.cell_set, op2.Dat(V.node_set, r0.copy())(op2.INC, cm), m.coordinates(op2.READ, cm), prob.u(op2.READ, cm),
                         prob.f(op2.READ, cm))[0]
    dat = lambda acc, dim, before, after=None: dict(kind="dat", access=acc, dtype="float64", dim=dim, map=mapd, index=None,   # noqa: E731
                                                    dataset_sizes=sizes(V.node_set), before=before, **({"after": after} if after else {}))
    common = {"headers": [], "requires_zeroed_output_arguments": False, "iterset": {"sizes": sizes(m.cell_set), "name": "cells"},
              "iteration_region": "ALL", "pass_layer_arg": False, "subset": False, "extruded": False}
    loops.append({"kernel_name": prob.kres.name, "kernel_c": prob.kres.code, "form": "F", **common,
                  "args": [dat("INC", [1], put("r_before", r0), put("r_after", r_after)), dat("READ", [2], put("coords", coords)),
                           dat("READ", [1], put("u", u)), dat("READ", [1], put("f", f))]})
    # Jacobian loop with the BC lgmaps Firedrake swaps in (parloop.py:279-302)
    mat, jl = prob.jacobian()
    mpa = jl.arguments[0]
    ocsr = oracle_run(prob.kjac, m.cell_set, mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps), m.coordinates(op2.READ, cm))[0]
    loops.append({"kernel_name": prob.kjac.name, "kernel_c": prob.kjac.code, "form": "J", **common,
                  "args": [{"kind": "mat", "access": "INC", "dtype": "float64", "dims": [[1], [1]], "maps": [mapd, mapd], "unroll": False,
                            "lgmaps": [put("rlg", np.asarray(mpa.lgmaps[0], dtype=np.int32)), put("clg", np.asarray(mpa.lgmaps[1], dtype=np.int32))],
                            "row_sizes": sizes(V.node_set), "col_sizes": sizes(V.node_set)},
                           dat("READ", [2], "coords")]})
    # what assemble() returns: the residual with the BC rows zeroed (bcs.py:192-221), the matrix with the BC diagonal set
    F = r_after.copy()
    F[prob.bc_nodes] = 0.0
    vals = np.array(ocsr.values, copy=True)
    rp, ci = np.asarray(ocsr.rowptr), np.asarray(ocsr.colidx)
    for b in prob.bc_nodes:
        row = ci[rp[b]:rp[b + 1]]
        vals[rp[b] + int(np.searchsorted(row, b))] = 1.0
    outputs = {"F": {"data": put("F_data", F[:V.node_set.size])},
               "J": {"indptr": put("J_indptr", rp.astype(np.int32)), "indices": put("J_indices", ci.astype(np.int32)), "data": put("J_data", vals)}}
    meta = {"format": 1, "config": "c1", "full_size": False, "source": "synthetic", "firedrake_version": None, "scalar_type": "float64",
            "int_type": "int32", "dofs": int(V.node_set.size), "tolerances": {"vector": 1e-12, "matrix": 1e-12}, "parloops": loops, "outputs": outputs}
    path = os.path.join(HERE, "firedrake_synthetic_c1.npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
