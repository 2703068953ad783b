"""Poisson CG1 residual + Jacobian (the C2 step) on UnitCubeMesh(n) over a range of n: where does the device fill up, and what do
the small-loop leaves (configuration["small_loop_blocks"]) buy in between?  One line per (n, small_loop_blocks): eager and
hipGraph-replayed step time, DoFs/s of the replayed step.
    python tools/size_sweep.py [--sizes 8,16,24,32,48,64,96,128] [--blocks 0,256,512,1024]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from firedrake_amd import _lib, forms, mesh as fmesh                       # noqa: E402
from firedrake_amd.configuration import configuration                      # noqa: E402
from firedrake_amd.graph import CapturedStep                               # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--sizes", default="8,16,24,32,48,64,96,128")
ap.add_argument("--blocks", default="0,256,512,1024")
ap.add_argument("--steps", type=int, default=100)
args = ap.parse_args()
_lib.require_gpu()
sync = lambda: _lib.call("fd_device_sync")                                 # noqa: E731
for n in [int(s) for s in args.sizes.split(",")]:
    for sb in [int(s) for s in args.blocks.split(",")]:
        configuration["small_loop_blocks"] = sb
        mesh = fmesh.UnitCubeMesh((n, n, n), degrees=(1,), perturb=0.1, numbering="lexicographic")
        prob = forms.PoissonProblem(mesh, 1, bcs=True)

        def step():
            prob.assemble_residual()
            prob.assemble_jacobian()

        for _ in range(3):
            step()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        eager = (time.perf_counter() - t0) / args.steps
        g = CapturedStep(step)
        g(); g.sync()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(args.steps):
                g()
            g.sync()
            best = min(best, (time.perf_counter() - t0) / args.steps)
        nd = prob.V.node_set.size
        print(json.dumps({"n": n, "cells": mesh.cell_set.size, "dofs": nd, "small_loop_blocks": sb, "eager_ms": round(eager * 1e3, 4),
                          "graph_ms": round(best * 1e3, 4), "dofs_per_s": round(nd / best, 0)}), flush=True)
        del g, prob, mesh
