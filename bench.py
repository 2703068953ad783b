#!/usr/bin/env python
"""bench.py -- assembled DoFs/s (residual + Jacobian) of the MI355X finite-element assembly path.

Workload (BASELINE.json configs[1], "C2"): Poisson CG1 on UnitCubeMesh(215^3) tets -- 59 630 250
cells, 10 077 696 DoFs per GPU; one "step" = one Newton-step assembly = assemble(F) + assemble(J):
zero the tensors, run the cell kernels (HIP wrapper kernels through the C ABI), exchange halos
(N > 1), apply the boundary conditions.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Weak scaling: every rank owns a 215-layer z-slab of a 215 x 215 x (215 N) cube grid.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC only on this driver (RCCL across processes)

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def algorithmic_bytes(ncell, arity, nnode, gdim, ncoeff, nnz=None):
    """every input array read once, every output written once (+ the zeroing pass)."""
    if nnz is None:   # residual: map + coords + coefficients + output write + zeroing
        return ncell * arity * 4 + nnode * gdim * 8 + ncoeff * nnode * 8 + nnode * 8 + nnode * 8
    return ncell * arity * 4 + nnode * gdim * 8 + nnz * 8 + nnz * 8      # Jacobian: map + coords + values + zeroing


def cpu_baseline(n_sample, degree, seconds_hint=20.0):
    """Time the oracle (CPU restatement of the PyOP2 wrapper, compiled with the reference's own
    flags) on a bounded sample of the same workload: an n_sample^3-cube mesh, 1 thread."""
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import forms, mesh as fmesh
    m = fmesh.UnitCubeMesh(n_sample, degrees=(degree,))
    V, X = m.space(degree), m.coord_space
    cm, xm = V.cell_node_map.values_with_halo, X.cell_node_map.values_with_halo
    nn = V.node_set.total_size
    coords = np.array(m.coordinates.data_ro_with_halos)
    pts = V.node_points
    u = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + 0.3 * pts[:, 2]
    f = (1 + 8 * np.pi ** 2) * np.cos(2 * np.pi * pts[:, 0]) * np.cos(2 * np.pi * pts[:, 1])
    r = np.zeros(nn)
    kr, kj = forms.poisson_residual_kernel(3, degree), forms.poisson_jacobian_kernel(3, degree)
    ncell = m.cell_set.size
    t0 = time.perf_counter()
    csr = oracle.build_sparsity(nn, nn, [(cm, cm)])
    t_sparsity = time.perf_counter() - t0
    def timed(threads):
        fn_r, a_r, k1, _ = oracle.par_loop(kr.code, kr.name, 0, ncell, [ODat(r, INC, cm), ODat(coords, READ, xm), ODat(u, READ, cm), ODat(f, READ, cm)],
                                           return_fn=True, threads=threads)
        fn_j, a_j, k2, cm_ = oracle.par_loop(kj.code, kj.name, 0, ncell, [OMat(csr, INC, cm, cm), ODat(coords, READ, xm)],
                                             return_fn=True, threads=threads)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            r[:] = 0
            fn_r(*a_r)
            t1 = time.perf_counter()
            csr.values[:] = 0
            fn_j(*a_j)
            t2 = time.perf_counter()
            ts.append((t2 - t0, t1 - t0, t2 - t1))
        ts.sort()
        return ts[len(ts) // 2]

    tot1, tr1, tj1 = timed(False)
    cores = os.cpu_count() or 1
    totN, trN, tjN = timed(True)          # OpenMP over cells, atomic scatter: the shared-memory analogue of N ranks
    multi = {"value": nn / totN, "cores": cores, "residual_dofs_per_s": nn / trN, "jacobian_dofs_per_s": nn / tjN,
             "note": "OpenMP over contiguous cell ranges, thread-private residual vectors summed afterwards, atomic CSR adds "
                     "(shared-memory analogue of N ranks)"}
    single = {"value": nn / tot1, "cores": 1, "residual_dofs_per_s": nn / tr1, "jacobian_dofs_per_s": nn / tj1,
              "note": "1 thread = what one MPI rank of the reference executes"}
    best = multi if multi["value"] >= single["value"] else single
    return {"value": best["value"], "unit": "DoFs/s", "cores": best["cores"], "kind": "port",
            "sample": f"Poisson CG{degree} on UnitCubeMesh({n_sample}) tets: {ncell} cells, {nn} DoFs, residual+Jacobian, "
                      f"median of 5; oracle = CPU restatement of the PyOP2 wrapper (not the reference binary), "
                      f"gcc -O3 -march=native -ffast-math; the faster of 1 thread and {cores} OpenMP threads is reported",
            "residual_dofs_per_s": best["residual_dofs_per_s"], "jacobian_dofs_per_s": best["jacobian_dofs_per_s"],
            "single_thread": single, "all_host_threads": multi, "sparsity_build_s": t_sparsity}


FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet; tools/microbench.hip measures 77.7 TFLOP/s with v_mfma_f64_16x16x4_f64


def run_c3(args):
    """BASELINE.json configs[2]: Helmholtz Q4 on an extruded hex mesh, stiffness+mass matrix by fp64 MFMA."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    _lib.require_gpu()
    n = args.n if args.n != 215 else 32
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m)
    for _ in range(args.warmup):
        prob.assemble_jacobian()
    _lib.call("fd_device_sync")
    ev = [(Event(), Event()) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        prob.assemble_jacobian()
        ev[k][1].record()
    _lib.call("fd_device_sync")
    elapsed = time.perf_counter() - t0
    ms = float(np.median([a.elapsed_ms(b) for a, b in ev]))
    ncell = m.ncells
    ndofs = m.node_set.size
    flops = prob.ALGO_FLOPS_PER_CELL * ncell
    out = {"metric": "assembled DoFs/sec (Jacobian)", "value": ndofs / (elapsed / args.steps), "unit": "DoFs/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"Helmholtz Q4 stiffness+mass on ExtrudedMesh(UnitSquareMesh({n},{n},quadrilateral), {n}) "
                                  f"(BASELINE.json configs[2])", "cells": ncell, "dofs": ndofs, "nnz": int(prob.sparsity.nz)},
           "roofline": {"kernel": "wrap_helmholtz_q4_hex_jacobian", "bound": "mfma", "achieved": flops / (ms * 1e-3) / 1e12,
                        "peak": FP64_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": flops / (ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                        "traffic": None, "ms": ms, "algorithmic_flops": flops,
                        "issued_mfma_flops": prob.FLOPS_PER_CELL * ncell},
           "cpu_baseline": None}
    print(json.dumps(out))


def run_c1(args):
    """BASELINE.json configs[0]: Poisson CG1 on UnitSquareMesh(64,64) -- launch-bound on a GPU; eager vs hipGraph replay."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.graph import CapturedStep
    _lib.require_gpu()
    prob = forms.PoissonProblem(fmesh.UnitSquareMesh(64, 64, perturb=0.1), 1, bcs=True)

    def step():
        prob.assemble_residual()
        prob.assemble_jacobian()

    for _ in range(max(args.warmup, 2)):
        step()
    _lib.call("fd_device_sync")
    n = max(args.steps, 200)
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    _lib.call("fd_device_sync")
    eager = (time.perf_counter() - t0) / n
    g = CapturedStep(step)
    g(); g.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        g()
    g.sync()
    graph = (time.perf_counter() - t0) / n
    nd = prob.V.node_set.size
    print(json.dumps({"metric": "assembled DoFs/sec (residual + Jacobian)", "value": nd / graph, "unit": "DoFs/s", "n_gpus": 1,
                      "steps": n, "warmup": args.warmup, "ms_per_step": graph * 1e3, "higher_is_better": True, "scaling": "weak",
                      "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": "Poisson CG1 residual+Jacobian on UnitSquareMesh(64,64) (BASELINE.json configs[0]), hipGraph replay",
                                 "cells": 8192, "dofs": nd},
                      "eager_ms_per_step": eager * 1e3, "graph_ms_per_step": graph * 1e3, "roofline": None, "cpu_baseline": None}))


def run_c4(args):
    """BASELINE.json configs[3]: DG_advection demo, DQ1 on quadrilaterals -- the 1-form L1 (cell + exterior-facet +
    interior-facet integrals, upwind flux) assembled matrix-free: three parloops INC-ing one Dat (C4)."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    _lib.require_gpu()
    n = args.n if args.n != 215 else 2048
    m = fmesh.make_quad_mesh(n, perturb=0.1)
    prob = forms.DGAdvectionProblem(m)
    for _ in range(max(args.warmup, 1)):
        prob.assemble_rhs()
    _lib.call("fd_device_sync")
    names = [lp.global_kernel.name for lp in prob.loops]
    ev = [[Event() for _ in range(len(prob.loops) + 1)] for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        # same sequence as DGAdvectionProblem.assemble_rhs, with an event between the three loops
        prob.L.zero()
        with prob.L.frozen_halo(forms.op2.INC):
            ev[k][0].record()
            for i, loop in enumerate(prob.loops):
                loop()
                ev[k][i + 1].record()
    _lib.call("fd_device_sync")
    elapsed = time.perf_counter() - t0
    per = [float(np.median([ev[k][i].elapsed_ms(ev[k][i + 1]) for k in range(args.steps)])) for i in range(len(prob.loops))]
    ncell, nint, next_ = m.cell_set.size, m.int_facet_set.size, m.ext_facet_set.size
    ndq, nq1 = m.dq_set.size, m.q1_set.size
    # algorithmic bytes per loop: maps + direct facet numbers + every node array once; the INC output is
    # written once by the first loop (+ the zeroing pass) and read+written by the loops that accumulate on top of it
    node_in = nq1 * (16 + 16) + ndq * 8
    b_cell = ncell * (4 + 4) * 4 + node_in + ndq * 8 + ndq * 8
    b_ext = next_ * ((4 + 4) * 4 + 4 + 4 * (16 + 16) + 4 * (8 + 16))     # boundary cells only: per-facet rows, no reuse
    b_int = nint * ((8 + 8) * 4 + 8) + node_in + ndq * 16
    roofs = []
    for name, ms, nbytes in zip(names, per, (b_cell, b_ext, b_int)):
        roofs.append({"kernel": name, "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms": ms, "algorithmic_bytes": nbytes})
    dominant = max(roofs, key=lambda r: r["ms"])
    print(json.dumps({"metric": "assembled DoFs/sec (DG advection RHS action)", "value": ndq / (elapsed / args.steps), "unit": "DoFs/s",
                      "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
                      "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
                      "config": {"workload": f"DG_advection demo 1-form L1, DQ1 on {n}x{n} quadrilaterals (BASELINE.json configs[3])",
                                 "cells": ncell, "dofs": ndq, "interior_facets": nint, "exterior_facets": next_},
                      "roofline": dominant, "roofline_per_loop": roofs, "cpu_baseline": None}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--size", dest="n", type=int, default=215, help="cubes per axis per GPU (215 -> ~10M DoF)")
    ap.add_argument("--degree", type=int, default=1)
    ap.add_argument("--cpu-sample", type=int, default=128, help="cube size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-bcs", action="store_true")
    ap.add_argument("--tile", type=str, default="8,8,4", help="cubes per traversal tile (= plan block)")
    ap.add_argument("--only", choices=["both", "residual", "jacobian"], default="both", help="profiling aid: run one form only")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4"], default="c2",
                    help="c2 = headline config (default); c1 = launch-bound 64x64 square (eager vs hipGraph); c3 = Q4 hex MFMA; "
                         "c4 = DG advection RHS action")
    args = ap.parse_args()
    if args.workload == "c3":
        import torch  # noqa: F401
        return run_c3(args)
    if args.workload == "c1":
        import torch  # noqa: F401
        return run_c1(args)
    if args.workload == "c4":
        import torch  # noqa: F401
        return run_c4(args)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    # test knobs (one-GPU rehearsal of the N > 1 path): FDHIP_FORCE_DEVICE pins every rank to one device and
    # FDHIP_DIST_BACKEND=gloo carries the halo buffers through the host instead of RCCL
    if os.environ.get("FDHIP_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["FDHIP_FORCE_DEVICE"])
    backend = os.environ.get("FDHIP_DIST_BACKEND", "nccl")
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    _lib.require_gpu()
    _lib.call("fd_set_device", local_rank)

    n = args.n
    t0 = time.perf_counter()
    mesh = fmesh.UnitCubeMesh((n, n, n * world), degrees=(args.degree,), rank=rank, nranks=world, perturb=0.1,
                              tile=tuple(int(v) for v in args.tile.split(",")))
    prob = forms.PoissonProblem(mesh, args.degree, bcs=not args.no_bcs)
    t_mesh = time.perf_counter() - t0
    V = prob.V
    ndofs_global = V.global_dofs
    ncell_local = mesh.cell_set.size
    t0 = time.perf_counter()
    mat, _ = prob.jacobian()
    mat.sparsity._build()
    nnz = mat.sparsity.nz
    _lib.call("fd_device_sync")
    t_sparsity = time.perf_counter() - t0

    ev = [[Event() for _ in range(4)] for _ in range(args.steps)]

    def step(k=None):
        if k is not None:
            ev[k][0].record()
        if args.only != "jacobian":
            if world > 1:
                prob.u.halo_valid = False      # a Newton step changes u: its ghost copies are refreshed every step
            prob.assemble_residual()
        if k is not None:
            ev[k][1].record()
            ev[k][2].record()
        if args.only != "residual":
            prob.assemble_jacobian()
        if k is not None:
            ev[k][3].record()

    def barrier():
        _lib.call("fd_device_sync")
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    t_res = float(np.median([ev[k][0].elapsed_ms(ev[k][1]) for k in range(args.steps)]))
    t_jac = float(np.median([ev[k][2].elapsed_ms(ev[k][3]) for k in range(args.steps)]))

    if rank == 0:
        arity = V.cell_node_map.arity
        nnode_local = V.node_set.total_size
        b_res = algorithmic_bytes(ncell_local, arity, nnode_local, 3, 2)
        b_jac = algorithmic_bytes(ncell_local, arity, nnode_local, 3, 0, nnz)
        roof_res = {"kernel": prob.res_loop.global_kernel.name, "bound": "hbm", "achieved": b_res / (t_res * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "ms": t_res, "algorithmic_bytes": b_res}
        roof_res["frac"] = roof_res["achieved"] / HBM_PEAK_GBS
        roof_jac = {"kernel": prob.jacobian()[1].global_kernel.name, "bound": "hbm", "achieved": b_jac / (t_jac * 1e-3) / 1e9,
                    "peak": HBM_PEAK_GBS, "unit": "GB/s", "traffic": None, "ms": t_jac, "algorithmic_bytes": b_jac}
        roof_jac["frac"] = roof_jac["achieved"] / HBM_PEAK_GBS
        # HBM bytes per launch from the rocprofv3 PMC passes of this same command (tools/pmc.sh ->
        # profiles/r1i_traffic.json: (2*FETCH_SIZE + WRITE_SIZE) KB, MI355X_MICROARCH.md HBM section); only
        # valid for the default workload it was collected on
        tname = next((t for t in ("r1j_traffic.json", "r1i_traffic.json")
                      if os.path.exists(os.path.join(ROOT, "profiles", t))), None)
        if tname and n == 215 and args.degree == 1 and world == 1 and args.tile == "8,8,4":
            tr = json.load(open(os.path.join(ROOT, "profiles", tname)))
            for roof in (roof_res, roof_jac):
                t = tr.get(roof["kernel"], {}).get("hbm_bytes_per_launch")
                if t:
                    roof["traffic"] = t
                    roof["traffic_source"] = f"profiles/{tname} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes)"
        dominant = roof_jac if t_jac >= t_res else roof_res
        out = {
            "metric": "assembled DoFs/sec (residual + Jacobian)",
            "value": ndofs_global / (ms_per_step * 1e-3),
            "unit": "DoFs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Poisson CG{args.degree} residual+Jacobian on UnitCubeMesh({n},{n},{n * world}) tets "
                                   f"(BASELINE.json configs[1]), z-slab per GPU",
                       "cells_per_gpu": ncell_local, "dofs_global": ndofs_global, "nnz_per_gpu": int(nnz),
                       "parallelism": f"domain-decomposition x{world}", "bcs": not args.no_bcs},
            "residual_dofs_per_s": V.node_set.size * world / (t_res * 1e-3),
            "jacobian_dofs_per_s": V.node_set.size * world / (t_jac * 1e-3),
            "roofline": dominant, "roofline_residual": roof_res, "roofline_jacobian": roof_jac,
            "setup_s": {"mesh": t_mesh, "sparsity_and_tables": t_sparsity},
        }
        if args.cpu_sample > 0 and world == 1:       # reported baseline: rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline(args.cpu_sample, args.degree)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
