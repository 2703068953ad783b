#!/usr/bin/env python
"""bench.py -- assembled DoFs/s (residual + Jacobian) of the MI355X finite-element assembly path.

Workload (BASELINE.json configs[1], "C2"): Poisson CG1 on UnitCubeMesh(215^3) tets -- 59 630 250
cells, 10 077 696 DoFs per GPU, numbered like a DMPlex mesh (cells in plain order, nodes by first touch, no hints to the
backend); one "step" = one Newton-step assembly = assemble(F) + assemble(J):
zero the tensors, run the cell kernels (HIP wrapper kernels through the C ABI), exchange halos
(N > 1), apply the boundary conditions.  Inputs are resident in HBM before the timed region.

    python bench.py --gpus 1 --steps 10 --warmup 2
    python bench.py --gpus N ...                      # starts the N ranks itself (one process per GPU, RCCL over xGMI)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Headline line, every N: weak scaling of C2 -- each rank owns a 215^3 block of cubes (z-slabs: a 215 x 215 x 215N grid).
At N > 1 the same line carries ``strong_c5``: BASELINE.json configs[4] as written -- Poisson CG2 on the 215^3 cube,
80 062 991 DoFs, split over the N ranks -- and ``multi_gpu``: ranks of the RCCL communicator, exchange time, per-rank
kernel times, exchange/kernel overlap.  ``--workload c5`` runs configs[4] alone (N >= 2: the whole cube's ~2.3e9 nonzeros exceed the 32-bit CSR index range).
--gpus N with fewer than N visible devices FAILS.  Prints ONE JSON line on rank 0.
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


def algorithmic_bytes(ncell, arity, nnode, gdim, ncoeff, nnz=None, zeroing=False, ncoord=None, coord_arity=0):
    """every input array read once, every output written once.  ``zeroing`` adds the separate zeroing
    pass of a13 -- counted only where a separate pass is actually executed inside the bracket the time comes from
    (the owner-computes-rows Jacobian has none: it overwrites complete rows).  ``ncoord`` / ``coord_arity``: the coordinate
    field lives on its own (P1) space -- its nodes and its own cell map are charged instead of the unknown's (CG2: 1.26 M
    coordinate nodes + a 4-entry map row per cell, not 9.94 M nodes)."""
    coords = (nnode if ncoord is None else ncoord) * gdim * 8 + ncell * coord_arity * 4
    if nnz is None:   # residual kernel: map + coords + coefficients + output
        return ncell * arity * 4 + coords + ncoeff * nnode * 8 + nnode * 8 + (nnode * 8 if zeroing else 0)
    return ncell * arity * 4 + coords + nnz * 8 + (nnz * 8 if zeroing else 0)      # Jacobian: map + coords + values


def host_threads():
    """(threads to use, affinity count, cgroup CPU quota in cores or None): an OpenMP team larger than the CPU time the container
    may use only oversubscribes (the GPU box shows 256 logical CPUs under a 16-core quota)."""
    aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:          # "max 100000" or "<quota> <period>"
            q = fh.read().split()
            quota = None if q[0] == "max" else float(q[0]) / float(q[1])
    except (OSError, ValueError, IndexError):
        pass
    n = aff if quota is None else max(1, min(aff, int(quota)))
    return n, aff, quota


def cpu_baseline(mesh, degree, reps=2, pattern=None, label=None, threads_reps=10):
    """Time the oracle (CPU restatement of the PyOP2 wrapper, compiled with the reference's own flags) on ``mesh`` -- by default
    THE benchmark workload itself, a bounded number of repetitions.  Headline = 1 thread (what one MPI rank of the reference
    executes).  Beside it: min(affinity, cgroup quota) host threads, node-partitioned (each thread owns a contiguous node range
    and runs the cells touching it, dropping foreign rows -- the shared-memory analogue of N ranks with a ghost-cell layer).
    ``pattern`` = (rowptr, colidx) of the matrix when the caller holds it already (the device-built CSR, bit-identical to the
    oracle's -- tests/test_gpu_fullsize.py); else the oracle builds it."""
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import forms
    m = mesh
    V, X = m.space(degree), m.coord_space
    cm, xm = V.cell_node_map.values_with_halo, X.cell_node_map.values_with_halo
    nn = V.node_set.total_size
    coords = np.array(m.coordinates.data_ro_with_halos)
    pts = V.node_points
    u = np.sin(3 * pts[:, 0]) * np.cos(2 * pts[:, 1]) + (0.3 * pts[:, 2] if m.gdim == 3 else 0.0)
    f = (1 + 8 * np.pi ** 2) * np.cos(2 * np.pi * pts[:, 0]) * np.cos(2 * np.pi * pts[:, 1])
    r = np.zeros(nn)
    kr, kj = forms.poisson_residual_kernel(m.gdim, degree), forms.poisson_jacobian_kernel(m.gdim, degree)
    ncell = m.cell_set.size
    t_sparsity = None
    if pattern is None:
        t0 = time.perf_counter()
        csr = oracle.build_sparsity(nn, nn, [(cm, cm)])
        t_sparsity = time.perf_counter() - t0
    else:
        rp, ci = pattern
        csr = oracle.OracleCSR(nn, nn, 1, 1, np.ascontiguousarray(rp, dtype=np.int32), np.ascontiguousarray(ci, dtype=np.int32),
                               np.zeros(len(ci)))
    cores, aff, quota = host_threads()

    def timed(threads, nrep):
        kw = {}
        if threads:
            kw = {"threads": "owner", "owner_partition": oracle.make_owner_partition(cm, ncell, nn, cores)}
            os.environ["OMP_NUM_THREADS"] = str(cores)
        fn_r, a_r, k1, _ = oracle.par_loop(kr.code, kr.name, 0, ncell, [ODat(r, INC, cm), ODat(coords, READ, xm), ODat(u, READ, cm), ODat(f, READ, cm)],
                                           return_fn=True, **kw)
        fn_j, a_j, k2, cm_ = oracle.par_loop(kj.code, kj.name, 0, ncell, [OMat(csr, INC, cm, cm), ODat(coords, READ, xm)],
                                             return_fn=True, **kw)
        ts = []
        for _ in range(nrep + 1):          # first repetition = warm-up (page faults, thread pool), dropped
            t0 = time.perf_counter()
            r[:] = 0
            fn_r(*a_r)
            t1 = time.perf_counter()
            csr.values[:] = 0
            fn_j(*a_j)
            t2 = time.perf_counter()
            ts.append((t2 - t0, t1 - t0, t2 - t1))
        ts = sorted(ts[1:])
        return ts[len(ts) // 2]

    tot1, tr1, tj1 = timed(False, reps)
    totN, trN, tjN = timed(True, max(reps, threads_reps))
    multi = {"value": nn / totN, "cores": cores, "affinity_cpus": aff, "cgroup_cpu_quota_cores": quota, "repetitions": max(reps, threads_reps),
             "residual_dofs_per_s": nn / trN, "jacobian_dofs_per_s": nn / tjN,
             "note": "OpenMP team of min(affinity, cgroup quota) threads, one per contiguous node range running the cells that touch it "
                     "(owned + ghost cells), foreign rows dropped: no atomics, no private vectors (shared-memory analogue of N MPI ranks)"}
    return {"value": nn / tot1, "unit": "DoFs/s", "cores": 1, "kind": "port",
            "sample": f"{label or 'Poisson CG%d on a cube of tets' % degree}: {ncell} cells, {nn} DoFs, residual+Jacobian, "
                      f"median of {reps} warm repetitions (one more dropped as warm-up; SURVEY.md 8d asks for >= 10: the 1-thread leg takes seconds "
                      f"per repetition at this size and is held to {reps}, the all-threads leg runs {max(reps, threads_reps)}); oracle = CPU restatement of the PyOP2 wrapper "
                      f"(not the reference binary), gcc -O3 -march=native -ffast-math; 1 thread = what one MPI rank of the reference executes",
            "residual_dofs_per_s": nn / tr1, "jacobian_dofs_per_s": nn / tj1,
            "all_host_threads": multi, "sparsity_build_s": t_sparsity}


def oracle_step(loops, zero=(), pattern_of=None):
    """A CPU restatement of ``loops`` (firedrake_amd Parloops) through the oracle, on COPIES of their host data: returns a
    callable that runs them in order (outputs in ``zero`` -- Dats or Mats -- cleared first, as every assemble does).  Carriers
    shared between the loops (one coordinate Dat, one output Dat INC'ed by three loops) are shared here too."""
    import oracle
    from firedrake_amd.parloop import DatParloopArg, GlobalParloopArg, MatParloopArg
    arrays = {}

    def host(d):
        if id(d) not in arrays:
            arrays[id(d)] = np.array(d.data_ro_with_halos if hasattr(d, "data_ro_with_halos") else d.data_ro, copy=True)
        return arrays[id(d)]

    def csr_of(mat):
        if id(mat) not in arrays:
            sp = mat.sparsity
            rp, ci = np.ascontiguousarray(sp.rowptr, dtype=np.int32), np.ascontiguousarray(sp.colidx, dtype=np.int32)
            arrays[id(mat)] = oracle.OracleCSR(sp.nrows, sp.ncols, 1, 1, rp, ci, np.zeros(len(ci)))
        return arrays[id(mat)]

    calls = []
    for loop in loops:
        it = loop.iterset
        ext = bool(it._extruded)
        oargs = []
        for pa, acc in zip(loop.arguments, loop.accesses):
            if isinstance(pa, MatParloopArg):
                rm, cm = pa.maps
                lg = pa.lgmaps or (None, None)
                oargs.append(oracle.OMat(csr_of(pa.data), int(acc), rm._base().values_with_halo, cm._base().values_with_halo,
                                         roffset=rm.offset if ext else None, coffset=cm.offset if ext else None,
                                         row_lgmap=None if lg[0] is None else np.ascontiguousarray(lg[0], dtype=np.int32),
                                         col_lgmap=None if lg[1] is None else np.ascontiguousarray(lg[1], dtype=np.int32)))
            elif isinstance(pa, GlobalParloopArg):
                oargs.append(oracle.OGlobal(host(pa.data), int(acc)))
            elif isinstance(pa, DatParloopArg):
                m = pa.map_
                oargs.append(oracle.ODat(host(pa.data), int(acc), None if m is None else m._base().values_with_halo,
                                         offset=(m.offset if (ext and m is not None) else None)))
            else:
                raise TypeError(f"oracle_step: {type(pa).__name__}")
        lk = loop.global_kernel.local_kernel
        layers = None
        if ext:
            layers = tuple(int(x) for x in it.layers_array[0])
        subset = getattr(it, "indices", None) if type(it).__name__ == "Subset" else None
        fn, cargs, k1, k2 = oracle.par_loop(lk.code, lk.name, 0, it.size, oargs, subset=subset, layers=layers, return_fn=True,
                                            init_with_zero=bool(getattr(lk, "requires_zeroed_output_arguments", False)))
        calls.append((fn, cargs, k1, k2))

    def step():
        for z in zero:
            a = arrays.get(id(z))
            if a is not None:
                (a.values if hasattr(a, "values") else a)[...] = 0
        for fn, cargs, _, _ in calls:
            fn(*cargs)
    return step


def cpu_baseline_loops(loops, zero, ndofs, sample, reps=2):
    """``cpu_baseline`` object for a secondary config: the oracle on a bounded sample, 1 thread, median of ``reps`` warm
    repetitions (one more dropped as warm-up)."""
    try:
        step = oracle_step(loops, zero)
        ts = []
        for _ in range(reps + 1):
            t0 = time.perf_counter()
            step()
            ts.append(time.perf_counter() - t0)
        t = sorted(ts[1:])[len(ts[1:]) // 2]
        return {"value": ndofs / t, "unit": "DoFs/s", "cores": 1, "kind": "port", "seconds_per_step": t,
                "sample": sample + f"; oracle = CPU restatement of the PyOP2 wrapper (gcc -O3 -march=native -ffast-math), 1 thread, "
                                   f"median of {reps} warm repetitions"}
    except Exception as exc:
        return {"error": repr(exc)}


FP64_MFMA_PEAK_TFLOPS = 78.6   # AMD datasheet; tools/microbench.hip measures 77.7 TFLOP/s with v_mfma_f64_16x16x4_f64
FP64_VECTOR_PEAK_TFLOPS = 78.6 # same datasheet figure for vector fp64 FMA (256 CUs x 64 lanes x 2 flop x 2.4 GHz); microbench: 70 TFLOP/s


def _grid_of(loop):
    """Work-items of the most recent launch of ``loop``'s wrapper (firedrake_amd.kernel.last_launch), or None."""
    from firedrake_amd.kernel import last_launch
    return last_launch.get(loop.global_kernel.name, (None,))[0]


def measure_c3(n, steps, warmup, coefficients=False, cpu_sample=0):
    """BASELINE.json configs[2]: Helmholtz Q4 on an extruded hex mesh through ordinary parloops -- the stiffness+mass
    matrix on the fp64 matrix cores (tp_matrix wrapper) and the sum-factorised operator action (tp_action).  Reports the
    kernel alone AND the whole assemble (zeroing pass + kernel [+ BC diagonal]) against the fp64 MFMA peak."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    # coefficients: the same Q4 operator with a variable diffusivity field and a nonlinear-reaction linearisation point as
    # coefficient arguments (forms.CoefficientHexProblem) -- the shape of a TSFC Jacobian of a nonlinear problem
    prob = forms.CoefficientHexProblem(m, bcs=True, nq=5) if coefficients else forms.HelmholtzQ4Problem(m, bcs=True)
    t0 = time.perf_counter()
    prob.assemble_jacobian()
    prob.assemble_action()
    _lib.call("fd_device_sync")
    first = time.perf_counter() - t0
    for _ in range(max(warmup - 1, 0)):
        prob.assemble_jacobian()
        prob.assemble_action()
    _lib.call("fd_device_sync")
    ev = [[Event() for _ in range(7)] for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        ev[k][0].record()
        prob.assemble_jacobian(events=(ev[k][1], ev[k][2]))
        ev[k][3].record()
        prob.assemble_action(events=(ev[k][4], ev[k][5]))
        ev[k][6].record()
    _lib.call("fd_device_sync")
    elapsed = time.perf_counter() - t0
    med = lambda i, j: float(np.median([ev[k][i].elapsed_ms(ev[k][j]) for k in range(steps)]))
    ncell, ndofs, nnz = m.ncells, m.node_set.size, int(prob.sparsity.nz)
    flops = prob.ALGO_FLOPS_PER_CELL * ncell
    k_ms, a_ms = med(1, 2), med(0, 3)
    act_k_ms, act_a_ms = med(4, 5), med(3, 6)
    tf = lambda ms: flops / (ms * 1e-3) / 1e12
    # action: map (one 125-entry row per COLUMN) + Q1 coordinates + u read + y written (+ zeroing pass at assemble level)
    ncol = m.base_set.size
    act_bytes = ncol * 125 * 4 + ncol * 8 * 4 + m.coord_node_set.size * 24 + ndofs * 8 + ndofs * 8
    act_flops = ncell * (25 * 450 * 2 + 125 * 200)
    cpu = None
    if cpu_sample:
        ms_ = fmesh.make_extruded_hex_mesh(cpu_sample, cpu_sample, 4, perturb=0.1)
        ps_ = forms.CoefficientHexProblem(ms_, bcs=True, nq=5) if coefficients else forms.HelmholtzQ4Problem(ms_, bcs=True)
        ps_.sparsity._build()
        cpu = cpu_baseline_loops([ps_.jac_loop, ps_.act_loop], [ps_.mat, ps_.y], ms_.node_set.size,
                                 f"the same operator (matrix + action) on a {cpu_sample}^3-cell sample of the extruded hex mesh "
                                 f"({ms_.ncells} cells, {ms_.node_set.size} DoFs; dense 125 x 125 element matrices, MatSetValuesLocal by row search)", reps=1)
        del ps_, ms_
    return {"config": {"workload": f"Helmholtz Q4 stiffness+mass on ExtrudedMesh(UnitSquareMesh({n},{n},quadrilateral), {n}) "
                                   f"(BASELINE.json configs[2]), Dirichlet BCs", "cells": ncell, "dofs": ndofs, "nnz": nnz},
            "cpu_baseline": cpu,
            "jacobian_dofs_per_s": ndofs / (a_ms * 1e-3), "action_dofs_per_s": ndofs / (act_a_ms * 1e-3),
            "ms_per_step": elapsed / steps * 1e3, "first_call_s": first,
            "roofline": {"kernel": "wrap_helmholtz_q4_hex_jacobian", "grid": _grid_of(prob.jac_loop), "bound": "mfma", "achieved": tf(k_ms), "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": tf(k_ms) / FP64_MFMA_PEAK_TFLOPS, "traffic": None, "ms": k_ms,
                         "algorithmic_flops": flops, "issued_mfma_flops": prob.FLOPS_PER_CELL * ncell,
                         "assemble_ms": a_ms, "frac_assemble": tf(a_ms) / FP64_MFMA_PEAK_TFLOPS,
                         "note": "assemble = zeroing pass over the CSR values + MFMA kernel incl. its atomic scatter + BC diagonal"},
            "roofline_action": {"kernel": "wrap_helmholtz_q4_hex_action", "grid": _grid_of(prob.act_loop), "bound": "hbm", "achieved": act_bytes / (act_k_ms * 1e-3) / 1e9,
                                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": act_bytes / (act_k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": None, "ms": act_k_ms, "algorithmic_bytes": act_bytes, "assemble_ms": act_a_ms,
                                # the action's floor is fp64 VALU issue, not HBM: 6 axis passes (450 FMAs per line, 25 lines per cell)
                                # + ~200 flop of geometry / point weights at each of the 125 Gauss points
                                "valu_flops": act_flops, "valu_floor_ms": act_flops / (FP64_VECTOR_PEAK_TFLOPS * 1e12) * 1e3,
                                "frac_valu": act_flops / (FP64_VECTOR_PEAK_TFLOPS * 1e12) * 1e3 / act_k_ms}}


def measure_c3_action(n, steps, warmup):
    """The Q4 operator action on its own at a size whose matrix would not fit the 32-bit CSR index range (n = 64: 262 144 cells,
    16.97 M DoFs, 3.7e9 nonzeros if assembled): the Krylov inner loop of config 3's matrix-free use."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    prob = forms.HelmholtzQ4Problem(m, bcs=False, matrix=False)
    for _ in range(max(warmup, 1)):
        prob.assemble_action()
    _lib.call("fd_device_sync")
    ev = [(Event(), Event()) for _ in range(steps)]
    for k in range(steps):
        prob.assemble_action(events=ev[k])
    _lib.call("fd_device_sync")
    k_ms = float(np.median([a.elapsed_ms(b) for a, b in ev]))
    ncell, ndofs, ncol = m.ncells, m.node_set.size, m.base_set.size
    act_bytes = ncol * 125 * 4 + ncol * 8 * 4 + m.coord_node_set.size * 24 + ndofs * 8 + ndofs * 8
    act_flops = ncell * (25 * 450 * 2 + 125 * 200)
    return {"n": n, "cells": ncell, "dofs": ndofs, "kernel_ms": k_ms, "dofs_per_s": ndofs / (k_ms * 1e-3),
            "frac_hbm": act_bytes / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "valu_flops": act_flops,
            "frac_valu": act_flops / (FP64_VECTOR_PEAK_TFLOPS * 1e12) * 1e3 / k_ms}


def measure_tensor_forms(steps, warmup):
    """The wider tensor-product descriptors through the same two wrappers: the Newton Jacobian of int (1 + |grad u|^2) grad(u).grad(v) dx
    on Q3 (a coefficient GRADIENT at the Gauss points) and linear elasticity on (Q2)^3 (a vector-valued space: Mat dims (3, 3), nine
    scalar MFMA contractions per cell).  Kernel times and the fraction of the fp64 MFMA peak, algorithmic flops = 2 nd^2 4 nq per
    scalar block."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    out = {}
    # ... and the Helmholtz operator of config C3 beyond Q4 (round 6): 16-row panels of 14 / 22 / 32 tiles cut into column chunks of <= 8, one
    # wavefront per (panel, chunk); 8+ Gauss points per axis take their point weights plane by plane
    for key, degree, n, make in (("nonlinear_diffusion_q3", 3, 24, lambda m: forms.NonlinearDiffusionHexProblem(m, bcs=True)),
                                 ("elasticity_q2", 2, 24, lambda m: forms.ElasticityHexProblem(m, bcs=True)),
                                 ("helmholtz_q5", 5, 16, lambda m: forms.HelmholtzHexProblem(m, bcs=True, nq=6)),
                                 ("helmholtz_q6", 6, 12, lambda m: forms.HelmholtzHexProblem(m, bcs=True, nq=8)),
                                 ("helmholtz_q7", 7, 8, lambda m: forms.HelmholtzHexProblem(m, bcs=True, nq=9))):
        m = fmesh.make_extruded_hex_mesh(n, n, degree, perturb=0.1)
        prob = make(m)
        for _ in range(max(warmup, 1)):
            prob.assemble_jacobian()
            prob.assemble_action()
        _lib.call("fd_device_sync")
        ev = [[Event() for _ in range(4)] for _ in range(steps)]
        for k in range(steps):
            prob.assemble_jacobian(events=(ev[k][0], ev[k][1]))
            prob.assemble_action(events=(ev[k][2], ev[k][3]))
        _lib.call("fd_device_sync")
        k_ms = float(np.median([e[0].elapsed_ms(e[1]) for e in ev]))
        a_ms = float(np.median([e[2].elapsed_ms(e[3]) for e in ev]))
        flops = prob.ALGO_FLOPS_PER_CELL * m.ncells
        out[key] = {"workload": f"Q{degree} on ExtrudedMesh(UnitSquareMesh({n},{n},quadrilateral), {n}), Dirichlet BCs", "cells": m.ncells,
                    "dofs": int(m.node_set.size * (3 if key.startswith("elast") else 1)), "nnz": int(prob.sparsity.nz),
                    "matrix_kernel": prob.jac_loop.global_kernel.name, "matrix_kernel_ms": k_ms, "frac_mfma": flops / (k_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                    "algorithmic_flops": flops, "action_kernel_ms": a_ms}
        del prob, m
    return out


def run_c3(args):
    from firedrake_amd import _lib
    _lib.require_gpu()
    r = measure_c3(args.n if args.n else 32, args.steps, args.warmup, cpu_sample=8 if args.cpu_sample else 0)
    out = {"metric": "assembled DoFs/sec (Jacobian)", "value": r["jacobian_dofs_per_s"], "unit": "DoFs/s", "n_gpus": 1,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_step"], "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": r["config"],
           "action_dofs_per_s": r["action_dofs_per_s"], "roofline": r["roofline"], "roofline_action": r["roofline_action"],
           "first_call_s": r["first_call_s"], "cpu_baseline": r["cpu_baseline"]}
    print(json.dumps(out))


def measure_c1(steps, warmup, with_cpu=False):
    """BASELINE.json configs[0]: Poisson CG1 on UnitSquareMesh(64,64) -- launch-bound on a GPU; eager vs hipGraph replay."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.graph import CapturedStep
    prob = forms.PoissonProblem(fmesh.UnitSquareMesh(64, 64, perturb=0.1), 1, bcs=True)

    def step():
        prob.assemble_residual()
        prob.assemble_jacobian()

    for _ in range(max(warmup, 2)):
        step()
    _lib.call("fd_device_sync")
    n = max(steps, 200)
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
    own = None
    try:                                 # (replays on the graph's own stream instead of the stream eager launches go to)
        go = CapturedStep(step, ordered=False)
        go(); go.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            go()
        go.sync()
        own = (time.perf_counter() - t0) / n
        del go
    except Exception as exc:
        own = repr(exc)
    # the two assemblies write different tensors: forked onto a side stream inside the capture they become two branches of the
    # graph and their kernel chains (zero, residual, BC rows | Jacobian, BC diagonal) run side by side on the mostly idle device
    two = None
    try:
        from firedrake_amd.device import Stream
        side = Stream()

        def step2():
            with side.fork():
                prob.assemble_jacobian()
            prob.assemble_residual()
            side.join()

        g2 = CapturedStep(step2)
        g2(); g2.sync()
        t0 = time.perf_counter()
        for _ in range(n):
            g2()
        g2.sync()
        two = (time.perf_counter() - t0) / n
    except Exception as exc:
        two = repr(exc)
    nd = prob.V.node_set.size
    cpu = None
    if with_cpu:
        try:
            sp = prob.jacobian()[0].sparsity
            cpu = cpu_baseline(prob.mesh, 1, reps=10, pattern=(np.asarray(sp.rowptr), np.asarray(sp.colidx)),
                               label="Poisson CG1 on UnitSquareMesh(64,64), the workload itself (BASELINE.json configs[0]: the reference's CPU-runnable case)")
        except Exception as exc:
            cpu = {"error": repr(exc)}
    return {"metric": "assembled DoFs/sec (residual + Jacobian)", "value": nd / graph, "unit": "DoFs/s", "n_gpus": 1,
            "steps": n, "warmup": warmup, "ms_per_step": graph * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "Poisson CG1 residual+Jacobian on UnitSquareMesh(64,64) (BASELINE.json configs[0]), hipGraph replay",
                       "cells": 8192, "dofs": nd},
            "eager_ms_per_step": eager * 1e3, "graph_ms_per_step": graph * 1e3,
            "graph_own_stream_ms_per_step": own * 1e3 if isinstance(own, float) else own,
            "graph_two_branches_ms_per_step": two * 1e3 if isinstance(two, float) else two, "roofline": None, "cpu_baseline": cpu or None}


def run_c1(args):
    from firedrake_amd import _lib
    _lib.require_gpu()
    print(json.dumps(measure_c1(args.steps, args.warmup, with_cpu=args.cpu_sample > 0)))


def measure_c4(n, steps, warmup, cpu_sample=0):
    """BASELINE.json configs[3]: DG_advection demo, DQ1 on quadrilaterals -- the 1-form L1 (cell + exterior-facet +
    interior-facet integrals, upwind flux) assembled matrix-free: three parloops INC-ing one Dat (C4)."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    from firedrake_amd.device import Event
    m = fmesh.make_quad_mesh(n, perturb=0.1)
    prob = forms.DGAdvectionProblem(m)
    for _ in range(max(warmup, 1)):
        prob.assemble_rhs()
    _lib.call("fd_device_sync")
    names = [lp.global_kernel.name for lp in prob.loops]
    ev = [[Event() for _ in range(len(prob.loops) + 1)] for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        # same sequence as DGAdvectionProblem.assemble_rhs, with an event between the three loops
        prob.L.zero()
        with prob.L.frozen_halo(forms.op2.INC):
            ev[k][0].record()
            for i, loop in enumerate(prob.loops):
                loop()
                ev[k][i + 1].record()
    _lib.call("fd_device_sync")
    elapsed = time.perf_counter() - t0
    per = [float(np.median([ev[k][i].elapsed_ms(ev[k][i + 1]) for k in range(steps)])) for i in range(len(prob.loops))]
    ncell, nint, next_ = m.cell_set.size, m.int_facet_set.size, m.ext_facet_set.size
    ndq, nq1 = m.dq_set.size, m.q1_set.size
    # algorithmic bytes per loop: maps + direct facet numbers + every node array once; the INC output is
    # written once by the first loop (+ the zeroing pass) and read+written by the loops that accumulate on top of it
    node_in = nq1 * (16 + 16) + ndq * 8
    b_cell = ncell * (4 + 4) * 4 + node_in + ndq * 8 + ndq * 8
    b_ext = next_ * ((4 + 4) * 4 + 4 + 4 * (16 + 16) + 4 * (8 + 16))     # boundary cells only: per-facet rows, no reuse
    b_int = nint * ((8 + 8) * 4 + 8) + node_in + ndq * 16
    roofs = []
    for loop, name, ms, nbytes in zip(prob.loops, names, per, (b_cell, b_ext, b_int)):
        roofs.append({"kernel": name, "grid": _grid_of(loop), "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                      "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms": ms, "algorithmic_bytes": nbytes})
    dominant = max(roofs, key=lambda r: r["ms"])
    cpu = None
    if cpu_sample:
        ms_ = fmesh.make_quad_mesh(cpu_sample, perturb=0.1)
        ps_ = forms.DGAdvectionProblem(ms_)
        cpu = cpu_baseline_loops(ps_.loops, [ps_.L], ms_.dq_set.size,
                                 f"the same 1-form (cell + exterior-facet + interior-facet loops) on a {cpu_sample} x {cpu_sample} sample "
                                 f"({ms_.cell_set.size} cells, {ms_.dq_set.size} DoFs)", reps=3)
        del ps_, ms_
    return {"metric": "assembled DoFs/sec (DG advection RHS action)", "value": ndq / (elapsed / steps), "unit": "DoFs/s",
            "n_gpus": 1, "steps": steps, "warmup": warmup, "ms_per_step": elapsed / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"DG_advection demo 1-form L1, DQ1 on {n}x{n} quadrilaterals (BASELINE.json configs[3])",
                       "cells": ncell, "dofs": ndq, "interior_facets": nint, "exterior_facets": next_},
            "roofline": dominant, "roofline_per_loop": roofs, "cpu_baseline": cpu}


def run_c4(args):
    from firedrake_amd import _lib
    _lib.require_gpu()
    print(json.dumps(measure_c4(args.n if args.n else 2048, args.steps, args.warmup, cpu_sample=512 if args.cpu_sample else 0)))


CALIB = (("wrap_fd_calib_read2", 2), ("wrap_fd_calib_read4", 4), ("wrap_fd_calib_read8", 8), ("wrap_fd_calib_read16", 16),
         ("wrap_fd_calib_gather8", 8), ("wrap_fd_calib_write8", 8), ("wrap_fd_calib_atomic8", 8))


def run_calibration():
    """Launch the PMC calibration kernels (csrc/fd_builtin.hip) on 1 GiB buffers: known byte counts in the access widths
    of the wrapper kernels.  Only useful under rocprofv3 --pmc (bench.py --inner-pmc)."""
    import ctypes
    from firedrake_amd import _lib
    from firedrake_amd.device import DeviceBuffer
    nbytes = 1 << 30
    src, dst, sink = DeviceBuffer(nbytes), DeviceBuffer(nbytes), DeviceBuffer(8 * 65536)
    src.zero(); dst.zero()
    ngather = nbytes // 8
    idx = DeviceBuffer.from_numpy(np.random.default_rng(0).permutation(ngather).astype(np.int32))
    for name, width in CALIB:
        h = ctypes.c_void_p()
        _lib.call("fd_kernel_builtin", name.encode(), ctypes.byref(h))
        n = nbytes // width
        if name.endswith("gather8"):
            ptrs = [src.ptr, idx.ptr, sink.ptr]
        elif name.endswith("write8") or name.endswith("atomic8"):
            ptrs = [dst.ptr]
        else:
            ptrs = [src.ptr, sink.ptr]
        arr = (ctypes.c_void_p * len(ptrs))(*[ctypes.c_void_p(q) for q in ptrs])
        for _ in range(2):
            _lib.call("fd_kernel_launch", h.value, 0, n, arr, len(ptrs), 256, 0, 16384, 0, None)
    _lib.call("fd_device_sync")


PMC_PASS_TIMEOUT_S = 240


def collect_traffic(argv_tail, kernels):
    """HBM bytes per launch of ``kernels`` from rocprofv3 PMC passes of THIS command, run as child processes after the
    timed region: FETCH_SIZE and WRITE_SIZE in separate passes with --kernel-trace only (MI355X_MICROARCH.md, HBM and PMC
    sections).  The child also runs the calibration kernels, so the same CSVs hold the counter-to-byte factors of this
    session for 2/4/8/16-byte streaming reads, 8-byte gathers, 8-byte writes and fp64 atomics."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None, {"error": "rocprofv3 not found"}
    mean = {}
    env = dict(os.environ, TMPDIR="/tmp")
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix=f"fd_pmc_{counter}_", dir="/tmp")
        cmd = [rocprof, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               sys.executable, os.path.join(ROOT, "bench.py"), "--inner-pmc", *argv_tail]
        # own session + a hard limit: a profiler pass that wedges is killed with everything it started and the line goes out
        # with traffic = null (a normal pass takes ~30 s)
        try:
            proc = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
        except OSError as exc:
            return None, {"error": f"{counter} pass: {exc}"}
        try:
            out_, err_ = proc.communicate(timeout=PMC_PASS_TIMEOUT_S)
        except subprocess.TimeoutExpired:
            import signal
            try:
                os.killpg(proc.pid, signal.SIGKILL)
            except OSError:
                pass
            proc.wait()
            shutil.rmtree(d, ignore_errors=True)
            return None, {"error": f"{counter} pass exceeded {PMC_PASS_TIMEOUT_S} s and was killed"}
        r = subprocess.CompletedProcess(cmd, proc.returncode, out_, err_)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None, {"error": f"{counter} pass failed (rc {r.returncode}): {(r.stderr or r.stdout)[-300:]}"}
        acc = {}
        for fcsv in files:
            with open(fcsv) as fh:
                for row in csv.DictReader(fh):
                    if row.get("Counter_Name") == counter:
                        # keyed by (kernel, work-items of the dispatch): one wrapper launched on two problem sizes (the CG2 share and
                        # the whole configs[4] cube) must not be averaged together, nor lend its counters by name
                        try:
                            grid = int(float(row.get("Grid_Size") or 0))
                        except ValueError:
                            grid = 0
                        acc.setdefault((row["Kernel_Name"].split("(")[0].strip(), grid), []).append(float(row["Counter_Value"]))
        for k, v in acc.items():
            mean.setdefault(k, {})[counter] = sum(v) / len(v)
        shutil.rmtree(d, ignore_errors=True)
    calib = {}
    gib = float(1 << 30)
    by_name = {}
    for (name, grid), c in mean.items():
        by_name.setdefault(name, {})[grid] = c
    for name, width in CALIB:
        c = next(iter(by_name.get(name, {}).values()), None)
        if c:
            moved = gib * (1.5 if name.endswith("gather8") else 1.0)        # the gather also streams its 4-byte index array
            calib[name[len("wrap_fd_calib_"):]] = {
                "known_read_bytes": 0.0 if ("write" in name or "atomic" in name) else moved,
                "known_write_bytes": gib if ("write" in name or "atomic" in name) else 0.0,
                "FETCH_SIZE_kb": c.get("FETCH_SIZE"), "WRITE_SIZE_kb": c.get("WRITE_SIZE")}
    out = {}
    for k in kernels:
        for grid, c in by_name.get(k, {}).items():
            if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
                # guide formula: FETCH_SIZE counts 64 B per 128-B request on gfx950 -> doubled; WRITE_SIZE as reported
                out.setdefault(k, {})[grid] = {"hbm_bytes_per_launch": (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0,
                                               "FETCH_SIZE_kb": c["FETCH_SIZE"], "WRITE_SIZE_kb": c["WRITE_SIZE"]}
    return out, {"calibration": calib, "formula": "(2*FETCH_SIZE + WRITE_SIZE) KiB, separate --pmc passes of this command",
                 "keyed_by": "(kernel name, work-items of the dispatch): a roofline whose launch size the profiled child did not run keeps traffic = null"}


def match_traffic(tr, roof):
    """Counters of the profiled child for THIS roofline's launch: same kernel name and the same number of work-items (``grid``);
    a roofline that does not know its grid takes the kernel's counters only when the child ran a single launch size of it."""
    sizes = (tr or {}).get(roof.get("kernel"))
    if not sizes:
        return None
    if roof.get("grid") is not None:
        return sizes.get(int(roof["grid"]))
    return next(iter(sizes.values())) if len(sizes) == 1 else None


def measure(prob, args, world, dist, backend, torch):
    """Warm-up (first call timed on its own: JIT load + plan construction), then EXACTLY args.steps timed steps."""
    from firedrake_amd import _lib
    from firedrake_amd.device import Event

    def sync():
        _lib.call("fd_device_sync")

    def barrier():
        sync()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    first = {}
    do_res, do_jac = args.only != "jacobian", args.only != "residual"

    def step(ev=None, dom=None):
        """``ev``: the full set of 7 events (assemble-level and kernel-level brackets of both forms); ``dom``: the two events that
        bracket the DOMINANT kernel only -- what the timed region carries (an event is a barrier packet between two dependent
        kernels: 7 per step cost 20 us of a 1.3 ms step, profiles/r6p_overlap_probe_c2.txt)."""
        if ev is not None:
            ev[0].record()
        if do_res:
            prob.u.dat_version += 1            # a Newton step changes u: nothing cached on its values may be reused ...
            if world > 1:
                prob.u.halo_valid = False      # ... and its ghost copies are refreshed every step
            prob.assemble_residual(events=(ev[1], ev[2]) if ev is not None else (dom if (dom is not None and not do_jac) else None))
        if ev is not None:
            ev[3].record()
        if do_jac:
            prob.assemble_jacobian(events=(ev[4], ev[5]) if ev is not None else dom)
        if ev is not None:
            ev[6].record()

    # first call of each form on its own: code-object load + plan construction (one-off)
    sync()
    t0 = time.perf_counter()
    if do_res:
        prob.assemble_residual()
    sync()
    first["plans_residual_first_call"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    if do_jac:
        prob.assemble_jacobian()
    sync()
    first["plans_jacobian_first_call"] = time.perf_counter() - t0
    for _ in range(max(args.warmup - 1, 0)):
        step()
    # the timed region: EXACTLY args.steps steps, each carrying the HIP-event bracket of the dominant kernel (the Jacobian wrapper
    # when both forms run: 2/3 of the step in every configuration of this file) on the stream it is launched on
    dom = [(Event(), Event()) for _ in range(args.steps)]
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(dom=dom[k])
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    dom_ms = float(np.median([a.elapsed_ms(b) for a, b in dom]))
    # the breakdown (the other kernel, the assemble-level brackets): the same steps once more with all 7 events, after the clock stopped
    nd = max(min(args.steps, 10), 3)
    ev = [[Event() for _ in range(7)] for _ in range(nd)]
    for k in range(nd):
        step(ev[k])
    barrier()
    med = lambda i, j: float(np.median([ev[k][i].elapsed_ms(ev[k][j]) for k in range(nd)]))
    return {"ms_per_step": elapsed / args.steps * 1e3, "first": first,
            "res_kernel_ms": (med(1, 2) if do_jac else dom_ms) if do_res else None, "jac_kernel_ms": dom_ms if do_jac else None,
            "jac_kernel_ms_detail_pass": med(4, 5) if do_jac else None, "detail_steps": nd,
            "res_assemble_ms": med(0, 3), "jac_assemble_ms": med(3, 6)}


def exchange_only_ms(prob, reps, torch):
    """Halo traffic of one step on its own (SURVEY.md 8e deliverable): forward exchange of u, reverse exchange of r."""
    from firedrake_amd import _lib, op2
    halo = prob.V.node_set.halo
    if halo is None:
        return None
    _lib.call("fd_device_sync")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        prob.u.halo_valid = False
        prob.u.global_to_local_begin(op2.READ)
        prob.u.global_to_local_end(op2.READ)
        prob.r.local_to_global_begin(op2.INC)
        prob.r.local_to_global_end(op2.INC)
    _lib.call("fd_device_sync")
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def self_launch(args):
    """``python bench.py --gpus N`` with no launcher environment: start the N ranks here -- one process per GPU, the same
    environment ``torch.distributed.run`` would give them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), rendezvous on
    127.0.0.1 -- pass rank 0's JSON line through and fail if any rank fails."""
    import socket
    import subprocess
    n = args.gpus
    check_devices(n)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), FDHIP_BENCH_CHILD="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    try:
        pending = list(procs)
        while pending:
            for q in list(pending):
                code = q.poll()
                if code is None:
                    continue
                pending.remove(q)
                if code != 0:
                    rc = rc or code
                    for o in pending:           # a dead rank leaves the others in a collective: stop them (exact PIDs)
                        o.kill()
            time.sleep(0.05)
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    if rc:
        raise SystemExit(f"bench.py: a rank failed (exit code {rc})")


def check_devices(n):
    """--gpus N needs N visible devices: fail loudly instead of measuring something else.  FDHIP_FORCE_DEVICE (the
    one-GPU rehearsal of the N > 1 path used by the tests: every rank on one device, gloo wire) lifts the check."""
    import ctypes
    from firedrake_amd import _lib
    _lib.require_gpu()
    ndev = ctypes.c_int()
    _lib.call("fd_device_count", ctypes.byref(ndev))
    if ndev.value < n and os.environ.get("FDHIP_FORCE_DEVICE") is None:
        raise SystemExit(f"bench.py: --gpus {n} but only {ndev.value} HIP device(s) visible; refusing to run a different "
                         f"configuration (set FDHIP_FORCE_DEVICE=0 FDHIP_DIST_BACKEND=gloo for the one-device rehearsal)")
    return ndev.value


def jacobian_accumulation(prob, args, ab):
    """{"accumulation": ...} of ``prob``'s Jacobian loop as it just ran, plus (``ab``) the kernel time of the same loop rebuilt with
    the other setting of FDHIP_OCR_FIXED_POINT on the same mesh."""
    from firedrake_amd import _lib, forms
    from firedrake_amd.configuration import configuration
    from firedrake_amd.device import Event

    def mode_of(p):
        loop = p.jacobian()[1]
        geos = [g for key, g in (loop._prepared or {}).get("parts", {}).items() if key[0] == "ocr" and isinstance(g, dict)]
        fx = any(g["cw"].src.mode.endswith("_fx") for g in geos)
        return ("fixed-point, quantum 2^-44..2^-47 of the row block's largest contribution (opt-in, normwise only)" if fx
                else "fp64 (ds_add_f64 in LDS; global_atomic_add_f64 off the owner-computes-rows path)"), fx, bool(geos)

    name, fx, ocr = mode_of(prob)
    out = {"accumulation": name}
    whole = ocr and not prob.jacobian()[1]._prepared["cw"].src.mode.startswith("ocrs")
    if not (ab and whole):
        return out
    saved = configuration["ocr_fixed_point"]
    try:
        configuration["ocr_fixed_point"] = 0 if fx else 1
        p2 = forms.PoissonProblem(prob.mesh, prob.degree, bcs=len(prob.bc_nodes) > 0)
        for _ in range(3):
            p2.assemble_jacobian()
        ev = [(Event(), Event()) for _ in range(max(min(args.steps, 10), 3))]
        for e in ev:
            p2.assemble_jacobian(events=e)
        _lib.call("fd_device_sync")
        name2, fx2, _ = mode_of(p2)
        out["other_accumulation"] = {"accumulation": name2, "kernel_ms": float(np.median([a.elapsed_ms(b) for a, b in ev])),
                                     "switch": f"FDHIP_OCR_FIXED_POINT={0 if fx else 1}"}
        del p2
    except Exception as exc:
        out["other_accumulation"] = {"error": repr(exc)}
    finally:
        configuration["ocr_fixed_point"] = saved
    return out


def poisson_line(args, ctx, degree, shape, scaling, label, numbering, variants, traffic, cpu=False, cpu_reps=None, accum_ab=False):
    """Measure residual + Jacobian assembly of Poisson CG<degree> on UnitCubeMesh(shape) box-partitioned over the ranks and
    return rank 0's result dict (None on the other ranks)."""
    from firedrake_amd import _lib, forms, mesh as fmesh
    rank, world, dist, backend, torch = ctx["rank"], ctx["world"], ctx["dist"], ctx["backend"], ctx["torch"]
    tile = tuple(int(v) for v in args.tile.split(","))

    def build(nb):
        t0 = time.perf_counter()
        mesh = fmesh.UnitCubeMesh(shape, degrees=(degree,), rank=rank, nranks=world, perturb=0.1, tile=tile, numbering=nb,
                                  partition=args.partition)
        prob = forms.PoissonProblem(mesh, degree, bcs=not args.no_bcs)
        t_mesh = time.perf_counter() - t0
        # inputs resident in HBM before anything is timed: the Maps and the coordinate / coefficient Dats (host -> device copies
        # of the mesh data, ~1 GB at C2 size; they used to be charged to whichever setup phase touched a carrier first)
        t0 = time.perf_counter()
        for mp in {id(m_): m_ for m_ in (prob.V.cell_node_map, mesh.coord_space.cell_node_map)}.values():
            mp._dev_values()
        for d in (mesh.coordinates, prob.u, prob.f):
            d._dev_ptr(False)
        _lib.call("fd_device_sync")
        t_up = time.perf_counter() - t0
        t0 = time.perf_counter()
        mat, _ = prob.jacobian()
        mat.sparsity._build()
        _lib.call("fd_device_sync")
        return mesh, prob, {"mesh": t_mesh, "upload": t_up, "sparsity": time.perf_counter() - t0}

    mesh, prob, setup = build(numbering)
    V = prob.V
    ndofs_global = V.global_dofs
    ncell_local = mesh.cell_set.size
    nnz = prob.jacobian()[0].sparsity.nz
    res = measure(prob, args, world, dist, backend, torch)
    setup.update(res["first"])
    if args.inner_pmc:
        return None
    multi = None
    if world > 1:
        multi = multi_gpu_detail(prob, args, res, ctx)
    out = None
    if rank == 0:
        arity = V.cell_node_map.arity
        nnode_local = V.node_set.total_size
        kres, kjac = prob.res_loop.global_kernel.name, prob.jacobian()[1].global_kernel.name
        jac_ocr = prob.jacobian()[1]._prepared["cw"].src.mode.startswith("ocr") if prob.jacobian()[1]._prepared else False

        def roof(kernel, ms, nbytes, **extra):
            from firedrake_amd.kernel import last_launch
            d = {"kernel": kernel, "bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "ms": ms, "algorithmic_bytes": nbytes,
                 "grid": last_launch.get(kernel, (None,))[0]}
            d.update(extra)
            return d

        roof_res = roof_jac = None
        xmap = mesh.coord_space.cell_node_map
        coord_kw = {} if xmap is V.cell_node_map else {"ncoord": mesh.coord_space.node_set.total_size, "coord_arity": xmap.arity}
        if res["res_kernel_ms"]:
            # the wrapper kernel alone: map + coords + 2 coefficients + the output, against the kernel's own duration
            b = algorithmic_bytes(ncell_local, arity, nnode_local, 3, 2, **coord_kw)
            roof_res = roof(kres, res["res_kernel_ms"], b, assemble_ms=res["res_assemble_ms"],
                            events=("HIP events around this kernel in every step of the timed region" if not res["jac_kernel_ms"] else
                                    f"HIP events in a pass of {res['detail_steps']} further steps after the timed region (the timed steps "
                                    "carry the dominant kernel's bracket only); assemble_ms from the same pass"),
                            note="kernel-only bytes and time; assemble_ms adds the zeroing pass (a13) and the BC fix-up (a14)"
                                 + ("; at N > 1 the bracket also holds the forward halo exchange (see multi_gpu.residual_kernel_only_ms)" if world > 1 else ""))
        if res["jac_kernel_ms"]:
            # owner-computes-rows writes complete rows and performs NO zeroing pass: strict bytes = map + coords + values.
            # frac_with_zeroing credits the nnz*8 zeroing traffic SURVEY.md 8(d) lists for the reference's two-pass scheme.
            b = algorithmic_bytes(ncell_local, arity, nnode_local, 3, 0, nnz, zeroing=not jac_ocr, **coord_kw)
            bz = algorithmic_bytes(ncell_local, arity, nnode_local, 3, 0, nnz, zeroing=True, **coord_kw)
            roof_jac = roof(kjac, res["jac_kernel_ms"], b, assemble_ms=res["jac_assemble_ms"],
                            events="HIP events around this kernel in every step of the timed region, median; assemble_ms and "
                                   f"ms_detail_pass from a pass of {res['detail_steps']} further steps with all 7 events per step",
                            ms_detail_pass=res["jac_kernel_ms_detail_pass"],
                            frac_with_zeroing=bz / (res["jac_kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                            note="kernel-only; strict bytes (no credit for the zeroing pass the kernel makes unnecessary)")
            # how the element matrices were ADDED in the timed launches (the reference: fp64, MatSetValuesLocal ADD_VALUES,
            # builder.py:573-625), and -- whole-entity owner-computes-rows loops at N = 1 -- the other mode's kernel time beside it
            roof_jac.update(jacobian_accumulation(prob, args, accum_ab and world == 1))
        roofs = [r for r in (roof_res, roof_jac) if r]
        dominant = max(roofs, key=lambda r: r["ms"])
        traffic_meta = None
        n_gpus = multi["n_gpus"] if multi else 1
        out = {
            "metric": "assembled DoFs/sec (residual + Jacobian)",
            "value": ndofs_global / (res["ms_per_step"] * 1e-3),
            "unit": "DoFs/s", "n_gpus": n_gpus, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"Poisson CG{degree} residual+Jacobian on UnitCubeMesh({shape[0]},{shape[1]},{shape[2]}) tets "
                                   f"({label}), partition {'x'.join(str(v) for v in mesh.partition)}",
                       "cells_per_gpu": ncell_local, "dofs_global": ndofs_global, "nnz_per_gpu": int(nnz),
                       "parallelism": f"domain-decomposition x{world}", "bcs": not args.no_bcs, "numbering": numbering},
            "residual_dofs_per_s": ndofs_global / (res["res_assemble_ms"] * 1e-3) if roof_res else None,
            "jacobian_dofs_per_s": ndofs_global / (res["jac_assemble_ms"] * 1e-3) if roof_jac else None,
            "roofline": dominant, "roofline_residual": roof_res, "roofline_jacobian": roof_jac,
            "traffic_meta": traffic_meta,
            "setup_s": setup,
            "steps_to_amortise_setup": (setup.get("plans_residual_first_call", 0) + setup.get("plans_jacobian_first_call", 0) + setup["sparsity"])
                                       / (res["ms_per_step"] * 1e-3),
            "exchange_ms": multi["exchange_ms"] if multi else None,
            "multi_gpu": multi,
        }
    if cpu and out is not None:
        # reported baseline (rank 0, N = 1): the oracle on THIS workload -- the same mesh object, the matrix pattern the device built
        try:
            sp = prob.jacobian()[0].sparsity
            out["cpu_baseline"] = cpu_baseline(mesh, degree, reps=cpu_reps or args.cpu_reps, pattern=(np.asarray(sp.rowptr), np.asarray(sp.colidx)),
                                               label=f"Poisson CG{degree} on UnitCubeMesh({shape[0]},{shape[1]},{shape[2]}) tets, the benchmark workload itself")
        except Exception as exc:                  # (a host without a C compiler, ...): the GPU line still goes out
            out["cpu_baseline"] = {"error": repr(exc)}
    # further numberings (no producer hints): locality dependence of the same step, N = 1 only
    if world == 1 and variants:
        for nb in [v for v in variants.split(",") if v and v != numbering]:
            del prob, mesh
            import gc
            gc.collect()
            mesh = prob = None
            try:                                          # the headline line must not die with a variant
                mesh, prob, st = build(nb)
                r2 = measure(prob, args, world, dist, backend, torch)
                out[f"value_{nb}_numbering"] = ndofs_global / (r2["ms_per_step"] * 1e-3)
                out[f"detail_{nb}_numbering"] = {"ms_per_step": r2["ms_per_step"], "residual_kernel_ms": r2["res_kernel_ms"],
                                                 "jacobian_kernel_ms": r2["jac_kernel_ms"], "setup_s": {**st, **r2["first"]}}
            except Exception as exc:
                out[f"value_{nb}_numbering"] = None
                out[f"detail_{nb}_numbering"] = {"error": repr(exc)}
    del prob, mesh
    import gc
    gc.collect()
    return out


def redundant_instances_frac(prob):
    """Share of this rank's owner-computes-rows Jacobian instances that sit on GHOST cells -- cells another rank owns and this one
    evaluates again so that its own rows are complete without a matrix exchange (SURVEY.md 8e option 1; the reference ships the
    foreign rows in MatAssemblyBegin/End instead, mat.py:776, 940-954).  None when the loop took another wrapper."""
    try:
        loop = prob.jacobian()[1]
        ncell_owned = prob.mesh.cell_set.size
        tot = red = 0
        for key, geo in (loop._prepared or {}).get("parts", {}).items():
            if key[0] != "ocr" or not isinstance(geo, dict):
                continue
            op = geo["ocr"]
            if op.ninst == 0:
                continue
            from firedrake_amd.device import DeviceBuffer
            ent = DeviceBuffer.wrap(op.inst_ent, int(op.ninst) * 4, owned=False).download(np.int32, (int(op.ninst),))
            live = np.ones(len(ent), dtype=bool)
            if getattr(op, "valid", None):
                live = DeviceBuffer.wrap(op.valid, int(op.ninst), owned=False).download(np.uint8, (int(op.ninst),)) != 0
            tot += int(live.sum())
            red += int((live & (ent >= ncell_owned)).sum())
        return red / tot if tot else None
    except Exception:
        return None


def multi_gpu_detail(prob, args, res, ctx):
    """N > 1 (SURVEY.md 8e deliverables): ranks of the library's own communicator, the wire in use, the halo traffic of one
    step on its own, per-rank kernel times, and how much of the exchange the core-entity kernel hides (parloop.py:250-253)."""
    from firedrake_amd import _lib, halo as fhalo
    from firedrake_amd.device import Event
    import ctypes
    rank, world, dist, torch = ctx["rank"], ctx["world"], ctx["dist"], ctx["torch"]
    h = prob.V.node_set.halo
    wire = h.wire if h is not None else None
    n_comm = world
    comm = fhalo.communicator()
    if comm:
        r_, n_ = ctypes.c_int(), ctypes.c_int()
        _lib.call("fd_comm_info", comm, ctypes.byref(r_), ctypes.byref(n_))
        n_comm = n_.value
    exch = exchange_only_ms(prob, max(args.steps, 5), torch)
    # the residual's kernels on their own: halos kept valid, so no exchange inside the bracket
    reps = max(args.steps, 5)
    ev = [(Event(), Event()) for _ in range(reps)]
    for k in range(reps):
        prob.u.halo_valid = True
        prob.r._halo_frozen, prob.r._frozen_access_mode = True, prob.res_loop.accesses[0]
        try:
            ev[k][0].record()
            prob.res_loop()
            ev[k][1].record()
        finally:
            prob.r._halo_frozen, prob.r._frozen_access_mode = False, None
    _lib.call("fd_device_sync")
    k_only = float(np.median([a.elapsed_ms(b) for a, b in ev]))
    mine = {"rank": rank, "residual_kernel_only_ms": k_only, "residual_with_exchange_ms": res["res_kernel_ms"],
            "jacobian_kernel_ms": res["jac_kernel_ms"], "exchange_ms": exch,
            "jacobian_redundant_instances_frac": redundant_instances_frac(prob),
            "halo_rows": {"send": int(sum(len(v) for v in h.lists.send.values())), "recv": int(sum(len(v) for v in h.lists.recv.values())),
                          "neighbours": len(set(h.lists.send) | set(h.lists.recv))} if h is not None else None}
    allr = [None] * world
    dist.all_gather_object(allr, mine)
    if rank != 0:
        return None
    kmax = max(r["residual_kernel_only_ms"] for r in allr)
    emax = max(r["exchange_ms"] or 0.0 for r in allr)
    wmax = max(r["residual_with_exchange_ms"] or 0.0 for r in allr)
    # bracket = forward exchange + kernels + reverse exchange; exchange_ms = forward + reverse on their own
    hidden = max(0.0, min(emax, kmax + emax - wmax))
    # what each partition shape costs for THIS cube and rank count (mesh.partition_overheads): the ghost cube layers every rank
    # computes again for the owner-computes-rows Jacobian, the halo rows and the neighbours of the worst rank
    from firedrake_amd.mesh import partition_overheads
    shape, deg = getattr(prob.mesh, "shape", None), getattr(prob, "degree", 1)
    overheads = None
    if shape is not None:
        overheads = {name: partition_overheads(shape, world, name, deg) for name in ("slabs", "blocks")}
    return {"n_gpus": n_comm, "wire": wire, "communicator": fhalo.communicator_status(), "exchange_ms": emax, "partition_overheads": overheads,
            "residual_kernel_only_ms": kmax, "residual_with_exchange_ms": wmax, "exchange_hidden_ms": hidden,
            "exchange_hidden_frac": hidden / emax if emax > 0 else None,
            "jacobian_redundant_instances_frac": max((r.get("jacobian_redundant_instances_frac") or 0.0) for r in allr), "per_rank": allr}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--size", dest="n", type=int, default=0,
                    help="cubes per axis (c2: per GPU, default 215 -> ~10M CG1 DoF per GPU; c5: of the WHOLE cube, default 215)")
    ap.add_argument("--n5", type=int, default=0, help="cubes per axis of the strong-scaling C5 cube appended at N > 1 (default 215)")
    ap.add_argument("--degree", type=int, default=0)
    ap.add_argument("--cpu-sample", type=int, default=1, help="0 = skip the CPU baseline (the oracle timed on the workload itself)")
    ap.add_argument("--cpu-reps", type=int, default=3, help="timed repetitions of the CPU baseline (one more is dropped as warm-up)")
    ap.add_argument("--no-bcs", action="store_true")
    ap.add_argument("--tile", type=str, default="8,8,4", help="cubes per traversal tile (= plan block)")
    ap.add_argument("--numbering", choices=["tiled", "lexicographic", "random"], default="lexicographic",
                    help="entity numbering of the headline measurement (SURVEY.md 8d).  Default: lexicographic cells, first-touch "
                         "nodes, NO producer hints -- what a DMPlex-produced mesh looks like to the backend (dmcommon.pyx:2688-2712); "
                         "'tiled' adds the producer's tile boundaries as hints, 'random' is the worst case")
    ap.add_argument("--variants", type=str, default="tiled,random",
                    help="further numberings measured after the headline one at N=1; '' = none")
    ap.add_argument("--partition", choices=["slabs", "blocks"], default="slabs",
                    help="N > 1: z-slabs (2 neighbours) or the most cubic process grid (8 -> 2x2x2), SURVEY.md 8e")
    ap.add_argument("--traffic", choices=["auto", "off"], default="auto",
                    help="auto: rocprofv3 PMC passes of this command (child processes) fill roofline.traffic at N=1")
    ap.add_argument("--inner-pmc", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--no-secondary", dest="secondary", action="store_false",
                    help="skip the secondary sections: config C3 (Q4 hex, fp64 MFMA) at N=1, config C5 (strong scaling) at N>1")
    ap.add_argument("--only", choices=["both", "residual", "jacobian"], default="both", help="profiling aid: run one form only")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4", "c5"], default="c2",
                    help="c2 = headline config (default; weak scaling: one 215^3 cube per GPU); c1 = launch-bound 64x64 square "
                         "(eager vs hipGraph); c3 = Q4 hex MFMA; c4 = DG advection RHS action; c5 = BASELINE configs[4] as written: "
                         "Poisson CG2 on the 215^3 cube, 80 062 991 DoFs, split over the N GPUs (strong scaling)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.workload in ("c1", "c3", "c4"):
        if args.gpus != 1:
            raise SystemExit(f"--workload {args.workload} is a single-GPU configuration")
        import torch  # noqa: F401
        return {"c1": run_c1, "c3": run_c3, "c4": run_c4}[args.workload](args)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        return self_launch(args)           # no launcher: start the ranks ourselves
    world = int(env_world) if env_world is not None else 1
    if args.gpus not in (1, world):
        raise SystemExit(f"bench.py: --gpus {args.gpus} disagrees with WORLD_SIZE={world}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    # test knobs (one-GPU rehearsal of the N > 1 path): FDHIP_FORCE_DEVICE pins every rank to one device and
    # FDHIP_DIST_BACKEND=gloo carries the halo buffers through the host instead of RCCL
    if os.environ.get("FDHIP_FORCE_DEVICE") is not None:
        local_rank = int(os.environ["FDHIP_FORCE_DEVICE"])
    backend = os.environ.get("FDHIP_DIST_BACKEND", "nccl")
    check_devices(world)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        # the line this program prints is its whole stdout: rendezvous chatter of the backends ("[Gloo] Rank 0 is connected ...")
        # goes to stderr
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
            else:
                dist.init_process_group(backend)
            dist.barrier()
        finally:
            os.dup2(saved, 1)
            os.close(saved)
    from firedrake_amd import _lib
    _lib.require_gpu()
    _lib.call("fd_set_device", local_rank)
    ctx = {"rank": rank, "world": world, "dist": dist, "backend": backend, "torch": torch}
    from firedrake_amd.mesh import partition_grid
    pg = partition_grid(world, args.partition)

    if args.workload == "c5":
        # BASELINE configs[4] as written: ONE cube, split over the ranks -> strong scaling
        n = args.n or 215
        degree = args.degree or 2
        # (the whole 80 M-DoF CG2 cube holds ~2.3e9 nonzeros: row starts are 64-bit, include/fdhip.h fd_nnz_t, so N = 1 anchors the curve)
        out = poisson_line(args, ctx, degree, (n, n, n), "strong", "BASELINE.json configs[4]" + ("" if n == 215 and degree == 2 else ", reduced"),
                           args.numbering, "", False)
    else:
        n = args.n or 215
        degree = args.degree or 1
        shape = (n * pg[0], n * pg[1], n * pg[2])        # weak scaling: every rank owns an n^3 cube of cubes
        out = poisson_line(args, ctx, degree, shape, "weak", "BASELINE.json configs[1] per GPU" if degree == 1 else f"CG{degree}, weak",
                           args.numbering, args.variants, args.traffic == "auto", cpu=(args.cpu_sample > 0 and world == 1),
                           accum_ab=(not args.inner_pmc and args.traffic == "auto"))      # (profiled / traced runs: one mode per kernel name)
    if args.inner_pmc:
        # profiled child of collect_traffic(): the secondary configs' kernels in the same pass (a few launches each)
        if args.secondary and args.workload == "c2" and world == 1:
            for fn in (lambda: measure_c3(32, 2, 1), lambda: measure_c4(2048, 2, 1),
                       lambda: poisson_line(args, ctx, 2, (107, 107, 107), "weak", "", "lexicographic", "", False)):
                try:
                    fn()
                except Exception as exc:
                    print(f"bench.py --inner-pmc: {exc!r}", file=sys.stderr)
        run_calibration()
        return
    if args.secondary and args.workload == "c2" and args.only == "both" and world > 1:
        # BASELINE configs[4] in the same run: the 215^3 CG2 cube split over the same ranks (strong scaling)
        try:
            n5 = args.n5 or 215
            c5 = poisson_line(args, ctx, 2, (n5, n5, n5), "strong", "BASELINE.json configs[4]" + ("" if n5 == 215 else ", reduced"),
                              args.numbering, "", False)
            if rank == 0:
                out["strong_c5"] = c5
        except Exception as exc:
            if rank == 0:
                out["strong_c5"] = {"error": repr(exc)}
            raise
    if rank == 0 and world == 1 and args.secondary and args.workload == "c2" and args.only == "both":
        # the other BASELINE configs in the same driver run (each guarded: the headline line must not die with a secondary one):
        # C3 = north_star's second numeric target (fp64 MFMA fraction on the Q4 hex config), C4 = the DG advection right-hand side,
        # C5 share = the CG2 problem one of 8 GPUs holds (n = 107, un-hinted numbering), C1 = the launch-bound 64 x 64 case
        def guarded(key, fn):
            try:
                out[key] = fn()
            except Exception as exc:
                out[key] = {"error": repr(exc)}
            import gc
            gc.collect()

        cs = args.cpu_sample > 0
        guarded("secondary_c3", lambda: measure_c3(32, max(3, args.steps // 2), 2, cpu_sample=8 if cs else 0))
        if "error" not in out["secondary_c3"]:
            # the same Q4 operator with two coefficient fields (variable diffusivity, nonlinear-reaction linearisation point)
            def with_coefficients():
                r = measure_c3(32, 3, 1, coefficients=True)
                return {"workload": "a(du, v) = int (1 + w0) grad(du).grad(v) + (1 + u0^2) du v dx, w0 and u0 Q4 fields (coefficient arguments)",
                        "matrix_kernel_ms": r["roofline"]["ms"], "frac_mfma": r["roofline"]["frac"], "assemble_ms": r["roofline"]["assemble_ms"],
                        "action_kernel_ms": r["roofline_action"]["ms"], "action_frac_valu": r["roofline_action"]["frac_valu"]}
            try:
                out["secondary_c3"]["variable_coefficients"] = with_coefficients()
            except Exception as exc:
                out["secondary_c3"]["variable_coefficients"] = {"error": repr(exc)}
            try:
                out["secondary_c3"]["action_n64"] = measure_c3_action(64, 10, 2)
            except Exception as exc:
                out["secondary_c3"]["action_n64"] = {"error": repr(exc)}
            try:
                out["secondary_c3"]["wider_descriptors"] = measure_tensor_forms(3, 1)
            except Exception as exc:
                out["secondary_c3"]["wider_descriptors"] = {"error": repr(exc)}
        guarded("secondary_c4", lambda: measure_c4(2048, max(3, args.steps // 2), 2, cpu_sample=512 if cs else 0))
        guarded("secondary_c5_share", lambda: poisson_line(args, ctx, 2, (107, 107, 107), "weak",
                                                             "one of the 8 partitions of BASELINE.json configs[4]", "lexicographic", "", False,
                                                             cpu=cs, cpu_reps=3))
        guarded("secondary_c1", lambda: measure_c1(200, 3, with_cpu=cs))
        # BASELINE configs[4] as written -- the whole 215^3 CG2 cube, 2.29e9 nonzeros -- on this one device: the N = 1 anchor of the
        # strong-scaling curve the N > 1 lines append as "strong_c5" (row starts are 64-bit, include/fdhip.h fd_nnz_t)
        n5 = args.n5 or 215
        guarded("strong_c5", lambda: poisson_line(args, ctx, 2, (n5, n5, n5), "strong",
                                                  "BASELINE.json configs[4]" + ("" if n5 == 215 else ", reduced"), args.numbering, "", False))
    if rank == 0 and world == 1 and args.traffic == "auto" and args.workload == "c2":
        # HBM bytes per launch of every kernel in the line, from ONE pair of rocprofv3 PMC passes of this command (FETCH_SIZE,
        # WRITE_SIZE) run as child processes now that nothing is being timed
        roofs = []

        def gather(d):
            if isinstance(d, dict):
                if "kernel" in d and "traffic" in d:
                    roofs.append(d)
                for v in d.values():
                    gather(v)
            elif isinstance(d, list):
                for v in d:
                    gather(v)
        gather(out)
        tail = ["--steps", "3", "--warmup", "1", "--cpu-sample", "0", "--n", str(args.n or 215), "--degree", str(args.degree or 1),
                "--tile", args.tile, "--numbering", args.numbering, "--only", args.only, "--traffic", "off", "--variants", "",
                "--workload", "c2"] + ([] if (args.secondary and args.only == "both") else ["--no-secondary"]) + (["--no-bcs"] if args.no_bcs else [])
        tr, out["traffic_meta"] = collect_traffic(tail, sorted({r["kernel"] for r in roofs}))
        for r in roofs:
            c = match_traffic(tr, r)
            if c:
                r["traffic"] = c["hbm_bytes_per_launch"]
                r["traffic_counters_kb"] = {"FETCH_SIZE": c["FETCH_SIZE_kb"], "WRITE_SIZE": c["WRITE_SIZE_kb"]}
                if r.get("algorithmic_bytes"):
                    r["traffic_over_algorithmic"] = r["traffic"] / r["algorithmic_bytes"]
    if rank == 0:
        out.setdefault("cpu_baseline", None)
        from firedrake_amd import compilation
        # wrappers this process had to compile (hipcc, ~0.35 s each, inside the first-call times) / found in the disk cache
        out["jit_compiles"] = dict(compilation.stats)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
