"""Device memory comes back when a problem goes: meshes, maps, plans, derived orders, sparsities, matrices, code objects' tables.
(Round 6, third session: a module-level table keyed by ``id(map)`` with the Map as its value kept every Map ever used in a parloop
alive -- 26 MB per problem of 384 k cells, found by building twelve problems in a row, profiles/r6s3_leak_probe.txt.  The reference's
caches hang on the objects they describe -- ``ObjectCached``, pyop2/caching.py:60-120 -- for the same reason.)"""
import ctypes
import gc

import numpy as np
import pytest

from firedrake_amd import _lib, forms, mesh as fmesh

pytestmark = pytest.mark.gpu


def _free_bytes():
    _lib.call("fd_device_sync")
    hip = ctypes.CDLL("libamdhip64.so")
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)) == 0
    return f.value


def _poisson(degree, n):
    def run():
        prob = forms.PoissonProblem(fmesh.UnitCubeMesh((n, n, n), degrees=(degree,), perturb=0.1, numbering="lexicographic"), degree, bcs=True)
        for _ in range(2):
            prob.assemble_residual()
            prob.assemble_jacobian()
        return float(np.abs(prob.r.data_ro).max())
    return run


def _poisson_graph():
    from firedrake_amd.graph import CapturedStep
    prob = forms.PoissonProblem(fmesh.UnitCubeMesh((20, 20, 20), degrees=(1,), perturb=0.1, numbering="lexicographic"), 1, bcs=True)

    def step():
        prob.assemble_residual()
        prob.assemble_jacobian()

    g = CapturedStep(step)
    for _ in range(3):
        g()
    return float(np.abs(prob.r.data_ro).max())        # (the download waits for the replays)


def _dg():
    prob = forms.DGAdvectionProblem(fmesh.make_quad_mesh(96, tile=(4, 4), perturb=0.1))
    for _ in range(2):
        prob.assemble_rhs()
    return float(np.abs(prob.L.data_ro).max())


def _q2_hex():
    prob = forms.HelmholtzHexProblem(fmesh.make_extruded_hex_mesh(10, 10, 2, perturb=0.1), bcs=True)
    for _ in range(2):
        prob.assemble_jacobian()
        prob.assemble_action()
    return float(np.abs(prob.y.data_ro).max())


@pytest.mark.parametrize("kind", ["p1", "p2", "dg", "q2_hex", "p1_graph"])
def test_device_memory_returns_when_a_problem_goes(kind):
    run = {"p1": _poisson(1, 24), "p2": _poisson(2, 12), "dg": _dg, "q2_hex": _q2_hex, "p1_graph": _poisson_graph}[kind]
    free = []
    for _ in range(4):
        assert run() > 0.0
        gc.collect()
        free.append(_free_bytes())
    # the first pass loads code objects and fills the kernel caches (per kernel, not per mesh); from then on nothing may stay behind
    assert min(free[1:]) >= free[1] - (1 << 20) and free[3] >= free[1] - (1 << 20), [f >> 20 for f in free]


@pytest.mark.parametrize("kind", ["p1", "p2", "dg", "q2_hex"])
def test_plans_and_tensors_go_with_the_last_reference(kind):
    """No reference cycle ties a Parloop to its geometry: with the cycle collector OFF, dropping a problem returns the device memory of
    its plans, derived orders, instance tables and tensors at once (the argument getters take the Parloop as an argument and are kept
    on it; closures over ``self`` stored in the geometry would make the memory wait for a full collection)."""
    gc.collect()
    run = {"p1": _poisson(1, 24), "p2": _poisson(2, 12), "dg": _dg, "q2_hex": _q2_hex}[kind]
    run()                                                # code objects, kernel caches
    gc.collect()
    before = _free_bytes()
    gc.disable()
    try:
        for _ in range(3):
            run()
        after = _free_bytes()
    finally:
        gc.enable()
    assert after >= before - (1 << 20), ((before - after) >> 20, "MB held by garbage the reference counts did not free")
