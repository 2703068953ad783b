"""The per-parloop tracing hook (firedrake_amd/profiling.py <-> pyop2/parloop.py:219-232, pyop2/profiling.py): region names
``Parloop_<iterset>_<kernel>``, flops = part.size * num_flops, roctx ranges pushed and popped in pairs."""
import numpy as np
import pytest

from firedrake_amd import _lib, op2, profiling
from firedrake_amd.configuration import configuration


def test_roctx_binding_and_region_pairs(monkeypatch):
    lib = _lib.load()
    assert lib.fd_trace_available() in (0, 1)
    monkeypatch.setitem(configuration, "trace", 1)
    calls = []
    if lib.fd_trace_available():
        assert lib.fd_trace_range_push(b"Parloop_set_k") == 0 and lib.fd_trace_range_pop() == 0
    profiling.reset()
    with profiling.timed_region("Parloop_cells_k"):
        profiling.log_flops("Parloop_cells_k", 10, 10 * 7)
        calls.append(1)
    with profiling.timed_region("Parloop_cells_k"):
        profiling.log_flops("Parloop_cells_k", 5, 5 * 7)
    assert profiling.summary() == {"Parloop_cells_k": {"calls": 2, "entities": 15, "flops": 105.0}}
    monkeypatch.setitem(configuration, "trace", 0)
    profiling.reset()
    with profiling.timed_region("x"):
        profiling.log_flops("x", 1, 1)
    assert profiling.summary() == {}


@pytest.mark.gpu
def test_parloops_log_their_flops_under_the_reference_region_names(monkeypatch):
    monkeypatch.setitem(configuration, "trace", 1)
    profiling.reset()
    nodes, ele = op2.Set(50, "nodes"), op2.Set(40, "cells")
    m = op2.Map(ele, nodes, 2, np.random.default_rng(0).integers(0, 50, (40, 2)).astype(np.int32))
    x, y = op2.Dat(nodes, np.ones(50)), op2.Dat(nodes)
    k = op2.Kernel("static void ax(double *y, const double *x) { y[0] += 2.0*x[1]; y[1] += 2.0*x[0]; }", "ax", flop_count=4)
    for _ in range(3):
        op2.par_loop(k, ele, y(op2.INC, m), x(op2.READ, m))
    s = profiling.summary()
    # (the reference's region name: f"Parloop_{iterset.name}_{global_kernel.name}", global_kernel.name = "wrap_" + the local kernel's)
    assert s["Parloop_cells_wrap_ax"] == {"calls": 3, "entities": 120, "flops": 480.0}, s
    assert np.allclose(y.data_ro.sum(), 3 * 2.0 * 80)
