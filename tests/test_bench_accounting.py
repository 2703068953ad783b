"""CPU: the byte accounting behind ``roofline.achieved`` in bench.py is the one SURVEY.md 8(d) defines (every input array
read once, every output written once; the zeroing pass only where a separate pass exists) -- pinned to the figures quoted
there for config C2 -- and the C2 mesh sizes are the ones the scope table states."""
import importlib.util
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_c2_algorithmic_bytes_match_the_scope_table():
    b = _bench()
    N = 215
    ncell, nnode = 6 * N ** 3, (N + 1) ** 3
    nnz = nnode + 2 * (7 * N ** 3 + 9 * N ** 2 + 3 * N)          # vertices + 2 * edges
    assert (ncell, nnode, nnz) == (59630250, 10077696, 150048286)
    # residual with ONE coefficient: map 954 MB + coords 242 MB + u 81 MB + output 81 MB = 1.36 GB (+ zeroing 81 MB = 1.44 GB)
    r1 = b.algorithmic_bytes(ncell, 4, nnode, 3, 1)
    assert abs(r1 - 1.36e9) < 0.01e9 and r1 == ncell * 16 + nnode * 24 + nnode * 8 + nnode * 8
    assert abs(b.algorithmic_bytes(ncell, 4, nnode, 3, 1, zeroing=True) - 1.44e9) < 0.01e9
    # the benchmark's residual kernel reads two coefficients (u and f): + 81 MB; the memset is outside the kernel bracket
    assert b.algorithmic_bytes(ncell, 4, nnode, 3, 2) == r1 + nnode * 8 == 1437813408
    # Jacobian, strict (owner-computes-rows: complete rows are written once, no zeroing pass): 954 MB + 242 MB + 1.20 GB
    j = b.algorithmic_bytes(ncell, 4, nnode, 3, 0, nnz)
    assert j == 2396334992 and abs(j - 2.396e9) < 0.001e9
    # with the reference's separate zeroing pass: 3.60 GB
    assert b.algorithmic_bytes(ncell, 4, nnode, 3, 0, nnz, zeroing=True) == 3596721280
    assert b.HBM_PEAK_GBS == 8000.0


def test_cpu_baseline_owner_partition_matches_serial():
    """The node-partitioned OpenMP mode behind cpu_baseline.all_host_threads assembles exactly what one thread does."""
    import numpy as np
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import forms, mesh as fmesh
    m = fmesh.UnitCubeMesh(9, degrees=(1,), perturb=0.1, tile=(4, 4, 2))
    V = m.space(1)
    cm, nn, nc = V.cell_node_map.values_with_halo, V.node_set.total_size, m.cell_set.size
    coords = np.array(m.coordinates.data_ro_with_halos)
    u, f = np.sin(3 * V.node_points[:, 0]), np.cos(V.node_points[:, 1])
    kr, kj = forms.poisson_residual_kernel(3, 1), forms.poisson_jacobian_kernel(3, 1)
    out = []
    for kw in ({}, {"threads": "owner", "owner_partition": oracle.make_owner_partition(cm, nc, nn, 5)}):
        csr = oracle.build_sparsity(nn, nn, [(cm, cm)])
        r = np.zeros(nn)
        oracle.par_loop(kr.code, kr.name, 0, nc, [ODat(r, INC, cm), ODat(coords, READ, cm), ODat(u, READ, cm), ODat(f, READ, cm)], **kw)
        oracle.par_loop(kj.code, kj.name, 0, nc, [OMat(csr, INC, cm, cm), ODat(coords, READ, cm)], **kw)
        out.append((r, csr.values.copy()))
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-13 * np.abs(out[0][0]).max()
    assert np.abs(out[0][1] - out[1][1]).max() <= 1e-13 * np.abs(out[0][1]).max()


def test_a_wedged_profiler_pass_is_killed_and_reported(tmp_path, monkeypatch):
    """bench.collect_traffic runs rocprofv3 as a child: a pass that never returns must not hold the bench line back -- it is killed
    (with whatever it started) after PMC_PASS_TIMEOUT_S and the line carries traffic = null with the reason."""
    import shutil
    import time
    import bench
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/bash\nsleep 600 &\nwait\n")
    fake.chmod(0o755)
    monkeypatch.setattr(shutil, "which", lambda name: str(fake))
    monkeypatch.setattr(bench, "PMC_PASS_TIMEOUT_S", 1)
    t0 = time.time()
    traffic, meta = bench.collect_traffic(["--steps", "1"], ["k"])
    assert traffic is None and "killed" in meta["error"]
    assert time.time() - t0 < 30


def test_cpu_baseline_runs_on_a_given_mesh_with_a_given_pattern():
    """bench.cpu_baseline: the oracle timed on the mesh it is handed (the benchmark's own), the matrix pattern taken from the
    caller (the device-built CSR in bench.py; here the oracle's), the OpenMP team sized by min(affinity, cgroup quota)."""
    import oracle
    from firedrake_amd import mesh as fmesh
    bench = _bench()
    m = fmesh.UnitCubeMesh(6, degrees=(1,))
    cm = m.space(1).cell_node_map.values_with_halo
    nn = m.space(1).node_set.total_size
    csr = oracle.build_sparsity(nn, nn, [(cm, cm)])
    out = bench.cpu_baseline(m, 1, reps=1, pattern=(csr.rowptr, csr.colidx), label="test")
    assert out["cores"] == 1 and out["value"] > 0 and out["sparsity_build_s"] is None and "test" in out["sample"]
    n, aff, quota = bench.host_threads()
    assert out["all_host_threads"]["cores"] == n <= aff and (quota is None or n <= max(1, int(quota)))
    out2 = bench.cpu_baseline(m, 1, reps=1)
    assert out2["sparsity_build_s"] is not None


def test_kd_leaf_sizes_within_a_few_percent_are_one_request():
    """parloop.kd_leaf_size: the staged loop asks for leaves of round(1536 * nodes / cells) = 260 nodes at C2 size, the
    owner-computes-rows loop for 3840 // 15 = 256 rows -- both become 256, so one k-d order of the mesh serves both
    (parloop.kd_order_of); small leaves are left alone and nothing is ever rounded to zero."""
    from firedrake_amd.parloop import kd_leaf_size
    assert kd_leaf_size(260) == kd_leaf_size(256) == 256
    assert kd_leaf_size(round(1536 * 10077696 / 59630250)) == 256
    assert kd_leaf_size(100) == 100 and kd_leaf_size(1) == 1 and kd_leaf_size(0) == 1
    assert kd_leaf_size(128) == 128 and kd_leaf_size(143) == 128 and kd_leaf_size(145) == 160
    for v in range(128, 5000, 37):
        w = kd_leaf_size(v)
        assert w % 32 == 0 and abs(w - v) <= 16
