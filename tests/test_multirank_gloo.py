"""CPU: the N > 1 path with real processes (gloo backend, world_size 2 and 3): partitioned mesh,
halo exchange protocol, reverse SUM reduction inside frozen_halo, owner-computes-rows Jacobian,
Global all-reduce.  Mirrors how the reference tests its halos: real ranks on one machine
(tests/firedrake/regression, @pytest.mark.parallel(nprocs=2/3))."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


# slabs two cubes thick; slabs ONE cube thick (an owned plane is then shared with both neighbours: the reverse exchange
# combines two contributions into one node); 1 x 2 x 2 and 2 x 2 x 2 blocks (edge and corner nodes with 3 / 7 neighbours)
@pytest.mark.parametrize("world,n,partition", [(2, 4, "slabs"), (3, 6, "slabs"), (3, 3, "slabs"), (4, 4, "blocks"), (8, 4, "blocks")])
def test_partitioned_assembly_protocol(world, n, partition):
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker.py"), str(r), str(world), str(port), str(n), partition],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=600)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n" + "\n=====\n".join(o[-2500:] for o in outs)
        assert f"rank {r}/{world} ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,degree,partition,numbering", [(2, 6, 1, "slabs", "tiled"), (3, 6, 2, "slabs", "tiled"), (3, 3, 1, "slabs", "tiled"),
                                                                (4, 4, 2, "blocks", "tiled"), (8, 4, 1, "blocks", "tiled"),
                                                                (2, 8, 1, "slabs", "lexicographic"), (4, 6, 1, "blocks", "random"),
                                                                (2, 6, 2, "slabs", "lexicographic")])
def test_partitioned_assembly_on_one_gpu(world, n, degree, partition, numbering):
    """Production device path (HIP wrappers, device pack/unpack, owner-computes-rows) with W ranks sharing
    cuda:0; only the wire (gloo instead of RCCL) differs from the multi-GPU run."""
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker_gpu.py"), str(r), str(world), str(port), str(n), str(degree), "gloo", partition, numbering],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n" + "\n=====\n".join(o[-2500:] for o in outs)
        assert f"rank {r}/{world} ok" in out


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,degree,partition", [(2, 6, 1, "slabs"), (2, 6, 2, "slabs"), (4, 4, 1, "blocks"), (8, 4, 2, "blocks")])
def test_partitioned_assembly_over_rccl(world, n, degree, partition):
    """The real wire: one GPU per rank, backend nccl (= RCCL), halos through fd_halo_* / ncclSend / ncclRecv and Globals
    through ncclAllReduce (csrc/fd_comm.hip).  Needs >= 2 devices: skipped on a one-GPU box, exercised by the
    driver's multi-GPU node."""
    from firedrake_amd import _lib
    import ctypes
    ndev = ctypes.c_int()
    _lib.call("fd_device_count", ctypes.byref(ndev))
    if ndev.value < world:
        pytest.skip(f"needs {world} GPUs, found {ndev.value}")
    port = _free_port()
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "mp_worker_gpu.py"), str(r), str(world), str(port), str(n), str(degree), "nccl", partition],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=900)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out)
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} failed:\n" + "\n=====\n".join(o[-2500:] for o in outs)
        assert f"rank {r}/{world} ok" in out


def test_partition_overheads_match_the_meshes_they_describe():
    """mesh.partition_overheads (the slabs-versus-blocks figures bench.py reports at N > 1) against the meshes themselves."""
    from firedrake_amd import mesh as fmesh
    for part, nranks in (("slabs", 4), ("blocks", 8)):
        ov = fmesh.partition_overheads(8, nranks, part, 1)
        worst, halo = 0.0, 0
        for r in range(nranks):
            m = fmesh.UnitCubeMesh(8, degrees=(1,), rank=r, nranks=nranks, partition=part)
            own, tot = m.cell_set.size, m.cell_set.total_size
            worst = max(worst, (tot - own) / own)
            ns = m.space(1).node_set
            halo = max(halo, ns.total_size - ns.size)
        assert abs(ov["redundant_cell_fraction"] - worst) < 1e-12 and ov["max_halo_nodes"] == halo
    big = fmesh.partition_overheads(215, 8, "blocks", 2)
    assert big["grid"] == (2, 2, 2) and big["max_neighbours"] == 7 and 0.02 < big["redundant_cell_fraction"] < 0.03
