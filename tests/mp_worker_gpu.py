"""Worker for the one-GPU two-rank test: both ranks drive cuda:0, gloo carries the packed halo buffers.
Everything except the RCCL transport itself is the production path: partitioned mesh, staged HIP wrappers,
device pack/unpack kernels, frozen-halo reverse reduction, owner-computes-rows Jacobian."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def point_key(pts):
    return [tuple(int(v) for v in np.round(p * 1e6)) for p in pts]


def main(rank, world, port, n, degree, backend="gloo", partition=None, numbering="tiled"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if numbering != "tiled":
        os.environ["FDHIP_LOCALITY_MIN"] = "64"       # no producer hints: the backend-derived blocks, also on these small meshes
    import torch
    import torch.distributed as dist
    if backend == "nccl":
        # one GPU per rank, the library's own RCCL communicator carries the halos (csrc/fd_comm.hip)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
        from firedrake_amd import _lib
        _lib.call("fd_set_device", rank)
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from firedrake_amd import forms, mesh as fmesh, op2
    serial = forms.PoissonProblem(fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(2, 2, 2), perturb=0.1), degree, bcs=True)
    part = forms.PoissonProblem(fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(2, 2, 2), perturb=0.1, rank=rank, nranks=world, partition=partition,
                                                   numbering=numbering), degree, bcs=True)
    rs = np.array(serial.assemble_residual().data_ro)
    As = serial.assemble_jacobian().toscipy().tocsr()
    for rep in range(2):
        part.u.halo_valid = False
        r = np.array(part.assemble_residual().data_ro)
        A = part.assemble_jacobian().toscipy().tocsr()
    Vs, V = serial.V, part.V
    lookup = {k: i for i, k in enumerate(point_key(Vs.node_points))}
    sidx = np.array([lookup[k] for k in point_key(V.node_points)])
    nown = V.node_set.size
    assert np.allclose(r, rs[sidx[:nown]], rtol=0, atol=1e-12 * max(1.0, np.abs(rs).max())), "residual"
    ref = As[sidx[:nown]][:, sidx].toarray()
    assert np.allclose(A[:nown].toarray(), ref, rtol=0, atol=1e-12 * np.abs(ref).max()), "jacobian"
    assert V.node_set.halo.wire == ("rccl" if backend == "nccl" else "host"), V.node_set.halo.wire
    # INC Global across ranks (MPI_Iallreduce of pyop2/parloop.py:411-442): every rank sums its owned nodes
    g = op2.Global(1, 5.0)
    k = op2.Kernel("static void gs(double *g, const double *x) { g[0] += x[0]; }", "gs")
    op2.par_loop(k, V.node_set, g(op2.INC), part.r(op2.READ))
    rs_owned = rs.copy()
    assert abs(g.data_ro[0] - (5.0 + rs_owned.sum())) <= 1e-9 * max(1.0, np.abs(rs_owned).sum()), (g.data_ro, rs_owned.sum())
    # integer MAX reduction through the halo (typed pack/unpack, identity fill of the ghost region by dtype)
    cm = V.cell_node_map
    cnt = op2.Dat(V.node_set, dtype=np.int32)
    kc = op2.Kernel("static void ct(int *c) { for (int i = 0; i < %d; ++i) c[i] += 1; }" % cm.arity, "ct")
    op2.par_loop(kc, part.mesh.cell_set, cnt(op2.INC, cm))
    cs = op2.Dat(Vs.node_set, dtype=np.int32)
    op2.par_loop(kc, serial.mesh.cell_set, cs(op2.INC, Vs.cell_node_map))
    assert np.array_equal(cnt.data_ro, cs.data_ro[sidx[:nown]]), "int32 INC through the halo"
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:6]], *sys.argv[6:9])
