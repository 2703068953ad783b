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


def main(rank, world, port, n, degree):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from firedrake_amd import forms, mesh as fmesh
    serial = forms.PoissonProblem(fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(2, 2, 2), perturb=0.1), degree, bcs=True)
    part = forms.PoissonProblem(fmesh.UnitCubeMesh(n, degrees=(degree,), tile=(2, 2, 2), perturb=0.1, rank=rank, nranks=world),
                                degree, bcs=True)
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
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main(*[int(a) for a in sys.argv[1:6]])
