"""Worker for tests/test_multirank_gloo.py: one process per rank, gloo backend, host-mode halos.
The wrapper kernel is replaced by the oracle here (no GPU in this container); what is under test
is everything around it: partitioner, core|owned|ghost layouts, halo lists, the begin/end exchange
protocol of Parloop/Dat, owner-computes-rows Jacobian masking and the Global all-reduce."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def point_key(pts):
    return [tuple(int(v) for v in np.round(p * 1e6)) for p in pts]


def main(rank, world, port, n, partition=None):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import forms, mesh as fmesh, op2
    from firedrake_amd.halo import allreduce_global, set_halo_factory
    from host_halo import HostHalo
    set_halo_factory(HostHalo)

    serial = fmesh.UnitCubeMesh(n, degrees=(1, 2), tile=(2, 2, 2), perturb=0.1)
    part = fmesh.UnitCubeMesh(n, degrees=(1, 2), tile=(2, 2, 2), perturb=0.1, rank=rank, nranks=world, partition=partition)
    fn = lambda p: np.sin(3 * p[:, 0]) * np.cos(2 * p[:, 1]) + 0.3 * p[:, 2]          # noqa: E731
    for deg in (1, 2):
        Vs, V = serial.space(deg), part.space(deg)
        nown, ntot = V.node_set.size, V.node_set.total_size
        # ---- forward halo: owners -> ghosts
        vals = fn(V.node_points)
        start = vals.copy()
        start[nown:] = np.nan
        u = op2.Dat(V.node_set, start)
        u.halo_valid = False
        u.global_to_local_begin(op2.READ)
        u.global_to_local_end(op2.READ)
        got = u.data_ro_with_halos
        lists = V.halo
        ghosts = np.concatenate([v for v in lists.recv.values()]) if lists.recv else np.zeros(0, int)
        assert np.allclose(got[ghosts], vals[ghosts]), f"rank {rank} deg {deg}: forward halo wrong"
        assert u.halo_valid
        # ---- residual with the Parloop protocol (oracle as the kernel)
        f = op2.Dat(V.node_set, (1 + V.node_points[:, 0]) * np.cos(V.node_points[:, 1]))
        r = op2.Dat(V.node_set)
        kr = forms.poisson_residual_kernel(3, deg)
        cm, xm = V.cell_node_map.values_with_halo, part.coord_space.cell_node_map.values_with_halo
        coords = np.array(part.coordinates.data_ro_with_halos)
        r.zero()
        with r.frozen_halo(op2.INC):
            host = r.data_with_halos          # ghost region was zeroed by fill_ghosts
            for (off, size) in (part.cell_set.core_part, part.cell_set.owned_part):
                if size:
                    oracle.par_loop(kr.code, kr.name, off, off + size,
                                    [ODat(host, INC, cm), ODat(coords, READ, xm), ODat(np.array(got), READ, cm),
                                     ODat(np.array(f.data_ro_with_halos), READ, cm)])
        # serial reference
        cs, xs = Vs.cell_node_map.values_with_halo, serial.coord_space.cell_node_map.values_with_halo
        rs = np.zeros(Vs.node_set.total_size)
        oracle.par_loop(kr.code, kr.name, 0, serial.cell_set.size,
                        [ODat(rs, INC, cs), ODat(np.array(serial.coordinates.data_ro_with_halos), READ, xs),
                         ODat(fn(Vs.node_points), READ, cs), ODat((1 + Vs.node_points[:, 0]) * np.cos(Vs.node_points[:, 1]), READ, cs)])
        # perturbed coordinates are a function of the unperturbed lattice point: match through those
        lookup = {k: i for i, k in enumerate(point_key(Vs.node_points))}
        sidx = np.array([lookup[k] for k in point_key(V.node_points)])
        assert np.allclose(r.data_ro, rs[sidx[:nown]], rtol=0, atol=1e-12 * np.abs(rs).max()), f"rank {rank} deg {deg}: residual"
        # ---- Jacobian, owner computes rows: owned + ghost cells, non-owned rows masked
        kj = forms.poisson_jacobian_kernel(3, deg)
        csr = oracle.build_sparsity(ntot, ntot, [(cm, cm)])
        rlg = np.arange(ntot, dtype=np.int32)
        rlg[nown:] = -1
        oracle.par_loop(kj.code, kj.name, 0, part.cell_set.total_size,
                        [OMat(csr, INC, cm, cm, row_lgmap=rlg, col_lgmap=None), ODat(coords, READ, xm)])
        A = csr.toscipy()
        cser = oracle.build_sparsity(len(rs), len(rs), [(cs, cs)])
        oracle.par_loop(kj.code, kj.name, 0, serial.cell_set.size,
                        [OMat(cser, INC, cs, cs), ODat(np.array(serial.coordinates.data_ro_with_halos), READ, xs)])
        As = cser.toscipy().tocsr()
        Aown = A[:nown].toarray()
        ref = np.zeros_like(Aown)
        ref[:, :] = As[sidx[:nown]][:, sidx].toarray()
        assert np.allclose(Aown, ref, rtol=0, atol=1e-12 * np.abs(ref).max()), f"rank {rank} deg {deg}: jacobian rows"
    # ---- Global INC all-reduce
    g = op2.Global(1, float(rank + 1))
    allreduce_global(g, op2.INC)
    assert g.data_ro[0] == world * (world + 1) / 2
    dist.barrier()
    dist.destroy_process_group()
    print(f"rank {rank}/{world} ok")


if __name__ == "__main__":
    main(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), *sys.argv[5:6])
