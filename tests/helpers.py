"""Test helpers: run the same parloop through the oracle (CPU restatement) for comparison."""
import numpy as np

import oracle
from firedrake_amd import op2
from firedrake_amd.parloop import DatLegacyArg, GlobalLegacyArg, MatLegacyArg


def oracle_run(kernel, iterset, *args, iteration_region=None, pass_layer_arg=False):
    """Execute ``op2.par_loop(kernel, iterset, *args)`` semantics on the CPU oracle using COPIES of the
    carriers' host data.  Returns a list with the post-loop array (or OracleCSR) per argument."""
    oargs, outs = [], []
    for a in args:
        if isinstance(a, DatLegacyArg):
            data = np.array(a.data.data_ro_with_halos, copy=True)
            m = a.map_
            perm = None
            mv = None
            off = None
            if m is not None:
                if isinstance(m, op2.PermutedMap):
                    perm = list(m.permutation)
                mv = m._base().values_with_halo
                off = m.offset if iterset._extruded else None
            oargs.append(oracle.ODat(data, int(a.access), mv, offset=off, perm=perm))
            outs.append(data)
        elif isinstance(a, GlobalLegacyArg):
            data = np.array(a.data.data_ro, copy=True)
            oargs.append(oracle.OGlobal(data, int(a.access)))
            outs.append(data)
        elif isinstance(a, MatLegacyArg):
            sp = a.data.sparsity
            rds, cds = sp.dsets
            pairs = []
            for r, c, _ in sp.rcmaps:
                if r.iterset._extruded:
                    pairs.append((r.values_with_halo, c.values_with_halo, r.iterset.layers - 1, r.offset, c.offset))
                else:
                    pairs.append((r.values_with_halo, c.values_with_halo))
            csr = oracle.build_sparsity(rds.set.total_size, cds.set.total_size, pairs, rbs=rds.cdim, cbs=cds.cdim,
                                        set_diag=sp._has_diagonal)
            rm, cm = a.maps
            lg = a.lgmaps or (None, None)
            oargs.append(oracle.OMat(csr, int(a.access), rm._base().values_with_halo, cm._base().values_with_halo,
                                     roffset=rm.offset if iterset._extruded else None,
                                     coffset=cm.offset if iterset._extruded else None,
                                     row_lgmap=None if lg[0] is None else np.ascontiguousarray(lg[0], dtype=np.int32),
                                     col_lgmap=None if lg[1] is None else np.ascontiguousarray(lg[1], dtype=np.int32)))
            outs.append(csr)
    subset = iterset.indices if isinstance(iterset, op2.Subset) else None
    layers = tuple(int(x) for x in iterset.layers_array[0]) if iterset._extruded else None
    reg = {None: oracle.ALL, op2.ALL: oracle.ALL, op2.ON_BOTTOM: oracle.ON_BOTTOM, op2.ON_TOP: oracle.ON_TOP,
           op2.ON_INTERIOR_FACETS: oracle.ON_INTERIOR_FACETS}[iteration_region]
    oracle.par_loop(kernel.code, kernel.name, 0, iterset.size, oargs, subset=subset, layers=layers,
                    iteration_region=reg, pass_layer_arg=pass_layer_arg)
    return outs


def structured_tri_mesh(nx, ny, seed=0, perturb=0.0):
    """P1 triangles on a grid: (coords (nv,2), cells (nc,3) int32)."""
    xs, ys = np.meshgrid(np.linspace(0, 1, nx + 1), np.linspace(0, 1, ny + 1), indexing="xy")
    coords = np.stack([xs.ravel(), ys.ravel()], axis=1)
    if perturb:
        rng = np.random.default_rng(seed)
        interior = ((coords > 1e-12) & (coords < 1 - 1e-12)).all(axis=1)      # keep |Omega| = 1
        coords[interior] += perturb * rng.uniform(-1, 1, size=(interior.sum(), 2)) / max(nx, ny)
    cells = []
    for j in range(ny):
        for i in range(nx):
            a = j * (nx + 1) + i
            b, c, d = a + 1, a + nx + 1, a + nx + 2
            cells += [(a, b, c), (b, d, c)]
    return coords, np.asarray(cells, dtype=np.int32)
