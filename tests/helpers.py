"""Test helpers: run the same parloop through the oracle (CPU restatement) for comparison."""
import numpy as np

import oracle
from firedrake_amd import op2
from firedrake_amd.parloop import DatLegacyArg, GlobalLegacyArg, MatLegacyArg


_REGION = {None: oracle.ALL, op2.ALL: oracle.ALL, op2.ON_BOTTOM: oracle.ON_BOTTOM, op2.ON_TOP: oracle.ON_TOP,
           op2.ON_INTERIOR_FACETS: oracle.ON_INTERIOR_FACETS}


def oracle_pattern(sp):
    """The CSR pattern of Sparsity ``sp`` built by the oracle (one contribution per map pair and iteration region)."""
    rds, cds = sp.dsets
    pairs = []
    for r, c, regions in sp._pairs:
        it = r.iterset
        if it._extruded:
            for reg in regions:
                nl = it.layers - 1 if it.constant_layers else np.asarray(it.layers_array, dtype=np.int32)
                pairs.append((r.values_with_halo, c.values_with_halo, nl, r.offset, c.offset,
                              r.offset_quotient, c.offset_quotient, bool(it._extruded_periodic), _REGION[reg]))
        else:
            pairs.append((r.values_with_halo, c.values_with_halo))
    return oracle.build_sparsity(rds.set.total_size, cds.set.total_size, pairs, rbs=rds.cdim, cbs=cds.cdim,
                                 set_diag=sp._has_diagonal)


def _odat(a, access, iterset):
    view = None
    if isinstance(a.data, op2.DatView):          # the oracle gets the parent's rows and the component to look at
        parent = a.data._parent
        data = np.array(parent.data_ro_with_halos, copy=True)
        view = int(np.ravel_multi_index(a.data.index, parent.dim))
    else:
        data = np.array(a.data.data_ro_with_halos, copy=True)
    m = a.map_
    perm = mv = off = quot = None
    if m is not None:
        if isinstance(m, op2.PermutedMap):
            perm = list(m.permutation)
        mv = m._base().values_with_halo
        off = m.offset if iterset._extruded else None
        quot = m.offset_quotient if iterset._extruded else None
    return oracle.ODat(data, int(access), mv, offset=off, perm=perm, offset_quotient=quot, view_index=view), data


def _omat(mat, maps, lgmaps, access, iterset, unroll=False):
    csr = oracle_pattern(mat.sparsity)
    rm, cm = maps
    lg = lgmaps or (None, None)
    ext = iterset._extruded
    return oracle.OMat(csr, int(access), rm._base().values_with_halo, cm._base().values_with_halo,
                       roffset=rm.offset if ext else None, coffset=cm.offset if ext else None,
                       row_lgmap=None if lg[0] is None else np.ascontiguousarray(lg[0], dtype=np.int32),
                       col_lgmap=None if lg[1] is None else np.ascontiguousarray(lg[1], dtype=np.int32),
                       unroll=unroll,
                       roffset_quotient=rm.offset_quotient if ext else None,
                       coffset_quotient=cm.offset_quotient if ext else None), csr


def oracle_run(kernel, iterset, *args, iteration_region=None, pass_layer_arg=False):
    """Execute ``op2.par_loop(kernel, iterset, *args)`` semantics on the CPU oracle using COPIES of the
    carriers' host data.  Returns a list with the post-loop array (or OracleCSR) per argument; mixed arguments
    give a list of arrays (MixedDat) or a list of lists of OracleCSRs (MixedMat)."""
    from firedrake_amd import parloop as _pl
    oargs, outs = [], []
    for a in args:
        if isinstance(a, getattr(_pl, "MixedDatLegacyArg", ())):
            parts = [_odat(p, a.access, iterset) for p in a.split]
            oargs.append(oracle.OMixedDat([p for p, _ in parts], int(a.access)))
            outs.append([d for _, d in parts])
        elif isinstance(a, getattr(_pl, "MixedMatLegacyArg", ())):
            blocks = [[_omat(b.data, b.maps, b.lgmaps, a.access, iterset) for b in row] for row in a.split]
            oargs.append(oracle.OMixedMat([[m for m, _ in row] for row in blocks], int(a.access)))
            outs.append([[c for _, c in row] for row in blocks])
        elif isinstance(a, DatLegacyArg):
            od, data = _odat(a, a.access, iterset)
            oargs.append(od)
            outs.append(data)
        elif isinstance(a, GlobalLegacyArg):
            data = np.array(a.data.data_ro, copy=True)
            oargs.append(oracle.OGlobal(data, int(a.access)))
            outs.append(data)
        elif isinstance(a, MatLegacyArg):
            om, csr = _omat(a.data, a.maps, a.lgmaps, a.access, iterset, unroll=getattr(a, "unroll_map", False))
            oargs.append(om)
            outs.append(csr)
    subset = iterset.indices if isinstance(iterset, op2.Subset) else None
    layers = None
    if iterset._extruded:
        la = iterset.layers_array
        layers = tuple(int(x) for x in la[0]) if iterset.constant_layers else np.asarray(la, dtype=np.int32)
    oracle.par_loop(kernel.code, kernel.name, 0, iterset.size, oargs, subset=subset, layers=layers,
                    iteration_region=_REGION[iteration_region], pass_layer_arg=pass_layer_arg,
                    periodic=bool(iterset._extruded and iterset._extruded_periodic))
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
