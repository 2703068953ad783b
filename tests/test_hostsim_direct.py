"""CPU: the DIRECT-mode wrappers emitted by firedrake_amd/codegen.py, compiled for the host against the
one-lane stand-in header (tests/hostsim/) and compared with the oracle.  Checks the generator's indexing
logic (maps, extruded offsets, iteration regions, subsets, permuted maps, lgmap masking, CSR search)
without a GPU; the -m gpu suite checks the same loops through the HIP build."""
import numpy as np
import pytest

from firedrake_amd import op2
import golden_kernels as gk
from helpers import oracle_run, structured_tri_mesh
from hostsim import run_direct
from mixed_cases import periodic_column_mesh


def _check(kernel, iterset, *args, **kw):
    pl = op2.LegacyParloop(kernel, iterset, *args, **kw)
    got = run_direct(pl)
    ref = oracle_run(kernel, iterset, *args, **kw)
    for g, r in zip(got, ref):
        if hasattr(r, "values"):
            assert np.array_equal(g.rowptr, r.rowptr) and np.array_equal(g.colidx, r.colidx)
            g, r = g.values, r.values
        if r.dtype.kind == "f":
            assert np.abs(g - r).max() <= 1e-12 * max(1.0, np.abs(r).max())
        else:
            assert np.array_equal(g, r)
    return got


def test_rhs_and_mass_triangles():
    coords, cells = structured_tri_mesh(5, 4, perturb=0.2)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x = op2.Dat(nodes ** 2, coords)
    f = op2.Dat(nodes, np.random.default_rng(1).standard_normal(len(coords)))
    b = op2.Dat(nodes)
    _check(op2.Kernel(gk.RHS_Q6, "rhs_q6"), ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    lg = np.arange(len(coords), dtype=np.int32)
    lg[[0, 3, 7]] = -1
    for lgmaps in (None, (lg, lg)):
        _check(op2.Kernel(gk.MASS_Q6, "mass_q6"), ele, mat(op2.INC, (m, m), lgmaps=lgmaps), x(op2.READ, m))


def test_vector_mass_unrolled_and_blocked():
    coords, cells = structured_tri_mesh(3, 3)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x = op2.Dat(nodes ** 2, coords)
    mat = op2.Mat(op2.Sparsity((nodes ** 2, nodes ** 2), [(m, m, None)]))
    _check(op2.Kernel(gk.MASS_VEC_AFFINE, "mass_vec_affine"), ele, mat(op2.INC, (m, m)), x(op2.READ, m))


def test_minmax_rw_subset_permuted_global():
    rng = np.random.default_rng(3)
    it, ind = op2.Set(40), op2.Set(17)
    mp = op2.Map(it, ind, 2, rng.integers(0, 17, size=(40, 2)))
    a = op2.Dat(ind, rng.integers(-50, 50, size=17), dtype=np.int32)
    b = op2.Dat(it, rng.integers(-50, 50, size=40), dtype=np.int32)
    _check(op2.Kernel("static void mx(int *a, int *b) { for (int i = 0; i < 2; ++i) a[i] = a[i] < *b ? *b : a[i]; }", "mx"),
           it, a(op2.MAX, mp), b(op2.READ))
    g = op2.Global(1, 0.0)
    d = op2.Dat(ind, rng.standard_normal(17))
    ss = op2.Subset(it, [1, 3, 5, 8, 13, 21, 34])
    _check(op2.Kernel("static void sm(double *g, double *d) { g[0] += d[0] - 2*d[1]; }", "sm"), ss, g(op2.INC), d(op2.READ, mp))
    m1 = op2.Map(op2.Set(5), op2.Set(20), 4, rng.permutation(20).reshape(5, 4))
    m2 = op2.PermutedMap(m1, [3, 2, 0, 1])
    d1 = op2.Dat(m1.toset, rng.integers(0, 99, size=20), dtype=np.int32)
    d2 = op2.Dat(m1.toset, dtype=np.int32)
    _check(op2.Kernel("void cp(int *to, const int *from) { for (int i = 0; i < 4; i++) to[i] = from[i]; }", "cp"),
           m1.iterset, d2(op2.WRITE, m2), d1(op2.READ, m1))


@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP, op2.ON_INTERIOR_FACETS])
def test_extruded_columns(region):
    rng = np.random.default_rng(5)
    nbase, L = 6, 5                     # 6 base cells, 5 node layers -> 4 cell layers
    base = op2.Set(nbase)
    ext = op2.ExtrudedSet(base, layers=L)
    nv = 8                               # base vertices; P1xP1 prism-like: 3 base vertices x 2 per cell
    nodes = op2.Set(nv * L)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    vals = np.concatenate([tri * L, tri * L + 1], axis=1).astype(np.int32)      # bottom + top vertex of layer 0
    cm = op2.Map(ext, nodes, 6, vals, offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nv * L, 2)))
    out = op2.Dat(nodes)
    nf = 2 if region == op2.ON_INTERIOR_FACETS else 1
    k = op2.Kernel("static void kk(double *o, const double *x) { for (int i = 0; i < %d; ++i) o[i] += x[2*i] + 0.5*x[2*i+1]; }" % (6 * nf), "kk")
    _check(k, ext, out(op2.INC, cm), x(op2.READ, cm), iteration_region=region)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, None)]))
    n = 6 * nf
    km = op2.Kernel("static void km(double *A, const double *x) { for (int i = 0; i < %d; ++i) for (int j = 0; j < %d; ++j) A[i*%d+j] += x[2*i]*x[2*j+1]; }" % (n, n, n), "km")
    if region in (None,):
        _check(km, ext, mat(op2.INC, (cm, cm)), x(op2.READ, cm), iteration_region=region)


@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP, op2.ON_INTERIOR_FACETS])
def test_periodic_extrusion(region):
    rng = np.random.default_rng(11)
    ncl = 4
    base, ext, nodes, cm = periodic_column_mesh(rng, ncl=ncl)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    out = op2.Dat(nodes)
    nf = 2 if region == op2.ON_INTERIOR_FACETS else 1
    k = op2.Kernel("static void kp(double *o, const double *x) { for (int i = 0; i < %d; ++i) o[i] += (i+1)*x[2*i] + 0.5*x[2*i+1]; }" % (6 * nf), "kp")
    got = _check(k, ext, out(op2.INC, cm), x(op2.READ, cm), iteration_region=region)
    # independent numpy restatement: the top cell's upper vertices are the column's level-0 vertices
    exp = np.zeros(nodes.size)
    xv = x.data_ro
    lay = {None: range(ncl), op2.ON_BOTTOM: [0], op2.ON_TOP: [ncl - 1], op2.ON_INTERIOR_FACETS: range(ncl)}[region]
    for e in range(base.size):
        for l in lay:
            for f in range(nf):
                for i in range(6):
                    lev = (l + f + (1 if i >= 3 else 0)) % ncl
                    n = (cm.values[e, i] // ncl) * ncl + lev
                    q = f * 6 + i
                    exp[n] += (q + 1) * xv[n, 0] + 0.5 * xv[n, 1]
    assert np.abs(got[0] - exp).max() < 1e-12
    # matrix + sparsity over the same region
    n = 6 * nf
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [region] if region is not None else None)]))
    km = op2.Kernel("static void kpm(double *A, const double *x) { for (int i = 0; i < %d; ++i) for (int j = 0; j < %d; ++j) A[i*%d+j] += x[2*i]*x[2*j+1]; }" % (n, n, n), "kpm")
    _check(km, ext, mat(op2.INC, (cm, cm)), x(op2.READ, cm), iteration_region=region)


def test_periodic_without_quotient_and_subset():
    """offset_quotient None: every entry wraps like (layer + k) % nl (builder.py:109-112)."""
    rng = np.random.default_rng(12)
    ncl = 3
    base = op2.Set(6)
    ext = op2.ExtrudedSet(base, layers=ncl + 1, extruded_periodic=True)
    dg = op2.Set(6 * ncl)
    cm = op2.Map(ext, dg, 1, np.arange(6) * ncl, offset=[1])
    d = op2.Dat(dg, rng.standard_normal(6 * ncl))
    g = op2.Global(1, 0.0)
    k = op2.Kernel("static void ks(double *g, const double *d) { g[0] += d[0] - 3*d[1]; }", "ks")
    _check(k, ext, g(op2.INC), d(op2.READ, cm), iteration_region=op2.ON_INTERIOR_FACETS)
    ss = op2.Subset(ext, [0, 2, 5])
    _check(k, ss, g(op2.INC), d(op2.READ, cm), iteration_region=op2.ON_INTERIOR_FACETS)


def test_extruded_interior_facet_sparsity_regions():
    rng = np.random.default_rng(13)
    nbase, L, nv = 4, 4, 6
    base = op2.Set(nbase)
    ext = op2.ExtrudedSet(base, layers=L)
    nodes = op2.Set(nv * L)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    vals = np.concatenate([tri * L, tri * L + 1], axis=1).astype(np.int32)
    cm = op2.Map(ext, nodes, 6, vals, offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nv * L, 2)))
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [op2.ALL, op2.ON_INTERIOR_FACETS])]))
    km = op2.Kernel("static void kif(double *A, const double *x) { for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) A[i*12+j] += x[2*i]*x[2*j+1]; }", "kif")
    _check(km, ext, mat(op2.INC, (cm, cm)), x(op2.READ, cm), iteration_region=op2.ON_INTERIOR_FACETS)


# ---- mixed spaces: the flattened loop (GlobalKernel.flattened) through the direct wrapper -------------------------
def _check_mixed(kernel, iterset, *args, **kw):
    """Host-sim result of the flattened loop against the oracle's native mixed packs."""
    pl = op2.LegacyParloop(kernel, iterset, *args, **kw)
    flat = iter(run_direct(pl))
    ref = oracle_run(kernel, iterset, *args, **kw)
    out = []
    for r in ref:
        if isinstance(r, list) and r and isinstance(r[0], list):          # MixedMat: rows of OracleCSR
            got = [[next(flat) for _ in row] for row in r]
            for grow, rrow in zip(got, r):
                for g, q in zip(grow, rrow):
                    assert np.array_equal(g.rowptr, q.rowptr) and np.array_equal(g.colidx, q.colidx)
                    assert np.abs(g.values - q.values).max() <= 1e-12 * max(1.0, np.abs(q.values).max())
        elif isinstance(r, list):                                          # MixedDat: list of arrays
            got = [next(flat) for _ in r]
            for g, q in zip(got, r):
                assert np.abs(g - q).max() <= 1e-12 * max(1.0, np.abs(q).max())
        else:
            got = next(flat)
            if hasattr(r, "values"):
                assert np.abs(got.values - r.values).max() <= 1e-12 * max(1.0, np.abs(r.values).max())
            else:
                assert np.abs(got - r).max() <= 1e-12 * max(1.0, np.abs(r).max())
        out.append(got)
    return out


def test_mixed_reference_goldens():
    """tests/pyop2/test_matrices.py:858-937 (TestMixedMatrices): data, kernels and expected blocks of the reference."""
    from mixed_cases import (ADDONE_MAT, ADDONE_RHS, ADDONE_RHS_VEC, LL, OD, rdata, reference_mixed_fixture)
    mset, mdat, mvdat, mmap, msparsity = reference_mixed_fixture()
    mat = op2.Mat(msparsity)
    got, _ = _check_mixed(op2.Kernel(ADDONE_MAT, "addone_mat"), mmap.iterset, mat(op2.INC, (mmap, mmap)), mdat(op2.READ, mmap))
    eps = 1e-12
    np.testing.assert_allclose(got[0][0].todense(), np.diag([1.0, 4.0, 9.0]), eps)
    np.testing.assert_allclose(got[0][1].todense(), OD, eps)
    np.testing.assert_allclose(got[1][0].todense(), OD.T, eps)
    np.testing.assert_allclose(got[1][1].todense(), LL, eps)
    dat = op2.MixedDat(mset)
    got, _ = _check_mixed(op2.Kernel(ADDONE_RHS, "addone_rhs"), mmap.iterset, dat(op2.INC, mmap), mdat(op2.READ, mmap))
    np.testing.assert_allclose(got[0], rdata(3), eps)
    np.testing.assert_allclose(got[1], [1.0, 4.0, 6.0, 4.0], eps)
    vdat = op2.MixedDat(mset ** 2)
    got, _ = _check_mixed(op2.Kernel(ADDONE_RHS_VEC, "addone_rhs_vec"), mmap.iterset, vdat(op2.INC, mmap), mvdat(op2.READ, mmap))
    np.testing.assert_allclose(got[0], np.kron(list(zip(rdata(3))), np.ones(2)), eps)
    np.testing.assert_allclose(got[1], np.kron(list(zip([1.0, 4.0, 6.0, 4.0])), np.ones(2)), eps)


@pytest.mark.parametrize("vdim", [1, 2])
def test_mixed_velocity_pressure(vdim):
    from mixed_cases import mixed_kernels, velocity_pressure_space
    ele, mset, mmap, x, vmap = velocity_pressure_space(4, 3, vdim)
    mds = op2.MixedDataSet(mset, (vdim, 1))
    sp = op2.Sparsity((mds, mds), {(i, j): [(rm, cm, None)] for i, rm in enumerate(mmap) for j, cm in enumerate(mmap)})
    mat = op2.Mat(sp)
    jac, res = mixed_kernels(vdim)
    _check_mixed(jac, ele, mat(op2.INC, (mmap, mmap)), x(op2.READ, vmap))
    rng = np.random.default_rng(4)
    w = op2.MixedDat([op2.Dat(ds, rng.standard_normal((ds.total_size,) + (() if ds.cdim == 1 else ds.dim))) for ds in mds])
    b = op2.MixedDat(mds)
    _check_mixed(res, ele, b(op2.INC, mmap), x(op2.READ, vmap), w(op2.READ, mmap))
    # boundary conditions on the velocity block: masked lgmaps, one (row, col) pair per block
    lgv = np.arange(mset[0].total_size, dtype=np.int32)
    lgv[[0, 2, 5]] = -1
    lgp = np.arange(mset[1].total_size, dtype=np.int32)
    lgmaps = [(lgv, lgv), (lgv, lgp), (lgp, lgv), (lgp, lgp)]
    _check_mixed(jac, ele, mat(op2.INC, (mmap, mmap), lgmaps=lgmaps), x(op2.READ, vmap))


# ---- variable layers (pyop2/types/set.py:326-337, codegen/builder.py:754-838) ---------------------------------------
def _variable_layer_columns(rng, nbase=7, nv=9, maxl=6):
    """Columns with their own [bottom, top) node levels; DoFs of a column are numbered from ITS bottom cell, so the
    map entry of a cell points at the cell's own bottom layer (layer - bottom = 0 there)."""
    bottom = rng.integers(0, 3, size=nbase)
    top = bottom + rng.integers(2, maxl, size=nbase)            # at least one cell per column
    layers = np.stack([bottom, top], axis=1).astype(np.int32)
    base = op2.Set(nbase)
    ext = op2.ExtrudedSet(base, layers)
    L = maxl + 3
    nodes = op2.Set(nv * L)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    vals = np.concatenate([tri * L, tri * L + 1], axis=1).astype(np.int32)
    return base, ext, nodes, op2.Map(ext, nodes, 6, vals, offset=[1] * 6), layers, L


@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP, op2.ON_INTERIOR_FACETS])
def test_variable_layers(region):
    rng = np.random.default_rng(21)
    base, ext, nodes, cm, layers, L = _variable_layer_columns(rng)
    assert not ext.constant_layers
    with pytest.raises(ValueError):
        ext.layers
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    out = op2.Dat(nodes)
    nf = 2 if region == op2.ON_INTERIOR_FACETS else 1
    k = op2.Kernel("static void kv(double *o, const double *x) { for (int i = 0; i < %d; ++i) o[i] += (i+1)*x[2*i] + 0.5*x[2*i+1]; }" % (6 * nf), "kv")
    got = _check(k, ext, out(op2.INC, cm), x(op2.READ, cm), iteration_region=region)
    exp = np.zeros(nodes.size)                                   # independent numpy restatement
    xv = x.data_ro
    for e in range(base.size):
        b, t = layers[e]
        ncl = t - 1 - b
        lay = {None: range(ncl), op2.ON_BOTTOM: range(0, 1), op2.ON_TOP: range(ncl - 1, ncl),
               op2.ON_INTERIOR_FACETS: range(ncl - 1)}[region]
        for l in lay:
            for f in range(nf):
                for i in range(6):
                    n = cm.values[e, i] + (l + f)
                    exp[n] += (f * 6 + i + 1) * xv[n, 0] + 0.5 * xv[n, 1]
    assert np.abs(got[0] - exp).max() < 1e-12
    # layer argument and subsets: the layers array is indexed by the entity, not by the position in the subset
    lay_out = op2.Dat(nodes, dtype=np.float64)
    kl = op2.Kernel("static void kl(double *o, int layer) { for (int i = 0; i < 6; ++i) o[i] = layer; }", "kl")
    ss = op2.Subset(ext, [1, 4, 5])
    got = _check(kl, ss, lay_out(op2.WRITE, cm), pass_layer_arg=True)
    for e in (1, 4, 5):
        b, t = layers[e]
        assert got[0][cm.values[e, 3] + (t - 2 - b)] == t - 2          # upper nodes of the top cell carry its layer


@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP, op2.ON_INTERIOR_FACETS])
def test_variable_layers_matrix(region):
    """Matrix assembly + sparsity over columns with their own layer ranges (sparsity.pyx:325-346)."""
    rng = np.random.default_rng(22)
    base, ext, nodes, cm, layers, L = _variable_layer_columns(rng)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    n = 6 * (2 if region == op2.ON_INTERIOR_FACETS else 1)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [region] if region is not None else None)]))
    km = op2.Kernel("static void kvm(double *A, const double *x) { for (int i = 0; i < %d; ++i) for (int j = 0; j < %d; ++j) A[i*%d+j] += x[2*i]*x[2*j+1] + 1.0; }" % (n, n, n), "kvm")
    got = _check(km, ext, mat(op2.INC, (cm, cm)), x(op2.READ, cm), iteration_region=region)[0]
    # the pattern holds the always-allocated diagonal plus exactly the rows some cell of the region inserts into
    rows = set()
    for e in range(base.size):
        b, t = layers[e]
        ncl = t - 1 - b
        lay = {None: range(ncl), op2.ON_BOTTOM: range(0, 1), op2.ON_TOP: range(ncl - 1, ncl),
               op2.ON_INTERIOR_FACETS: range(ncl - 1)}[region]
        nf = 2 if region == op2.ON_INTERIOR_FACETS else 1
        for l in lay:
            rows.update(int(cm.values[e, i] + l + f) for f in range(nf) for i in range(6))
    touched = {r for r in range(got.nrows) if got.rowptr[r + 1] - got.rowptr[r] > 1}
    assert touched == rows


# ---- DatView (pyop2/types/dat.py:714-805; tests/pyop2/test_dats.py:286-340) ------------------------------------------
def test_dat_view_host_semantics():
    s = op2.Set(5)
    vdat = op2.Dat(op2.DataSet(s, 2), np.zeros(2 * 5), dtype=op2.ScalarType)
    vdat.data[:, 0] = 3
    vdat.data[:, 1] = 4
    comp = op2.DatView(vdat, 1)
    comp.data[:] = 7
    assert not vdat.halo_valid and not comp.halo_valid
    assert all(comp.data == 7) and all(vdat.data[:, 0] == 3) and all(vdat.data[:, 1] == 7)
    comp.zero()
    assert vdat.halo_valid and comp.halo_valid
    assert all(comp.data == 0) and all(vdat.data[:, 0] == 3) and all(vdat.data[:, 1] == 0)
    v2 = op2.Dat(op2.DataSet(s, 2), np.zeros(2 * 5), dtype=op2.ScalarType)
    c2 = op2.DatView(v2, 1)
    assert v2.halo_valid and c2.halo_valid and v2.dat_version == 0 and c2.dat_version == 0
    c2.data_ro_with_halos
    assert v2.halo_valid and c2.halo_valid and v2.dat_version == 0
    c2.data_with_halos                                      # marks the parent's halo dirty and bumps its version
    assert not v2.halo_valid and not c2.halo_valid and v2.dat_version == 1 and c2.dat_version == 1
    assert c2.cdim == 1 and c2.dim == (1,) and c2.shape == (5,)
    from firedrake_amd import exceptions
    with pytest.raises(exceptions.IndexValueError):
        op2.DatView(v2, 2)


def test_dat_view_in_parloops():
    rng = np.random.default_rng(31)
    coords, cells = structured_tri_mesh(4, 3)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    v = op2.Dat(nodes ** 3, rng.standard_normal((len(coords), 3)))
    out = op2.Dat(nodes ** 2, rng.standard_normal((len(coords), 2)))
    # READ one component, INC into one component of another Dat
    k = op2.Kernel("static void kvw(double *o, const double *a) { for (int i = 0; i < 3; ++i) o[i] += 2.0*a[i] + i; }", "kvw")
    got = _check(k, ele, op2.DatView(out, 1)(op2.INC, m), op2.DatView(v, 2)(op2.READ, m))
    exp = out.data_ro.copy()
    for c in cells:
        for i in range(3):
            exp[c[i], 1] += 2.0 * v.data_ro[c[i], 2] + i
    assert np.abs(got[0] - exp).max() < 1e-13 and np.array_equal(got[0][:, 0], out.data_ro[:, 0])
    # direct loop over the nodes on a view: the kernel sees one value per node
    kd = op2.Kernel("static void kd(double *x, const double *y) { x[0] = 3.0*y[0]; }", "kd")
    got = _check(kd, nodes, op2.DatView(out, 0)(op2.WRITE), op2.DatView(v, 1)(op2.READ))
    assert np.allclose(got[0][:, 0], 3.0 * v.data_ro[:, 1]) and np.array_equal(got[0][:, 1], out.data_ro[:, 1])
    # a tensor-valued Dat: index (1, 0) of a (2, 2) row
    t = op2.Dat(nodes ** (2, 2), rng.standard_normal((len(coords), 2, 2)))
    g = op2.Global(1, 0.0)
    ks = op2.Kernel("static void ks(double *g, const double *a) { g[0] += a[0] + a[1] + a[2]; }", "ks")
    got = _check(ks, ele, g(op2.INC), op2.DatView(t, (1, 0))(op2.READ, m))
    assert abs(got[0][0] - t.data_ro[:, 1, 0][cells].sum()) < 1e-12


@pytest.fixture(params=[False, True], ids=["gathered", "plan-copies"])
def plan_copies(request, monkeypatch):
    """READ Dats staged by the in-kernel gather (null plan copy) or streamed from their plan-ordered copy (Parloop._plan_copy)"""
    import hostsim
    monkeypatch.setattr(hostsim, "PLAN_COPIES", request.param)
    return request.param


# ---- the STAGED wrapper (LDS staging, lane order, LDS reduction, flush) with one OS thread per lane ---------------------
@pytest.mark.parametrize("lane_strided", [1, 0])
def test_staged_wrapper_residuals_on_host(lane_strided, monkeypatch, plan_copies):
    """The hot residual path: staged wrappers of the P1 right-hand side (golden kernel) and of the benchmark's Poisson
    residuals (P1 and P2, compile-time LDS strides) against the oracle, on a mesh that spans several plan blocks."""
    from firedrake_amd import forms
    from firedrake_amd.configuration import configuration
    from hostsim import run_staged
    monkeypatch.setitem(configuration, "lane_strided", lane_strided)
    coords, cells = structured_tri_mesh(9, 8, perturb=0.2)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x = op2.Dat(nodes ** 2, coords)
    f = op2.Dat(nodes, np.random.default_rng(1).standard_normal(len(coords)))
    k = op2.Kernel(gk.RHS_Q6, "rhs_q6")
    args = (op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    got = run_staged(op2.LegacyParloop(k, ele, *args), epb=40)[0]
    ref = oracle_run(k, ele, *args)[0]
    assert np.abs(got - ref).max() <= 1e-13 * max(1.0, np.abs(ref).max())
    u = op2.Dat(nodes, np.random.default_rng(2).standard_normal(len(coords)))
    kr = forms.poisson_residual_kernel(2, 1)
    args = (op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), u(op2.READ, m), f(op2.READ, m))
    got = run_staged(op2.LegacyParloop(kr, ele, *args), epb=33)[0]
    ref = oracle_run(kr, ele, *args)[0]
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def test_staged_wrapper_p2_tets_and_global_reduction_on_host():
    from firedrake_amd import forms, mesh as fmesh
    from hostsim import run_staged
    mesh = fmesh.UnitCubeMesh(3, degrees=(1, 2), perturb=0.1)
    prob = forms.PoissonProblem(mesh, 2, bcs=False)
    pl = prob.res_loop                                             # arity-10 map + arity-4 coordinate map, 512 lanes
    got = run_staged(pl, epb=70)
    ref_args = [pa.data(acc, pa.map_) for pa, acc in zip(pl.arguments, pl.accesses)]
    ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *ref_args)[0]
    assert np.abs(got[0] - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # an indirect READ with a Global INC (block reduction) in the staged shape
    cm = mesh.space(1).cell_node_map
    d = mesh.space(1).dat(data=np.random.default_rng(3).standard_normal(mesh.space(1).node_set.total_size))
    g = op2.Global(1, 0.0)
    kg = op2.Kernel("static void kg(double *g, const double *d) { g[0] += d[0] - 2*d[1] + d[2]*d[3]; }", "kg")
    got = run_staged(op2.LegacyParloop(kg, mesh.cell_set, g(op2.INC), d(op2.READ, cm)), epb=50)[0]
    ref = oracle_run(kg, mesh.cell_set, op2.Global(1, 0.0)(op2.INC), d(op2.READ, cm))[0]
    assert abs(got[0] - ref[0]) <= 1e-11 * max(1.0, abs(ref[0]))


@pytest.mark.parametrize("bcs", [False, True])
def test_owner_computes_rows_wrapper_on_host(bcs, plan_copies):
    """The hot Jacobian path: the owner-computes-rows wrapper (instance lists, per-instance row-offset table, LDS row
    accumulators, complete-row flush, BC masking through the lgmaps) of the benchmark's P1 and P2 Poisson Jacobians
    against the oracle's MatSetValuesLocal."""
    from firedrake_amd import forms, mesh as fmesh
    from hostsim import run_ocr
    mesh = fmesh.UnitCubeMesh(3, degrees=(1, 2), perturb=0.1)
    for degree in (1, 2):
        prob = forms.PoissonProblem(mesh, degree, bcs=bcs)
        mat, pl = prob.jacobian()
        mpa = pl.arguments[0]
        got = run_ocr(pl, rows_per_block=17 if degree == 1 else 40)
        args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
        ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
        if degree == 1:                              # accumulate into existing values (no pending Mat.zero())
            got2 = run_ocr(pl, rows_per_block=17, zero_pending=False)
            assert np.abs(got2.values - (ref.values + 1.0)).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())
        # the same loop with ONE bit-packed record per instance (fd_ocr_pack_records) instead of the uint16 / uint8 index rows;
        # P1: one map on both sides, so the diagonal offsets come from the row nodes' words
        got3 = run_ocr(pl, rows_per_block=17 if degree == 1 else 40, records=True)
        assert np.abs(got3.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


def test_staged_wrapper_mixed_and_other_dtypes_on_host():
    """Staged wrappers beyond fp64 scalars: a flattened mixed residual (two maps, vector block) and INC into
    float32 / int32 / uint32 Dats with arity 27 and cdim 2 (LDS atomics of every staged dtype)."""
    from hostsim import run_staged
    from mixed_cases import mixed_kernels, velocity_pressure_space
    ele, mset, mmap, x, vmap = velocity_pressure_space(6, 5, 2)
    mds = op2.MixedDataSet(mset, (2, 1))
    rng = np.random.default_rng(4)
    w = op2.MixedDat([op2.Dat(ds, rng.standard_normal((ds.total_size,) + (() if ds.cdim == 1 else ds.dim))) for ds in mds])
    b = op2.MixedDat(mds)
    _, res = mixed_kernels(2)
    pl = op2.LegacyParloop(res, ele, b(op2.INC, mmap), x(op2.READ, vmap), w(op2.READ, mmap))
    got = run_staged(pl, epb=25)
    ref = oracle_run(res, ele, op2.MixedDat(mds)(op2.INC, mmap), x(op2.READ, vmap), w(op2.READ, mmap))[0]
    for g, r in zip(got[:2], ref):
        assert np.abs(g - r).max() <= 1e-12 * max(1.0, np.abs(r).max())
    it, tgt = op2.Set(300), op2.Set(90)
    m27 = op2.Map(it, tgt, 27, rng.integers(0, 90, size=(300, 27)))
    for dt, ct in ((np.float32, "float"), (np.int32, "int"), (np.uint32, "unsigned int")):
        src = op2.Dat(it, rng.integers(1, 9, size=300), dtype=dt)
        acc = op2.Dat(tgt ** 2, dtype=dt)
        k = op2.Kernel("static void acc27(%s *a, const %s *s) { for (int i = 0; i < 27; ++i) { a[2*i] += s[0]; a[2*i+1] += 2*s[0]; } }" % (ct, ct), "acc27")
        got = run_staged(op2.LegacyParloop(k, it, acc(op2.INC, m27), src(op2.READ)), epb=64)[0]
        ref = oracle_run(k, it, op2.Dat(tgt ** 2, dtype=dt)(op2.INC, m27), src(op2.READ))[0]
        assert np.array_equal(got, ref)


def test_zeroed_output_arguments_for_min_max():
    """requires_zeroed_output_arguments: MIN/MAX packs (Dat through a map, and Global) start from ZERO, not from the
    current values (pyop2/codegen/builder.py:276-279, 368-371, 855, 871).  The kernel below only ever raises its pack to
    the entity value when that beats what the pack holds: with gathered packs a large current value survives, with
    zeroed packs the result is max(current, max over entities of max(0, b))."""
    rng = np.random.default_rng(8)
    it, ind = op2.Set(30), op2.Set(11)
    mp = op2.Map(it, ind, 2, rng.integers(0, 11, size=(30, 2)))
    b = op2.Dat(it, rng.standard_normal(30))
    code = "static void mz(double *a, double *g, const double *b) { for (int i = 0; i < 2; ++i) a[i] = a[i] < *b ? *b : a[i]; g[0] = g[0] < *b ? *b : g[0]; }"
    for zeroed in (False, True):
        a = op2.Dat(ind, np.full(11, -5.0))
        g = op2.Global(1, -7.0)
        k = op2.Kernel(code, "mz", requires_zeroed_output_arguments=zeroed)
        got = _check(k, it, a(op2.MAX, mp), g(op2.MAX), b(op2.READ))
        bm = max(0.0, b.data_ro.max()) if zeroed else b.data_ro.max()
        assert got[1][0] == max(-7.0, bm)
        if zeroed:
            assert (got[0] >= 0.0).all()          # every touched node saw a zeroed pack (all 11 nodes are touched)


def test_mat_through_permuted_maps():
    """A Mat accessed through PermutedMaps: rows/columns are map[e][perm[i]] (MatPack goes through the map's indexed(),
    builder.py:144-176, 573-625)."""
    coords, cells = structured_tri_mesh(4, 3, perturb=0.2)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    pr, pc = op2.PermutedMap(m, [2, 0, 1]), op2.PermutedMap(m, [1, 2, 0])
    x = op2.Dat(nodes ** 2, coords)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    km = op2.Kernel("static void kpm(double *A, const double *x) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) "
                    "A[i*3+j] += (i+1)*x[2*i] + 10*(j+1)*x[2*j+1]; }", "kpm")
    _check(km, ele, mat(op2.INC, (pr, pc)), x(op2.READ, m))


@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP])
def test_staged_wrapper_on_extruded_columns_and_subsets_on_host(region):
    """The staged wrapper over a virtual iteration space: (column, layer) cells of an extruded set -- plans on the derived
    map ``map + offset*layer`` -- also under a Subset of the columns, with a direct (base-entity) READ argument and the
    layer argument; and a plain Subset of a non-extruded set.  Against the oracle."""
    from hostsim import run_staged
    rng = np.random.default_rng(21)
    nbase, L, nv = 70, 6, 30                    # 350 cells per sweep: more than the 256 lanes of a workgroup
    base = op2.Set(nbase)
    ext = op2.ExtrudedSet(base, layers=L)
    nodes = op2.Set(nv * L)
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * L, tri * L + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nv * L, 2)))
    w = op2.Dat(base, rng.standard_normal(nbase))                       # direct: addressed by the BASE entity (parloop.py:494-497)
    out = op2.Dat(nodes)
    k = op2.Kernel("static void ks(double *o, const double *x, const double *w, int layer) "
                   "{ for (int i = 0; i < 6; ++i) o[i] += (1 + layer) * w[0] * (x[2*i] + 0.5*x[2*i+1]); }", "ks")
    for iterset, epb in ((ext, 330), (op2.Subset(ext, [5, 1, 3, 6] + list(range(10, 70))), 300), (ext, 5)):
        args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
        pl = op2.LegacyParloop(k, iterset, *args, iteration_region=region, pass_layer_arg=True)
        got = run_staged(pl, epb=epb)[0]
        ref = oracle_run(k, iterset, *args, iteration_region=region, pass_layer_arg=True)[0]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # variable layers (set.py:326-337): every column its own [bottom, top); the map row of a column points at ITS bottom cell.  The
    # virtual space is ragged -- the derived map has one row per existing cell, the cell tables give the base entity and the layer
    bot = rng.integers(0, 3, nbase)
    top = bot + 1 + rng.integers(1, L - 2, nbase)              # node levels [bottom, top): 1 .. L-3 cells per column
    top[3] = bot[3] + 1                                         # (a column without any cell)
    extv = op2.ExtrudedSet(base, layers=np.stack([bot, top], axis=1))
    cmv = op2.Map(extv, nodes, 6, np.concatenate([tri * L + bot[:, None], tri * L + bot[:, None] + 1], axis=1).astype(np.int32), offset=[1] * 6)
    for iterset, epb in ((extv, 100), (op2.Subset(extv, [5, 1, 3, 6] + list(range(10, 70))), 37)):
        args = (out(op2.INC, cmv), x(op2.READ, cmv), w(op2.READ))
        pl = op2.LegacyParloop(k, iterset, *args, iteration_region=region, pass_layer_arg=True)
        from firedrake_amd.codegen import select_mode
        assert select_mode(pl.global_kernel) == "staged"
        got = run_staged(pl, epb=epb)[0]
        ref = oracle_run(k, iterset, *args, iteration_region=region, pass_layer_arg=True)[0]
        assert np.abs(ref).max() > 0 and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    if region is None:
        it, ind = op2.Set(900), op2.Set(190)
        mp = op2.Map(it, ind, 3, rng.integers(0, 190, size=(900, 3)))
        d, o2 = op2.Dat(ind, rng.standard_normal(190)), op2.Dat(ind)
        ws = op2.Dat(it, rng.standard_normal(900))
        ss = op2.Subset(it, rng.choice(900, 700, replace=False))
        k2 = op2.Kernel("static void k2(double *o, const double *d, const double *w) { for (int i = 0; i < 3; ++i) o[i] += w[0]*d[(i+1)%3]; }", "k2")
        pl = op2.LegacyParloop(k2, ss, o2(op2.INC, mp), d(op2.READ, mp), ws(op2.READ))
        got = run_staged(pl, epb=640)[0]
        ref = oracle_run(k2, ss, o2(op2.INC, mp), d(op2.READ, mp), ws(op2.READ))[0]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("numbering", ["lexicographic", "random"])
def test_staged_wrapper_over_a_locality_order_on_host(numbering):
    """"stagedo": the staged wrapper over a backend-derived entity order (the cells around the k-d leaves of the vertices,
    helpers.locality_order_ref = numpy restatement of fd_locality_order) -- plans on the gathered map rows, slot -> entity
    through fd_order_ for the direct argument.  The benchmark's P1 residual plus a direct READ Dat, against the oracle."""
    from firedrake_amd import forms, mesh as fmesh
    from helpers import locality_order_ref
    from hostsim import run_staged
    mesh = fmesh.UnitCubeMesh(5, degrees=(1,), perturb=0.1, numbering=numbering)
    V = mesh.space(1)
    cm = V.cell_node_map
    order, starts = locality_order_ref(cm.values_with_halo, 0, mesh.cell_set.size, np.array(mesh.coordinates.data_ro), target=96)
    assert sorted(order.tolist()) == list(range(mesh.cell_set.size)) and starts[-1] == mesh.cell_set.size
    assert np.diff(starts).max() <= (3 * 96) // 2
    prob = forms.PoissonProblem(mesh, 1, bcs=False)
    got = run_staged(prob.res_loop, epb=600, order=order)[0]         # > 256 lanes: lanes iterate (index-row prefetch path)
    ref = oracle_run(prob.kres, mesh.cell_set, prob.r(op2.INC, cm), mesh.coordinates(op2.READ, cm), prob.u(op2.READ, cm),
                     prob.f(op2.READ, cm))[0]
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    w = op2.Dat(mesh.cell_set, np.random.default_rng(2).standard_normal(mesh.cell_set.total_size))
    out = op2.Dat(V.node_set)
    k = op2.Kernel("static void kw(double *o, const double *x, const double *w) { for (int i = 0; i < 4; ++i) o[i] += w[0]*x[3*i+1]; }", "kw")
    pl = op2.LegacyParloop(k, mesh.cell_set, out(op2.INC, cm), mesh.coordinates(op2.READ, cm), w(op2.READ))
    got = run_staged(pl, epb=700, order=order)[0]
    ref = oracle_run(k, mesh.cell_set, out(op2.INC, cm), mesh.coordinates(op2.READ, cm), w(op2.READ))[0]
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


@pytest.mark.parametrize("bcs", [False, True])
@pytest.mark.parametrize("numbering", ["lexicographic", "random"])
def test_owner_computes_rows_over_a_row_order_on_host(bcs, numbering):
    """"ocrp": owner-computes-rows whose row blocks are ranges of row POSITIONS under a backend-derived row order (first
    touch under the Morton order of the cells): ownership and accumulator offsets through pinv / prowptr, complete rows
    flushed row by row.  P1 and P2 Jacobians against the oracle, including accumulation into existing values."""
    from firedrake_amd import forms, mesh as fmesh
    from helpers import locality_order_ref
    from hostsim import run_ocr
    mesh = fmesh.UnitCubeMesh(4, degrees=(1, 2), perturb=0.1, numbering=numbering)
    pos = np.array(mesh.coordinates.data_ro)
    order = locality_order_ref(mesh.coord_space.cell_node_map.values_with_halo, 0, mesh.cell_set.size, pos, target=96)[0]
    for degree, rpb in ((1, 19), (2, 45)):
        prob = forms.PoissonProblem(mesh, degree, bcs=bcs)
        mat, pl = prob.jacobian()
        mpa = pl.arguments[0]
        got = run_ocr(pl, rows_per_block=rpb, order=order)
        args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
        ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
        if degree == 1:
            got2 = run_ocr(pl, rows_per_block=rpb, zero_pending=False, order=order)
            assert np.abs(got2.values - (ref.values + 1.0)).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())
        # bit-packed records, fresh and accumulating
        for zp in (True, False):
            got4 = run_ocr(pl, rows_per_block=rpb, zero_pending=zp, order=order, records=True)
            assert np.abs(got4.values - (ref.values + (0.0 if zp else 1.0))).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())
        if bcs and degree == 1:
            # "ocrpm": the BC columns masked by the flush's place table (-2 - place) instead of a select per contribution -- entries in
            # masked columns are stored as 0.0 when the loop overwrites its rows (the value array starts as garbage) and left
            # alone when it accumulates
            for zp in (True, False):
                got5 = run_ocr(pl, rows_per_block=rpb, zero_pending=zp, order=order, records=True, flush_colmask=True)
                assert np.abs(got5.values - (ref.values + (0.0 if zp else 1.0))).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())


@pytest.mark.parametrize("bcs", [False, True])
def test_fixed_point_accumulation_on_host(bcs):
    """Mode suffix "_fx": the LDS accumulators hold CHECKED 64-bit fixed-point sums -- every contribution enters as the integer
    round(x S) through an INTEGER atomic add (fdw::fx_acc), the flush scales back (fdw::fx_get).  Every row block keeps its own
    scale S = 2^(50 - L) with the window [2^(L-6), 2^L) its largest contribution must fall into; a block without a scale, or whose
    maximum leaves the window, (re)does its rows with fp64 atomics and writes the record of its next launch.  First run: fp64
    blocks, records written; with them the result is the oracle's to 1e-12 max|A| in the caller's row order and in a derived one,
    fresh and accumulating, with records and the run-coded flush -- and bitwise independent of the instance order."""
    from firedrake_amd import forms, mesh as fmesh
    from firedrake_amd.configuration import configuration
    from helpers import locality_order_ref
    from hostsim import run_ocr
    mesh = fmesh.UnitCubeMesh(4, degrees=(1,), perturb=0.1, numbering="lexicographic")
    order = locality_order_ref(mesh.coord_space.cell_node_map.values_with_halo, 0, mesh.cell_set.size,
                               np.array(mesh.coordinates.data_ro), target=96)[0]
    prob = forms.PoissonProblem(mesh, 1, bcs=bcs)
    mat, pl = prob.jacobian()
    mpa = pl.arguments[0]
    args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
    ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
    vmax = np.abs(ref.values).max()
    H = int(configuration["ocr_fx_headroom"])
    for kw in ({}, {"records": True}, {"order": order}, {"order": order, "records": True}):
        # no scale yet: fp64 blocks, every block with contributions writes its record
        cal = run_ocr(pl, rows_per_block=19, fixed_point=True, **kw)
        nb = len(cal.fx_records)
        assert np.abs(cal.values - ref.values).max() <= 1e-12 * vmax, kw
        assert cal.fx_stat[0] == 0 and cal.fx_stat[1] == nb
        have = cal.fx_records["S"] > 0
        assert have.sum() >= nb - 2 and np.all(cal.fx_records["S"][have] == 2.0 ** (50 - cal.fx_records["L"][have]))
        assert 2.0 ** (cal.fx_records["L"][have].max() - H) >= vmax / 32          # (a contribution is of the order of the entries)
        got = run_ocr(pl, rows_per_block=19, fixed_point=cal.fx_records, **kw)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * vmax, kw
        assert got.fx_stat[0] == 0 and got.fx_stat[1] == nb - have.sum(), kw
        assert np.array_equal(got.fx_records, cal.fx_records)                       # in the window: kept
        got2 = run_ocr(pl, rows_per_block=19, zero_pending=False, fixed_point=cal.fx_records, **kw)
        assert np.abs(got2.values - (ref.values + 1.0)).max() <= 1e-12 * (1.0 + vmax), kw
    # the sums are exact integers: two different instance orders give bitwise the same matrix (one scale for all blocks)
    L = int(cal.fx_records["L"].max())
    a = run_ocr(pl, rows_per_block=19, fixed_point=L).values
    b = run_ocr(pl, rows_per_block=31, fixed_point=L, order=order).values
    assert np.array_equal(a, b)
    # scales the contributions have outgrown / that have become too coarse: the blocks fall back to fp64 -- same matrix, fallbacks
    # counted, records re-derived
    for bad in (L - H - 2, L + 12):
        for kw in ({}, {"order": order, "records": True}):
            low = run_ocr(pl, rows_per_block=19, fixed_point=bad, **kw)
            assert low.fx_stat[0] >= len(low.fx_records) - 2 and low.fx_stat[1] == 0
            assert np.abs(low.values - ref.values).max() <= 1e-12 * vmax, kw
            Ls = low.fx_records["L"][low.fx_records["S"] > 0]
            assert np.all((Ls <= L) | (Ls == bad)) and (Ls != bad).sum() >= len(Ls) - 2       # (a block without contributions keeps its record)
            low2 = run_ocr(pl, rows_per_block=19, zero_pending=False, fixed_point=bad, **kw)
            assert np.abs(low2.values - (ref.values + 1.0)).max() <= 1e-12 * (1.0 + vmax), kw


@pytest.mark.parametrize("bcs", [False, True])
@pytest.mark.parametrize("numbering", ["tiled", "random"])
def test_row_sliced_owner_computes_rows_on_host(bcs, numbering, plan_copies):
    """"ocrs" / "ocrsp": the row-sliced owner-computes-rows wrapper -- instances (entity, local row) grouped by local row and
    padded to whole wavefronts, one instantiation of the local kernel per row index, lgmaps folded into the per-instance
    slot / column-position tables, complete rows flushed contiguously (caller's row order) or row by row (backend-derived
    order).  P2 (the shape it is selected for) and P1 Jacobians against the oracle's MatSetValuesLocal, several trips per
    lane, plus accumulation into existing values."""
    from firedrake_amd import forms, mesh as fmesh
    from firedrake_amd.codegen import select_mode
    from helpers import locality_order_ref
    from hostsim import run_ocrs
    mesh = fmesh.UnitCubeMesh(3, degrees=(1, 2), perturb=0.1, numbering=numbering)
    order = None
    if numbering == "random":
        pos = np.array(mesh.coordinates.data_ro)
        order = locality_order_ref(mesh.coord_space.cell_node_map.values_with_halo, 0, mesh.cell_set.size, pos, target=96)[0]
    for degree, cap in ((2, 700), (2, 96), (1, 150)):
        prob = forms.PoissonProblem(mesh, degree, bcs=bcs)
        mat, pl = prob.jacobian()
        assert select_mode(pl.global_kernel) == ("ocrs" if degree == 2 else "ocr")
        mpa = pl.arguments[0]
        got = run_ocrs(pl, nnz_per_block=cap, order=order)
        args = [mat(op2.INC, mpa.maps, lgmaps=mpa.lgmaps)] + [pa.data(op2.READ, pa.map_) for pa in pl.arguments[1:]]
        ref = oracle_run(pl.global_kernel.local_kernel, pl.iterset, *args)[0]
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
        if cap == 96:
            got2 = run_ocrs(pl, nnz_per_block=cap, zero_pending=False, order=order)
            assert np.abs(got2.values - (ref.values + 1.0)).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())
            # one bit-packed record per instance (local map, column positions, slot; dropped = all ones) instead of three arrays
            got5 = run_ocrs(pl, nnz_per_block=cap, order=order, records=True, run_flush=order is not None)
            assert np.abs(got5.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
        if degree == 2:
            # two rows per instance ("_g" variants, fd_ocrplan_create_paired): one evaluation of the local kernel and one record for both
            # rows of a pair; a row another block owns is dropped by the instance (and added by that block's own instance)
            for zp in (True, False):
                got6 = run_ocrs(pl, nnz_per_block=cap, zero_pending=zp, order=order, records=True, run_flush=order is not None, pairs=True)
                assert np.abs(got6.values - (ref.values + (0.0 if zp else 1.0))).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())
            if order is not None:
                # the same blocks flushed through run-coded places ("ocrspr": one byte per entry, the block's displacements in LDS)
                # instead of row by row
                for zp in (True, False):
                    got4 = run_ocrs(pl, nnz_per_block=cap, zero_pending=zp, order=order, run_flush=True)
                    assert np.abs(got4.values - (ref.values + (0.0 if zp else 1.0))).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())


@pytest.mark.parametrize("numbering,bcs", [("tiled", False), ("tiled", True), ("random", True)])
def test_row_sliced_vector_valued_blocks_on_host(numbering, bcs):
    """MatSetValuesBlockedLocal through the row-sliced wrapper: vector P1 on tetrahedra (12x12 element matrices, 3x3 blocks;
    an instance owns the three scalar rows of one node, addressed through the node row's length), block lgmaps, contiguous
    and row-by-row flush."""
    from firedrake_amd import mesh as fmesh
    from firedrake_amd.codegen import select_mode
    from helpers import locality_order_ref
    from hostsim import run_ocrs
    from mixed_cases import vector_p1_elasticity_kernel
    mesh = fmesh.UnitCubeMesh(3, degrees=(1,), perturb=0.1, numbering=numbering)
    V = mesh.space(1)
    cm = V.cell_node_map
    mat = op2.Mat(op2.Sparsity((V.node_set ** 3, V.node_set ** 3), [(cm, cm, None)]))
    lg = None
    if bcs:
        lgv = np.arange(V.node_set.total_size, dtype=np.int32)
        lgv[np.random.default_rng(1).choice(V.node_set.size, V.node_set.size // 6, replace=False)] = -1
        lg = (lgv, lgv)
    k = vector_p1_elasticity_kernel(3)
    pl = op2.LegacyParloop(k, mesh.cell_set, mat(op2.INC, (cm, cm), lgmaps=lg), mesh.coordinates(op2.READ, cm))
    assert select_mode(pl.global_kernel) == "ocrs"
    order = None
    if numbering == "random":
        order = locality_order_ref(cm.values_with_halo, 0, mesh.cell_set.size, np.array(mesh.coordinates.data_ro), target=96)[0]
    ref = oracle_run(k, mesh.cell_set, mat(op2.INC, (cm, cm), lgmaps=lg), mesh.coordinates(op2.READ, cm))[0]
    for zero_pending in (True, False):
        got = run_ocrs(pl, nnz_per_block=60, zero_pending=zero_pending, order=order)
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - (ref.values + (0.0 if zero_pending else 1.0))).max() <= 1e-12 * (1.0 + np.abs(ref.values).max())


@pytest.mark.parametrize("region", ["all", "bottom", "top"])
@pytest.mark.parametrize("subset", [False, True])
def test_row_sliced_matrix_over_extruded_sets_and_subsets_on_host(region, subset):
    """Matrix loops over virtual iteration spaces through both owner-computes-rows wrappers (whole-entity and row-sliced
    instances): the plans are built on the derived maps (one row ``map + offset*layer`` per (column, layer) cell, resp. the
    subset's rows), the virtual id is decoded for the direct argument (base entity, parloop.py:494-497) and the layer
    argument only.  P1 prisms (arity 6) with a coordinate field, a cell-wise coefficient and pass_layer_arg, against the
    oracle."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_ocrs
    nb, layers = 7, 6
    rng = np.random.default_rng(8)
    base = op2.Set(nb)
    ext = op2.ExtrudedSet(base, layers=layers)
    nodes = op2.Set((nb + 2) * layers)
    vm = np.array([[i * layers, i * layers + 1, (i + 1) * layers, (i + 1) * layers + 1, (i + 2) * layers, (i + 2) * layers + 1]
                   for i in range(nb)], dtype=np.int32)
    m = op2.Map(ext, nodes, 6, vm, offset=[1] * 6)
    it = op2.Subset(ext, [0, 2, 3, 6]) if subset else ext
    x = op2.Dat(nodes ** 2, rng.uniform(0, 1, (nodes.total_size, 2)), np.float64)
    w = op2.Dat(ext, rng.uniform(1, 2, nb), np.float64)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.Kernel("""
static void prism(double *A, const double *x, const double *w, int layer)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      A[i*6 + j] += w[0] * (x[2*i] + 2.0 * x[2*j + 1]) + 0.25 * layer + (i == j ? 1.0 : 0.0);
}""", "prism")
    reg = {"all": None, "bottom": op2.ON_BOTTOM, "top": op2.ON_TOP}[region]
    kw = dict(iteration_region=reg, pass_layer_arg=True)
    pl = op2.LegacyParloop(k, it, mat(op2.INC, (m, m)), x(op2.READ, m), w(op2.READ), **kw)
    assert select_mode(pl.global_kernel) == "ocr"            # 6 rows: whole-entity instances, also on a virtual space
    ref = oracle_run(k, it, mat(op2.INC, (m, m)), x(op2.READ, m), w(op2.READ), **kw)[0]
    from hostsim import run_ocr
    from helpers import locality_order_ref
    # a backend-derived row order on the virtual space (Morton order of the (column, layer) cells -> first touch): what the
    # Parloop uses for un-hinted extruded maps, whose own row ranges are vertical pencils
    vrows = np.asarray(pl._plan_map(m, staged=True).values_with_halo)
    order = locality_order_ref(vrows, 0, len(vrows), np.array(x.data_ro), target=96)[0]
    for got in (run_ocrs(pl, nnz_per_block=150), run_ocr(pl, rows_per_block=9), run_ocrs(pl, nnz_per_block=150, order=order),
                run_ocr(pl, rows_per_block=9, order=order)):
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("region", ["all", "bottom", "top"])
def test_matrix_loops_over_variable_layers_on_host(region):
    """Variable layers (set.py:326-337: every column its own [bottom, top)) through both owner-computes-rows wrappers: the derived
    map holds one row per EXISTING cell, the direct argument and the layer argument are decoded through the cell tables.  P1
    prisms against the oracle (whose sparsity and loop walk the ragged columns, sparsity.pyx:328-371 / builder.py:754-831)."""
    from firedrake_amd.codegen import select_mode
    from firedrake_amd.configuration import configuration
    from hostsim import run_ocr, run_ocrs
    nb, L = 9, 7
    rng = np.random.default_rng(18)
    base = op2.Set(nb)
    bot = rng.integers(0, 2, nb)
    top = bot + 2 + rng.integers(0, L - 3, nb)
    ext = op2.ExtrudedSet(base, layers=np.stack([bot, top], axis=1))
    nodes = op2.Set((nb + 2) * L)
    vm = np.array([[i * L + bot[i], i * L + bot[i] + 1, (i + 1) * L + bot[i], (i + 1) * L + bot[i] + 1, (i + 2) * L + bot[i], (i + 2) * L + bot[i] + 1]
                   for i in range(nb)], dtype=np.int32)
    m = op2.Map(ext, nodes, 6, vm, offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.uniform(0, 1, (nodes.total_size, 2)), np.float64)
    w = op2.Dat(ext, rng.uniform(1, 2, nb), np.float64)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.Kernel("""
static void prismv(double *A, const double *x, const double *w, int layer)
{
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j)
      A[i*6 + j] += w[0] * (x[2*i] + 2.0 * x[2*j + 1]) + 0.25 * layer + (i == j ? 1.0 : 0.0);
}""", "prismv")
    reg = {"all": None, "bottom": op2.ON_BOTTOM, "top": op2.ON_TOP}[region]
    kw = dict(iteration_region=reg, pass_layer_arg=True)
    pl = op2.LegacyParloop(k, ext, mat(op2.INC, (m, m)), x(op2.READ, m), w(op2.READ), **kw)
    assert select_mode(pl.global_kernel) == "ocr"
    v = pl._virtual(staged=True)
    assert v.ragged and v.size(nb) == (int((top - 1 - bot).sum()) if region == "all" else nb)
    ref = oracle_run(k, ext, mat(op2.INC, (m, m)), x(op2.READ, m), w(op2.READ), **kw)[0]
    for got in (run_ocr(pl, rows_per_block=9), run_ocr(pl, rows_per_block=9, records=True), run_ocrs(pl, nnz_per_block=150),
                run_ocrs(pl, nnz_per_block=150, records=True)):
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("numbering", ["tiled", "random"])
def test_row_sliced_per_dof_lgmaps_on_host(numbering):
    """``unroll_map``: MatSetValuesLocal on dof indices with per-dof lgmaps (mat.py:700-716) -- component-wise Dirichlet
    conditions on a vector space: the x-component fixed on some nodes, the z-component on others.  The sliced wrapper carries
    them as an 8-bit row mask and a 64-bit column mask per instance."""
    from firedrake_amd import mesh as fmesh
    from firedrake_amd.codegen import select_mode
    from helpers import locality_order_ref
    from hostsim import run_ocrs
    from mixed_cases import vector_p1_elasticity_kernel
    mesh = fmesh.UnitCubeMesh(3, degrees=(1,), perturb=0.1, numbering=numbering)
    V = mesh.space(1)
    cm = V.cell_node_map
    mat = op2.Mat(op2.Sparsity((V.node_set ** 3, V.node_set ** 3), [(cm, cm, None)]))
    rng = np.random.default_rng(6)
    nn = V.node_set.total_size
    rlg, clg = np.arange(3 * nn, dtype=np.int32), np.arange(3 * nn, dtype=np.int32)
    fixed_x, fixed_z = rng.choice(nn, nn // 4, replace=False), rng.choice(nn, nn // 5, replace=False)
    rlg[3 * fixed_x] = -1
    rlg[3 * fixed_z + 2] = -1
    clg[3 * fixed_x] = -1
    clg[3 * rng.choice(nn, nn // 7, replace=False) + 1] = -1
    k = vector_p1_elasticity_kernel(3)
    args = lambda: (mat(op2.INC, (cm, cm), lgmaps=(rlg, clg), unroll_map=True), mesh.coordinates(op2.READ, cm))
    pl = op2.LegacyParloop(k, mesh.cell_set, *args())
    assert select_mode(pl.global_kernel) == "ocrs"
    order = None
    if numbering == "random":
        order = locality_order_ref(cm.values_with_halo, 0, mesh.cell_set.size, np.array(mesh.coordinates.data_ro), target=96)[0]
    ref = oracle_run(k, mesh.cell_set, *args())[0]
    got = run_ocrs(pl, nnz_per_block=60, order=order)
    assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
    assert np.count_nonzero(ref.values == 0.0) > 0.05 * len(ref.values)         # the masks dropped something


def test_row_sliced_instances_with_negative_map_entries_on_host():
    """Negative map entries (MatSetValuesLocal ignores negative indices, builder.py:573-625; composed maps with undefined
    intermediate entries): the row is never an instance, the column never a position.  (Whole-entity instances and block
    plans do not take such maps: Parloop._reject_negative_mat_maps demotes those loops to the direct wrapper.)"""
    from hostsim import run_direct, run_ocrs
    rng = np.random.default_rng(12)
    nn, ne, ar = 90, 70, 10
    mv = np.stack([rng.choice(nn, ar, replace=False) for _ in range(ne)]).astype(np.int32)
    mv[rng.random(mv.shape) < 0.15] = -1
    nodes, ele = op2.Set(nn), op2.Set(ne)
    m = op2.Map(ele, nodes, ar, mv)
    xs = op2.Dat(ele ** 2, rng.uniform(0, 1, (ne, 2)), np.float64)
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.Kernel("""
static void neg(double *A, const double *w)
{
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 10; ++j) A[i*10 + j] += w[0] * (i + 1) + w[1] * j;
}""", "neg")
    pl = op2.LegacyParloop(k, ele, mat(op2.INC, (m, m)), xs(op2.READ))
    ref = oracle_run(k, ele, mat(op2.INC, (m, m)), xs(op2.READ))[0]
    for got in (run_ocrs(pl, nnz_per_block=200), run_direct(pl)[0]):
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
    from firedrake_amd.parloop import PlanDoesNotFit
    with pytest.raises(PlanDoesNotFit):
        pl._reject_negative_mat_maps()


def test_rectangular_matrix_through_both_owner_computes_rows_shapes_on_host():
    """An off-diagonal block of a mixed operator: P2 rows x P1 columns on tetrahedra (10 x 4 element matrices, different
    row and column sets), with a column lgmap -- row-sliced and whole-entity instances against the oracle."""
    from firedrake_amd import mesh as fmesh
    from hostsim import run_ocr, run_ocrs
    mesh = fmesh.UnitCubeMesh(3, degrees=(1, 2), perturb=0.1)
    V2, V1 = mesh.space(2), mesh.space(1)
    rm, cm = V2.cell_node_map, V1.cell_node_map
    mat = op2.Mat(op2.Sparsity((V2.node_set ** 1, V1.node_set ** 1), [(rm, cm, None)]))
    rlg = np.arange(V2.node_set.total_size, dtype=np.int32)
    clg = np.arange(V1.node_set.total_size, dtype=np.int32)
    clg[np.random.default_rng(3).choice(len(clg), len(clg) // 5, replace=False)] = -1
    k = op2.Kernel("""
static void rect(double *A, const double *x)
{
  for (int i = 0; i < 10; ++i) for (int j = 0; j < 4; ++j) A[i*4 + j] += (1.0 + i) * x[3*j] - 0.5 * (j + 1) * x[3*(i % 4) + 2];
}""", "rect")
    args = lambda: (mat(op2.INC, (rm, cm), lgmaps=(rlg, clg)), mesh.coordinates(op2.READ, cm))
    pl = op2.LegacyParloop(k, mesh.cell_set, *args())
    ref = oracle_run(k, mesh.cell_set, *args())[0]
    for got in (run_ocrs(pl, nnz_per_block=120), run_ocr(pl, rows_per_block=23)):
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("seed", range(12))
def test_owner_computes_rows_shapes_on_random_loops(seed):
    """Randomised matrix loops through the row-sliced wrapper (and the whole-entity one where it applies) against the oracle:
    random arities, rectangular row / column maps on different sets, duplicate nodes inside an entity, block sizes, per-node
    or per-dof lgmaps, a direct and a mapped read-only argument, contiguous or row-ordered blocks, fresh or accumulating."""
    from hostsim import run_ocr, run_ocrs
    rng = np.random.default_rng(100 + seed)
    ar, ac = int(rng.integers(1, 9)), int(rng.integers(1, 7))
    rbs, cbs = (int(rng.integers(1, 4)), int(rng.integers(1, 3))) if seed % 3 == 0 else (1, 1)
    nr, nc, ne = int(rng.integers(20, 120)), int(rng.integers(15, 90)), int(rng.integers(40, 260))
    rows, cols, ele = op2.Set(nr), op2.Set(nc), op2.Set(ne)
    rm = op2.Map(ele, rows, ar, rng.integers(0, nr, (ne, ar)).astype(np.int32))          # duplicates inside a row allowed
    cm = op2.Map(ele, cols, ac, rng.integers(0, nc, (ne, ac)).astype(np.int32))
    w = op2.Dat(ele ** 2, rng.uniform(0.5, 1.5, (ne, 2)), np.float64)
    y = op2.Dat(cols ** 2, rng.uniform(-1, 1, (nc, 2)), np.float64)
    mat = op2.Mat(op2.Sparsity((rows ** rbs, cols ** cbs), [(rm, cm, None)]))
    R, C = ar * rbs, ac * cbs
    k = op2.Kernel(f"""
static void rnd{seed}(double *A, const double *w, const double *y)
{{
  for (int i = 0; i < {R}; ++i)
    for (int j = 0; j < {C}; ++j)
      A[i*{C} + j] += w[0] * (1.0 + i) + w[1] * y[2*(j / {cbs}) + (i & 1)] * (j + 1);
}}""", f"rnd{seed}")
    unroll = bool(seed % 4 == 1) and rbs <= 8 and C <= 64
    lgs = None
    if seed % 2 == 0 or unroll:
        nrd, ncd = (nr * rbs, nc * cbs) if unroll else (nr, nc)
        rlg, clg = np.arange(nrd, dtype=np.int32), np.arange(ncd, dtype=np.int32)
        rlg[rng.random(nrd) < 0.2] = -1
        clg[rng.random(ncd) < 0.25] = -1
        lgs = (rlg, clg)
    args = lambda: (mat(op2.INC, (rm, cm), lgmaps=lgs, unroll_map=unroll), w(op2.READ), y(op2.READ, cm))
    pl = op2.LegacyParloop(k, ele, *args())
    ref = oracle_run(k, ele, *args())[0]
    order = rng.permutation(ne).astype(np.int32) if seed % 5 == 2 else None
    zero = bool(seed % 3 != 1)
    tol = 1e-12 * (1.0 + np.abs(ref.values).max())
    got = run_ocrs(pl, nnz_per_block=int(rng.integers(30, 400)), zero_pending=zero, order=order)
    assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
    assert np.abs(got.values - (ref.values + (0.0 if zero else 1.0))).max() <= tol
    if rbs * cbs == 1 and not unroll:
        got = run_ocr(pl, rows_per_block=int(rng.integers(3, 40)), zero_pending=zero, order=order)
        assert np.abs(got.values - (ref.values + (0.0 if zero else 1.0))).max() <= tol
        got = run_ocr(pl, rows_per_block=int(rng.integers(3, 40)), zero_pending=zero, order=order, records=True)
        assert np.abs(got.values - (ref.values + (0.0 if zero else 1.0))).max() <= tol


@pytest.mark.parametrize("seed", range(10))
def test_dat_loops_on_random_maps_through_the_staged_and_direct_wrappers(seed):
    """Randomised Dat loops against the oracle: two maps of random arity onto different sets (duplicate nodes inside an entity
    allowed), INC outputs and READ inputs of random widths through either map, a direct read-only argument and a Global
    reduction, random plan-block sizes; the staged wrapper (LDS staging + LDS reduction + one global atomic per node) and the
    direct one must both agree with the sequential semantics."""
    from hostsim import run_direct, run_staged
    rng = np.random.default_rng(500 + seed)
    a1, a2 = int(rng.integers(1, 7)), int(rng.integers(1, 6))
    n1, n2, ne = int(rng.integers(10, 90)), int(rng.integers(8, 70)), int(rng.integers(30, 300))
    d_out, d_in1, d_in2 = int(rng.integers(1, 4)), int(rng.integers(1, 4)), int(rng.integers(1, 3))
    s1, s2, ele = op2.Set(n1), op2.Set(n2), op2.Set(ne)
    m1 = op2.Map(ele, s1, a1, rng.integers(0, n1, (ne, a1)).astype(np.int32))
    m2 = op2.Map(ele, s2, a2, rng.integers(0, n2, (ne, a2)).astype(np.int32))
    out1 = op2.Dat(s1 ** d_out if d_out > 1 else s1, rng.standard_normal((n1, d_out)) if d_out > 1 else rng.standard_normal(n1))
    out2 = op2.Dat(s2, rng.standard_normal(n2))
    in1 = op2.Dat(s1 ** d_in1 if d_in1 > 1 else s1, rng.standard_normal((n1, d_in1)) if d_in1 > 1 else rng.standard_normal(n1))
    in2 = op2.Dat(s2 ** d_in2 if d_in2 > 1 else s2, rng.standard_normal((n2, d_in2)) if d_in2 > 1 else rng.standard_normal(n2))
    w = op2.Dat(ele, rng.uniform(0.5, 1.5, ne))
    g = op2.Global(1, 0.25, np.float64)
    k = op2.Kernel(f"""
static void rdat{seed}(double *o1, double *o2, const double *i1, const double *i2, const double *w, double *g)
{{
  double s = 0.0;
  for (int i = 0; i < {a1 * d_in1}; ++i) s += (1.0 + 0.1*i) * i1[i];
  for (int i = 0; i < {a2 * d_in2}; ++i) s -= (0.5 + 0.2*i) * i2[i];
  for (int i = 0; i < {a1 * d_out}; ++i) o1[i] += w[0] * s * (i + 1);
  for (int i = 0; i < {a2}; ++i) o2[i] += w[0] + 0.01 * s * i;
  g[0] += w[0] * s;
}}""", f"rdat{seed}")
    args = lambda: (out1(op2.INC, m1), out2(op2.INC, m2), in1(op2.READ, m1), in2(op2.READ, m2), w(op2.READ), g(op2.INC))
    ref = oracle_run(k, ele, *args())
    tol = lambda r: 1e-11 * (1.0 + np.abs(r).max())
    for run in (lambda: run_staged(op2.LegacyParloop(k, ele, *args()), epb=int(rng.integers(16, 120))),
                lambda: run_direct(op2.LegacyParloop(k, ele, *args()))):
        got = run()
        for q in (0, 1, 5):
            assert np.abs(np.asarray(got[q]).reshape(-1) - np.asarray(ref[q]).reshape(-1)).max() <= tol(ref[q]), (seed, q)


@pytest.mark.parametrize("region", [None, op2.ON_BOTTOM, op2.ON_TOP])
@pytest.mark.parametrize("quotient", [True, False])
def test_periodic_columns_through_the_staged_and_owner_computes_rows_wrappers_on_host(region, quotient):
    """Periodic extrusion (builder.py:101-123: the layer offset wraps around the column) through the fast shapes: the wrap is
    folded into the rows of the derived map over the (column, layer) cells, so the staged wrapper (Dat loop, with the layer
    argument and a direct READ argument) and both owner-computes-rows wrappers (Mat loop) run it unchanged.  With the quotient
    table of a vertex-based space (the top cell's upper vertices are the column's level-0 vertices) and without one (a DG-like
    space: every entry wraps with the layer), against the oracle."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_ocr, run_ocrs, run_staged
    rng = np.random.default_rng(31)
    ncl = 5
    if quotient:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=40, ncl=ncl, nv=23)
        ar = 6
    else:
        base = op2.Set(40)
        ext = op2.ExtrudedSet(base, layers=ncl + 1, extruded_periodic=True)
        nodes = op2.Set(47 * ncl)
        pick = np.array([rng.choice(47, 3, replace=False) for _ in range(40)])
        cm = op2.Map(ext, nodes, 3, (pick * ncl).astype(np.int32), offset=[1, 2, 1])
        ar = 3
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.standard_normal(base.size))
    out = op2.Dat(nodes)
    k = op2.Kernel("static void kpw(double *o, const double *x, const double *w, int layer) { for (int i = 0; i < %d; ++i) "
                   "o[i] += (1 + layer) * w[0] * ((i+1)*x[2*i] + 0.5*x[2*i+1]); }" % ar, "kpw")
    kw = dict(iteration_region=region, pass_layer_arg=True)
    args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, ext, *args, **kw)
    assert select_mode(pl.global_kernel) == "staged"
    ref = oracle_run(k, ext, *args, **kw)[0]
    for epb in (64, 300):
        got = run_staged(pl, epb=epb)[0]
        assert np.abs(ref).max() > 0 and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [region] if region is not None else None)]))
    km = op2.Kernel("static void kpm2(double *A, const double *x, const double *w) { for (int i = 0; i < %d; ++i) for (int j = 0; j < %d; ++j) "
                    "A[i*%d+j] += w[0]*x[2*i]*x[2*j+1] + (i == j); }" % (ar, ar, ar), "kpm2")
    margs = (mat(op2.INC, (cm, cm)), x(op2.READ, cm), w(op2.READ))
    plm = op2.LegacyParloop(km, ext, *margs, iteration_region=region)
    assert select_mode(plm.global_kernel).startswith("ocr")
    mref = oracle_run(km, ext, *margs, iteration_region=region)[0]
    for got in (run_ocr(plm, rows_per_block=17), run_ocrs(plm, nnz_per_block=200), run_ocrs(plm, nnz_per_block=200, records=True)):
        assert np.array_equal(got.rowptr, mref.rowptr) and np.array_equal(got.colidx, mref.colidx)
        assert np.abs(got.values - mref.values).max() <= 1e-12 * np.abs(mref.values).max()


@pytest.mark.parametrize("subset", [False, True])
def test_write_min_max_through_maps_beside_staged_arguments_on_host(subset):
    """WRITE / MIN / MAX through a map in a loop whose READ arguments are staged: an interpolation writes its target through the
    cell-node map (every cell touching a node writes the same value: pyop2 WRITE semantics leave the winner open, builder.py:405-429),
    a limiter takes the minimum / maximum of cell values into its vertices.  The staged wrapper gathers the READ rows into LDS and lets
    the lane address the WRITE / MIN / MAX arguments in global memory as the direct wrapper does."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_staged
    rng = np.random.default_rng(41)
    nc, nn = 700, 260
    cells, nodes = op2.Set(nc), op2.Set(nn)
    cm = op2.Map(cells, nodes, 3, np.array([rng.choice(nn, 3, replace=False) for _ in range(nc)], dtype=np.int32))
    x = op2.Dat(nodes ** 2, rng.standard_normal((nn, 2)))
    cv = op2.Dat(cells, rng.standard_normal(nc))
    it = op2.Subset(cells, rng.choice(nc, 500, replace=False)) if subset else cells
    out = op2.Dat(nodes, np.full(nn, -7.0))
    kw = op2.Kernel("static void interp(double *o, const double *x) { for (int i = 0; i < 3; ++i) o[i] = 2.0*x[2*i] - x[2*i+1]*x[2*i+1]; }", "interp")
    pl = op2.LegacyParloop(kw, it, out(op2.WRITE, cm), x(op2.READ, cm))
    assert select_mode(pl.global_kernel) == "staged"
    got = run_staged(pl, epb=300)[0]
    ref = oracle_run(kw, it, out(op2.WRITE, cm), x(op2.READ, cm))[0]
    assert np.allclose(got, ref, rtol=1e-14, atol=0) and (ref != -7.0).sum() > 100
    lo, hi = op2.Dat(nodes, np.full(nn, 1e30)), op2.Dat(nodes, np.full(nn, -1e30))
    km = op2.Kernel("static void bounds(double *lo, double *hi, const double *c, const double *x) { for (int i = 0; i < 3; ++i) { "
                    "const double v = c[0] + 0.1*x[2*i]; if (v < lo[i]) lo[i] = v; if (v > hi[i]) hi[i] = v; } }", "bounds")
    args = (lo(op2.MIN, cm), hi(op2.MAX, cm), cv(op2.READ), x(op2.READ, cm))
    plm = op2.LegacyParloop(km, it, *args)
    assert select_mode(plm.global_kernel) == "staged"
    res = run_staged(plm, epb=200)
    refs = oracle_run(km, it, *args)
    # (to rounding: the oracle's compiler contracts c + 0.1 x into an FMA, the host-sim's does not)
    assert np.allclose(res[0], refs[0], rtol=1e-14, atol=0) and np.allclose(res[1], refs[1], rtol=1e-14, atol=0) and (refs[0] < 1e29).sum() > 100


@pytest.mark.parametrize("periodic", [False, True])
def test_interior_facets_of_extruded_columns_through_the_staged_wrapper_on_host(periodic):
    """ON_INTERIOR_FACETS on an extruded set (the horizontal facets between stacked cells: dS_h): the local kernel sees the nodes
    of the cell below and of the cell above (builder.py:94-124, f = 0, 1).  In the staged wrapper a row of the derived map holds
    both cells' nodes -- 2 x arity local indices per trip -- also across the seam of a periodic column."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_staged
    rng = np.random.default_rng(51)
    ncl = 5
    if periodic:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=40, ncl=ncl, nv=23)
    else:
        nv = 23
        base = op2.Set(40)
        ext = op2.ExtrudedSet(base, layers=ncl + 1)
        nodes = op2.Set(nv * (ncl + 1))
        tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(40)])
        cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (ncl + 1), tri * (ncl + 1) + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.standard_normal(base.size))
    out = op2.Dat(nodes)
    k = op2.Kernel("static void kfac(double *o, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                   "o[i] += (1 + layer) * w[0] * ((i+1)*x[2*i] + 0.5*x[2*((i+5)%12)+1]); }", "kfac")
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS, pass_layer_arg=True)
    args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, ext, *args, **kw)
    assert select_mode(pl.global_kernel) == "staged"
    v = pl._virtual(staged=True)
    assert v[0] == (ncl if periodic else ncl - 1)
    ref = oracle_run(k, ext, *args, **kw)[0]
    for epb in (37, 300):
        got = run_staged(pl, epb=epb)[0]
        assert np.abs(ref).max() > 0 and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    sub = op2.Subset(ext, [5, 1, 3, 6] + list(range(10, 40)))
    pls = op2.LegacyParloop(k, sub, *args, **kw)
    refs = oracle_run(k, sub, *args, **kw)[0]
    gots = run_staged(pls, epb=64)[0]
    assert np.abs(gots - refs).max() <= 1e-12 * max(1.0, np.abs(refs).max())


@pytest.mark.parametrize("periodic", [False, True])
def test_interior_facet_matrix_loop_through_the_row_sliced_wrapper_on_host(periodic):
    """The MATRIX of an interior-facet integral on an extruded set (dS_h: a 2a x 2a element tensor coupling the cell below and the
    cell above a horizontal facet, builder.py:573-625 with both maps doubled): the row-sliced owner-computes-rows wrapper sees the
    derived maps of the facets -- rows of 2 x arity nodes in the order of the kernel's packs -- as ordinary maps of twice the arity."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_ocrs
    rng = np.random.default_rng(61)
    ncl = 5
    if periodic:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=30, ncl=ncl, nv=19)
    else:
        nv = 19
        base = op2.Set(30)
        ext = op2.ExtrudedSet(base, layers=ncl + 1)
        nodes = op2.Set(nv * (ncl + 1))
        tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(30)])
        cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (ncl + 1), tri * (ncl + 1) + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.uniform(1, 2, base.size))
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [op2.ON_INTERIOR_FACETS])]))
    k = op2.Kernel("static void kfm(double *A, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                   "for (int j = 0; j < 12; ++j) A[i*12 + j] += w[0] * (x[2*i] * x[2*j+1] + 0.125 * layer) + (i == j ? 1.0 : 0.0) + (i < 6 && j >= 6 ? 0.5 : 0.0); }", "kfm")
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS, pass_layer_arg=True)
    args = (mat(op2.INC, (cm, cm)), x(op2.READ, cm), w(op2.READ))
    pl = op2.LegacyParloop(k, ext, *args, **kw)
    assert select_mode(pl.global_kernel).startswith("ocrs")
    ref = oracle_run(k, ext, *args, **kw)[0]
    for got in (run_ocrs(pl, nnz_per_block=260), run_ocrs(pl, nnz_per_block=260, records=True)):
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("region", [None, op2.ON_TOP])
@pytest.mark.parametrize("periodic", [False, True])
def test_write_and_max_through_maps_on_extruded_columns_on_host(region, periodic):
    """An interpolation on an extruded mesh writes its target through the (column, layer) addressing of builder.py:94-124: in the staged
    wrapper the READ arguments come from LDS (plans on the derived map) and the lane applies map + offset * layer -- with the periodic
    wrap -- to the BASE entity's map row for the WRITE / MAX arguments."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_staged
    rng = np.random.default_rng(71)
    ncl = 5
    if periodic:
        base, ext, nodes, cm = periodic_column_mesh(rng, nbase=40, ncl=ncl, nv=23)
    else:
        nv = 23
        base = op2.Set(40)
        ext = op2.ExtrudedSet(base, layers=ncl + 1)
        nodes = op2.Set(nv * (ncl + 1))
        tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(40)])
        cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (ncl + 1), tri * (ncl + 1) + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    out = op2.Dat(nodes, np.full(nodes.size, -7.0))
    hi = op2.Dat(nodes, np.full(nodes.size, -1e30))
    cv = op2.Dat(base, rng.standard_normal(base.size))
    k = op2.Kernel("static void interp_max(double *o, double *hi, const double *x, const double *c) { for (int i = 0; i < 6; ++i) { "
                   "o[i] = 2.0*x[2*i] - x[2*i+1]*x[2*i+1]; const double v = c[0] + x[2*i]; if (v > hi[i]) hi[i] = v; } }", "interp_max")
    for it in (ext, op2.Subset(ext, [5, 1, 3, 6] + list(range(10, 40)))):
        args = (out(op2.WRITE, cm), hi(op2.MAX, cm), x(op2.READ, cm), cv(op2.READ))
        pl = op2.LegacyParloop(k, it, *args, iteration_region=region)
        assert select_mode(pl.global_kernel) == "staged"
        res = run_staged(pl, epb=90)
        refs = oracle_run(k, it, *args, iteration_region=region)
        assert np.allclose(res[0], refs[0], rtol=1e-14, atol=0) and np.allclose(res[1], refs[1], rtol=1e-14, atol=0)
        assert (refs[0] != -7.0).sum() > 20 and (refs[1] > -1e29).sum() > 20


def test_interior_facets_of_variable_layer_columns_on_host():
    """Round 6: ON_INTERIOR_FACETS over columns of different heights through the staged wrapper (Dat loop) and the row-sliced
    owner-computes-rows wrapper (Mat loop): the ragged virtual space holds top_e - bottom_e - 2 facets per column (a one-cell column
    none), every derived map row the nodes of the cell below and of the cell above (builder.py:94-124, 754-776, 806-809)."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_ocrs, run_staged
    rng = np.random.default_rng(33)
    nbase, L, nv = 60, 7, 25
    base = op2.Set(nbase)
    bot = rng.integers(0, 3, nbase)
    top = bot + 2 + rng.integers(0, L - 3, nbase)
    top[:5] = bot[:5] + 2                                        # one cell: no interior facet
    ext = op2.ExtrudedSet(base, layers=np.stack([bot, top], axis=1))
    nodes = op2.Set(nv * (L + 2))
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (L + 2) + bot[:, None], tri * (L + 2) + bot[:, None] + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    w = op2.Dat(base, rng.uniform(1, 2, nbase))
    out = op2.Dat(nodes)
    kw = dict(iteration_region=op2.ON_INTERIOR_FACETS, pass_layer_arg=True)
    k = op2.Kernel("static void kfvh(double *o, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                   "o[i] += (1 + layer) * w[0] * ((i+1)*x[2*i] + 0.5*x[2*((i+5)%12)+1]); }", "kfvh")
    for it, epb in ((ext, 64), (op2.Subset(ext, [7, 2, 3] + list(range(12, 55))), 37)):
        args = (out(op2.INC, cm), x(op2.READ, cm), w(op2.READ))
        pl = op2.LegacyParloop(k, it, *args, **kw)
        assert select_mode(pl.global_kernel) == "staged"
        v = pl._virtual(staged=True)
        assert v.ragged and (it is not ext or v.size(nbase) == int(np.maximum(top - bot - 2, 0).sum()))
        got = run_staged(pl, epb=epb)[0]
        ref = oracle_run(k, it, *args, **kw)[0]
        assert np.abs(ref).max() > 0 and np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, [op2.ON_INTERIOR_FACETS])]))
    km = op2.Kernel("static void kfvhm(double *A, const double *x, const double *w, int layer) { for (int i = 0; i < 12; ++i) "
                    "for (int j = 0; j < 12; ++j) A[i*12 + j] += w[0] * (x[2*i] * x[2*j+1] + 0.125 * layer) + (i == j ? 1.0 : 0.0) + (i < 6 && j >= 6 ? 0.5 : 0.0); }", "kfvhm")
    margs = (mat(op2.INC, (cm, cm)), x(op2.READ, cm), w(op2.READ))
    plm = op2.LegacyParloop(km, ext, *margs, **kw)
    assert select_mode(plm.global_kernel) == "ocrs"
    ref = oracle_run(km, ext, *margs, **kw)[0]
    for got in (run_ocrs(plm, nnz_per_block=300), run_ocrs(plm, nnz_per_block=300, records=True)):
        assert np.array_equal(got.rowptr, ref.rowptr) and np.array_equal(got.colidx, ref.colidx)
        assert np.abs(got.values - ref.values).max() <= 1e-12 * np.abs(ref.values).max()


@pytest.mark.parametrize("shape", ["variable", "facets", "variable-facets"])
def test_write_and_max_through_maps_on_variable_layers_and_interior_facets_on_host(shape):
    """Round 6: WRITE / MAX arguments through maps beside staged READ arguments on the last two extruded shapes that demoted such a
    loop to the direct wrapper -- columns of VARIABLE height (the lane takes the column's own bottom from the layers array,
    builder.py:754-776) and INTERIOR FACETS (the lane walks both stacked cells of the facet, builder.py:94-124 with f = 0, 1)."""
    from firedrake_amd.codegen import select_mode
    from hostsim import run_staged
    rng = np.random.default_rng(83)
    nbase, L, nv = 50, 7, 21
    base = op2.Set(nbase)
    if shape.startswith("variable"):
        bot = rng.integers(0, 3, nbase)
        top = bot + 2 + rng.integers(0, L - 3, nbase)
        layers = np.stack([bot, top], axis=1)
    else:
        bot = np.zeros(nbase, dtype=np.int64)
        layers = L
    ext = op2.ExtrudedSet(base, layers=layers)
    nodes = op2.Set(nv * (L + 2))
    tri = np.array([rng.choice(nv, 3, replace=False) for _ in range(nbase)])
    cm = op2.Map(ext, nodes, 6, np.concatenate([tri * (L + 2) + bot[:, None], tri * (L + 2) + bot[:, None] + 1], axis=1).astype(np.int32), offset=[1] * 6)
    x = op2.Dat(nodes ** 2, rng.standard_normal((nodes.size, 2)))
    cv = op2.Dat(base, rng.standard_normal(nbase))
    facets = shape.endswith("facets")
    n = 12 if facets else 6
    k = op2.Kernel(f"static void interp_max_v(double *o, double *hi, const double *x, const double *c) {{ for (int i = 0; i < {n}; ++i) {{ "
                   "o[i] = 2.0*x[2*i] - x[2*i+1]*x[2*i+1]; const double v = c[0] + x[2*i]; if (v > hi[i]) hi[i] = v; } }", "interp_max_v")
    region = op2.ON_INTERIOR_FACETS if facets else None
    for it in (ext, op2.Subset(ext, [5, 1, 3, 6] + list(range(10, 45)))):
        out = op2.Dat(nodes, np.full(nodes.size, -7.0))
        hi = op2.Dat(nodes, np.full(nodes.size, -1e30))
        args = (out(op2.WRITE, cm), hi(op2.MAX, cm), x(op2.READ, cm), cv(op2.READ))
        pl = op2.LegacyParloop(k, it, *args, iteration_region=region)
        assert select_mode(pl.global_kernel) == "staged"
        res = run_staged(pl, epb=64)
        refs = oracle_run(k, it, *args, iteration_region=region)
        assert np.allclose(res[0], refs[0], rtol=1e-14, atol=0) and np.allclose(res[1], refs[1], rtol=1e-14, atol=0)
        assert (refs[0] != -7.0).sum() > 20 and (refs[1] > -1e29).sum() > 20


@pytest.mark.parametrize("stage_batch", [4, 2, 0])
def test_staged_wrapper_batched_staging_and_flush_on_host(stage_batch, monkeypatch, plan_copies):
    """Blocks with MORE nodes than lanes: a lane stages a batch of nodes (all node ids, then all rows, then LDS: clamped indices,
    unconditional stores) and preloads the node ids of the flush ahead of the main loop's last barrier; the index rows travel as raw
    words.  P1 residual (scalar INC) and a vector-valued INC Dat (cdim 2: the flush walks (node, component) pairs) against the
    oracle, with blocks whose node counts are not multiples of the lane count and a last block that is nearly empty."""
    import re
    from firedrake_amd import forms
    from firedrake_amd.codegen import generate_wrapper, mode_variant
    from firedrake_amd.configuration import configuration
    from hostsim import run_staged
    monkeypatch.setitem(configuration, "stage_batch", stage_batch)
    coords, cells = structured_tri_mesh(37, 29, perturb=0.2)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x = op2.Dat(nodes ** 2, coords)
    rng = np.random.default_rng(5)
    f = op2.Dat(nodes, rng.standard_normal(len(coords)))
    u = op2.Dat(nodes, rng.standard_normal(len(coords)))
    kr = forms.poisson_residual_kernel(2, 1)
    args = (op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), u(op2.READ, m), f(op2.READ, m))
    pl = op2.LegacyParloop(kr, ele, *args)
    src = generate_wrapper(pl.global_kernel, mode_variant("staged", 1, [700])).source
    assert ("i_b += 3*nthr" in src) == (stage_batch == 4) and ("i_b += 2*nthr" in src) == (stage_batch == 2)
    for epb in (1100, 2000):                      # ~600 / ~1050 nodes per block for 256 lanes
        got = run_staged(pl, epb=epb)[0]
        ref = oracle_run(kr, ele, *args)[0]
        assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    # even arity (raw index rows) and a vector-valued INC argument
    it = op2.Set(2500)
    mp = op2.Map(it, nodes, 4, rng.integers(0, len(coords), size=(2500, 4)))
    o2, d2 = op2.Dat(nodes ** 2), op2.Dat(nodes ** 2, rng.standard_normal((len(coords), 2)))
    k2 = op2.Kernel("static void k2v(double *o, const double *d) { for (int i = 0; i < 4; ++i) for (int c = 0; c < 2; ++c) "
                    "o[2*i + c] += d[2*((i+1)%4) + c] - 0.5*d[2*i + 1 - c]; }", "k2v")
    pl2 = op2.LegacyParloop(k2, it, o2(op2.INC, mp), d2(op2.READ, mp))
    got = run_staged(pl2, epb=1300)[0]
    ref = oracle_run(k2, it, o2(op2.INC, mp), d2(op2.READ, mp))[0]
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
