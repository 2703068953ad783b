"""CPU: the PyOP2-side seam (firedrake_amd/bridge.py).  PyOP2 cannot be imported here, so its descriptors are played
by stand-ins with the reference's class names and fields (pyop2/global_kernel.py:27-325, local_kernel.py:86-227):
the translated kernel must generate exactly the wrapper the natively built kernel generates."""
import types

import numpy as np
import pytest

from firedrake_amd import _lib, bridge, op2
from firedrake_amd.codegen import generate_wrapper


@pytest.fixture(autouse=True)
def _fresh_seam():
    """every test allocates and frees its own device buffers: addresses repeat between tests"""
    yield
    bridge.reset()


def _cls(name, **defaults):
    def init(self, **kw):
        for k, v in {**defaults, **kw}.items():
            setattr(self, k, v)
    return type(name, (), {"__init__": init})


# stand-ins named like the reference's classes
MapKernelArg = _cls("MapKernelArg", arity=1, offset=None, offset_quotient=None)
PermutedMapKernelArg = _cls("PermutedMapKernelArg")
DatKernelArg = _cls("DatKernelArg", dim=(), map_=None, index=None)
GlobalKernelArg = _cls("GlobalKernelArg", dim=(1,), double=False)
MatKernelArg = _cls("MatKernelArg", unroll=False)
MixedDatKernelArg = _cls("MixedDatKernelArg")
CStringLocalKernel = _cls("CStringLocalKernel", cpp=False, headers=(), flop_count=None, requires_zeroed_output_arguments=False)
GlobalKernel = _cls("GlobalKernel", _extruded=False, _extruded_periodic=False, _constant_layers=False, _subset=False,
                    _iteration_region=None, _pass_layer_arg=False)

RHS = "static void rhs(double *b, const double *x, const double *f) { for (int i = 0; i < 3; ++i) b[i] += x[2*i]*f[i]; }"


def test_translated_kernel_generates_the_native_wrapper():
    m = MapKernelArg(arity=3)
    lk = CStringLocalKernel(code=RHS, name="rhs", accesses=(4, 1, 1), dtypes=(np.float64,) * 3)
    ref_gk = GlobalKernel(local_kernel=lk, arguments=[DatKernelArg(dim=(), map_=m), DatKernelArg(dim=(2,), map_=m),
                                                      DatKernelArg(dim=(1,), map_=m)])
    fd = bridge.as_fd_global_kernel(ref_gk)
    assert fd.local_kernel.accesses == (op2.INC, op2.READ, op2.READ)
    assert fd.arguments[0].map_ is fd.arguments[1].map_                     # one map object -> one map pointer
    nodes, ele = op2.Set(4), op2.Set(2)
    mm = op2.Map(ele, nodes, 3, [0, 1, 2, 1, 2, 3])
    native = op2.LegacyParloop(op2.Kernel(RHS, "rhs"), ele, op2.Dat(nodes)(op2.INC, mm), op2.Dat(nodes ** 2)(op2.READ, mm),
                               op2.Dat(nodes)(op2.READ, mm)).global_kernel
    for mode in ("staged", "direct"):
        assert generate_wrapper(fd, mode).source == generate_wrapper(native, mode).source


def test_flags_permuted_maps_views_and_mixed():
    base = MapKernelArg(arity=4, offset=(1, 1, 1, 1), offset_quotient=(0, 0, 1, 1))
    pm = PermutedMapKernelArg(base_map=base, permutation=(3, 2, 0, 1))
    lk = CStringLocalKernel(code="static void k(int *a, const int *b, double *g) {}", name="k", accesses=(2, 1, 4),
                            dtypes=(np.int32, np.int32, np.float64), requires_zeroed_output_arguments=True)
    gk = GlobalKernel(local_kernel=lk, arguments=[DatKernelArg(dim=(1,), map_=pm), DatKernelArg(dim=(3,), map_=base, index=(2,)),
                                                  GlobalKernelArg(dim=(1,))],
                      _extruded=True, _extruded_periodic=True, _constant_layers=True, _subset=True, _iteration_region=3,
                      _pass_layer_arg=False)
    fd = bridge.as_fd_global_kernel(gk)
    assert fd._extruded and fd._extruded_periodic and fd._constant_layers and fd._subset
    assert fd._iteration_region == op2.ON_INTERIOR_FACETS
    assert fd.arguments[0].map_.base_map is fd.arguments[1].map_ and fd.arguments[0].map_.permutation == (3, 2, 0, 1)
    assert fd.arguments[1].index == (2,) and fd.arguments[1].map_.offset_quotient == (0, 0, 1, 1)
    assert fd.local_kernel.requires_zeroed_output_arguments
    src = generate_wrapper(fd, "direct")
    assert [d[0] for d in src.layout[:2]] == ["layers", "subset"]            # the reference's positional order
    m1, m2 = MapKernelArg(arity=1), MapKernelArg(arity=2)
    mixed = GlobalKernel(local_kernel=CStringLocalKernel(code="static void k(double *v, double *d) {}", name="k",
                                                         accesses=(4, 1), dtypes=(np.float64, np.float64)),
                         arguments=[MixedDatKernelArg(arguments=(DatKernelArg(dim=(1,), map_=m1), DatKernelArg(dim=(1,), map_=m2))),
                                    MixedDatKernelArg(arguments=(DatKernelArg(dim=(1,), map_=m1), DatKernelArg(dim=(1,), map_=m2)))])
    fdm = bridge.as_fd_global_kernel(mixed)
    assert fdm.is_mixed and len(fdm.flattened().arguments) == 4
    with pytest.raises(TypeError):
        bridge.as_fd_global_kernel(GlobalKernel(local_kernel=lk, arguments=[types.SimpleNamespace()] * 3))


def test_function_level_seam_selects_the_fast_paths():
    """The seam picks the same wrapper shapes as the native Parloop: staged for indirect Dat loops, owner-computes-rows
    for a scalar INC matrix; a Mat slot must hold a DeviceMat handle."""
    from firedrake_amd import _lib
    m = MapKernelArg(arity=3)
    lk = CStringLocalKernel(code="static void k(double *A, const double *x) {}", name="k", accesses=(4, 1), dtypes=(np.float64,) * 2)
    gk = GlobalKernel(local_kernel=lk, arguments=[MatKernelArg(dims=((1, 1),), maps=(m, m)), DatKernelArg(dim=(2,), map_=m)])
    assert bridge.as_fd_global_kernel(gk).arguments[0].dims == (1, 1)
    func = bridge.compile_global_kernel_hip(gk)
    assert func.mode == "ocr"
    with pytest.raises(ValueError):
        func(0, 2, 1, 2)                                     # Mat handle + Dat pointer + one map pointer are expected
    with pytest.raises(ValueError):
        func(0, 2, 12345, 0, 0)                              # not a DeviceMat handle
    lk2 = CStringLocalKernel(code=RHS, name="rhs", accesses=(4, 1, 1), dtypes=(np.float64,) * 3)
    gk2 = GlobalKernel(local_kernel=lk2, arguments=[DatKernelArg(dim=(), map_=m), DatKernelArg(dim=(2,), map_=m),
                                                    DatKernelArg(dim=(1,), map_=m)])
    func2 = bridge.compile_global_kernel_hip(gk2)
    assert func2.mode == "staged"
    if not _lib.gpu_available():
        with pytest.raises(_lib.FDHipError):                 # no device, no fallback
            func2(0, 2, 0, 0, 0, 0)
    direct = GlobalKernel(local_kernel=CStringLocalKernel(code="static void d(double *a) { a[0] = 1.0; }", name="d", accesses=(2,),
                                                          dtypes=(np.float64,)), arguments=[DatKernelArg(dim=(1,))])
    assert bridge.compile_global_kernel_hip(direct).mode == "direct"


@pytest.mark.gpu
def test_function_level_seam_on_device():
    """func(start, end, *arglist) with device pointers in the reference's positional order."""
    from firedrake_amd.device import DeviceBuffer
    rng = np.random.default_rng(0)
    nn, ne = 500, 900
    cells = rng.integers(0, nn, size=(ne, 3)).astype(np.int32)
    x, f = rng.standard_normal((nn, 2)), rng.standard_normal(nn)
    m = MapKernelArg(arity=3)
    lk = CStringLocalKernel(code=RHS, name="rhs", accesses=(4, 1, 1), dtypes=(np.float64,) * 3)
    gk = GlobalKernel(local_kernel=lk, arguments=[DatKernelArg(dim=(), map_=m), DatKernelArg(dim=(2,), map_=m),
                                                  DatKernelArg(dim=(1,), map_=m)])
    func = bridge.compile_global_kernel_hip(gk)
    assert func.mode == "staged"
    b_d, x_d, f_d, m_d = (DeviceBuffer.from_numpy(a) for a in (np.zeros(nn), x, f, cells))
    func(0, ne, b_d.ptr, x_d.ptr, f_d.ptr, m_d.ptr)
    func(0, ne // 2, b_d.ptr, x_d.ptr, f_d.ptr, m_d.ptr)            # a second, partial range accumulates
    exp = np.zeros(nn)
    for rng_ in (range(ne), range(ne // 2)):
        for e in rng_:
            for i in range(3):
                exp[cells[e, i]] += x[cells[e, i], 0] * f[cells[e, i]]
    got = b_d.download(np.float64, (nn,))
    assert np.abs(got - exp).max() < 1e-12 * max(1.0, np.abs(exp).max())


@pytest.mark.gpu
def test_seam_func_sees_buffers_rewritten_between_calls():
    """The owner of a borrowed buffer (PyOP2, PETSc, another seam func) may rewrite it between calls -- the Newton state fed
    to a residual: from the third call on nothing cached on the VALUES of a READ argument (plan-ordered copies) may be
    served.  The same for a native Dat written through a live view without the carrier hearing of it."""
    from firedrake_amd.device import DeviceBuffer
    rng = np.random.default_rng(1)
    nn, ne = 700, 2000
    cells = rng.integers(0, nn, size=(ne, 3)).astype(np.int32)
    x = rng.standard_normal((nn, 2))
    m = MapKernelArg(arity=3)
    lk = CStringLocalKernel(code=RHS, name="rhs", accesses=(4, 1, 1), dtypes=(np.float64,) * 3)
    gk = GlobalKernel(local_kernel=lk, arguments=[DatKernelArg(dim=(), map_=m), DatKernelArg(dim=(2,), map_=m),
                                                  DatKernelArg(dim=(1,), map_=m)])
    func = bridge.compile_global_kernel_hip(gk)
    x_d, m_d, f_d, b_d = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(cells), DeviceBuffer(nn * 8), DeviceBuffer(nn * 8)
    for it in range(5):
        f = rng.standard_normal(nn)
        f_d.upload(f)                                   # same pointer, new values
        b_d.upload(np.zeros(nn))
        func(0, ne, b_d.ptr, x_d.ptr, f_d.ptr, m_d.ptr)
        exp = np.zeros(nn)
        np.add.at(exp, cells.ravel(), (x[:, 0] * f)[cells.ravel()])
        assert np.abs(b_d.download(np.float64, (nn,)) - exp).max() < 1e-12 * max(1.0, np.abs(exp).max()), it
    nodes, ele = op2.Set(nn), op2.Set(ne)
    mm = op2.Map(ele, nodes, 3, cells)
    xd, fd_, bd = op2.Dat(nodes ** 2, x), op2.Dat(nodes, np.zeros(nn)), op2.Dat(nodes)
    pl = op2.LegacyParloop(op2.Kernel(RHS, "rhs"), ele, bd(op2.INC, mm), xd(op2.READ, mm), fd_(op2.READ, mm))
    view = fd_.data                                     # a writable view kept alive across the calls
    for it in range(5):
        view[:] = rng.standard_normal(nn)
        bd.zero()
        pl.compute()
        exp = np.zeros(nn)
        np.add.at(exp, cells.ravel(), (x[:, 0] * view)[cells.ravel()])
        assert np.abs(bd.data_ro - exp).max() < 1e-12 * max(1.0, np.abs(exp).max()), it


@pytest.mark.gpu
@pytest.mark.parametrize("bcs", [False, True])
def test_c1_residual_and_jacobian_through_func_only(bcs):
    """BASELINE.json configs[0] (Poisson CG1 on UnitSquareMesh(64, 64)) assembled through ``func(start, end, *arglist)``
    ALONE -- device pointers in the reference's positional order, a DeviceMat handle in the Mat slot, BC lgmaps swapped
    in like pyop2/parloop.py:279-314 -- on the staged and owner-computes-rows wrappers, against the oracle."""
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import forms, mesh as fmesh
    from firedrake_amd.device import DeviceBuffer
    msh = fmesh.UnitSquareMesh(64, 64, perturb=0.1)
    V = msh.space(1)
    cells = np.ascontiguousarray(V.cell_node_map.values_with_halo)
    nn, ne = V.node_set.total_size, msh.cell_set.size
    x = np.array(msh.coordinates.data_ro)
    rng = np.random.default_rng(1)
    u, f = rng.standard_normal(nn), rng.standard_normal(nn)
    kr, kj = forms.poisson_residual_kernel(2, 1), forms.poisson_jacobian_kernel(2, 1)
    m = MapKernelArg(arity=3)
    gres = GlobalKernel(local_kernel=CStringLocalKernel(code=kr.code, name=kr.name, accesses=(4, 1, 1, 1), dtypes=(np.float64,) * 4),
                        arguments=[DatKernelArg(dim=(1,), map_=m), DatKernelArg(dim=(2,), map_=m), DatKernelArg(dim=(1,), map_=m),
                                   DatKernelArg(dim=(1,), map_=m)])
    gjac = GlobalKernel(local_kernel=CStringLocalKernel(code=kj.code, name=kj.name, accesses=(4, 1), dtypes=(np.float64,) * 2),
                        arguments=[MatKernelArg(dims=((1, 1),), maps=(m, m)), DatKernelArg(dim=(2,), map_=m)])
    fres, fjac = bridge.compile_global_kernel_hip(gres), bridge.compile_global_kernel_hip(gjac)
    assert (fres.mode, fjac.mode) == ("staged", "ocr")
    # what the patched carriers would hold: device mirrors of the Map, the Dats and the matrix
    m_d, x_d, u_d, f_d, r_d = (DeviceBuffer.from_numpy(a) for a in (cells, x, u, f, np.zeros(nn)))
    bridge.register_map(m_d.ptr, ne, 3, toset_sizes=(nn, nn, nn))
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(V.cell_node_map, V.cell_node_map, None)])
    sp._build()
    vals = DeviceBuffer(sp.nz * 8)
    vals.upload(np.full(sp.nz, 7.0))                                   # stale values: Mat.zero() must clear them
    dm = bridge.DeviceMat(sp._rowptr.ptr, sp._colidx.ptr, vals.ptr, nn, sp.nz, rowptr_bytes=_lib.NNZ_BYTES)
    bc = V.boundary_nodes
    lg = np.arange(nn, dtype=np.int32)
    if bcs:
        lg[bc] = -1
        lg_d = DeviceBuffer.from_numpy(lg)
        dm.set_lgmaps(lg_d.ptr, lg_d.ptr)
    # residual: core part then (empty) owned part, exactly the reference's two calls (parloop.py:250-253)
    fres(0, ne, r_d.ptr, x_d.ptr, u_d.ptr, f_d.ptr, m_d.ptr)
    fres(ne, ne, r_d.ptr, x_d.ptr, u_d.ptr, f_d.ptr, m_d.ptr)
    dm.zero()
    fjac(0, ne, dm.handle, x_d.ptr, m_d.ptr)
    fjac(ne, ne, dm.handle, x_d.ptr, m_d.ptr)
    ro = np.zeros(nn)
    oracle.par_loop(kr.code, kr.name, 0, ne, [ODat(ro, INC, cells), ODat(x, READ, cells), ODat(u, READ, cells), ODat(f, READ, cells)])
    csr = oracle.build_sparsity(nn, nn, [(cells, cells)])
    olg = lg if bcs else None
    oracle.par_loop(kj.code, kj.name, 0, ne, [OMat(csr, INC, cells, cells, row_lgmap=olg, col_lgmap=olg), ODat(x, READ, cells)])
    r = r_d.download(np.float64, (nn,))
    v = vals.download(np.float64, (sp.nz,))
    assert np.array_equal(sp.rowptr, csr.rowptr) and np.array_equal(sp.colidx, csr.colidx)
    assert np.abs(r - ro).max() <= 1e-12 * max(1.0, np.abs(ro).max())
    assert np.abs(v - csr.values).max() <= 1e-12 * np.abs(csr.values).max()
    # a second assembly WITHOUT Mat.zero() accumulates (MatSetValuesLocal ADD_VALUES)
    fjac(0, ne, dm.handle, x_d.ptr, m_d.ptr)
    v2 = vals.download(np.float64, (sp.nz,))
    assert np.abs(v2 - 2.0 * csr.values).max() <= 1e-12 * np.abs(csr.values).max()
    assert len(fjac.loops) == 1 and len(fres.loops) == 1                # plans resolved once, cached on the argument list
    dm.free()


@pytest.mark.gpu
def test_p2_jacobian_through_func_takes_the_sliced_wrapper():
    """The same seam with a 10x10 element matrix: ``func`` resolves to the row-sliced owner-computes-rows wrapper, whose
    kernel has no lgmap parameters -- the pair installed with DeviceMat.set_lgmaps selects the per-instance tables, and
    swapping the pair between calls (pyop2/parloop.py:279-314) switches them."""
    import oracle
    from oracle import ODat, OMat, READ, INC
    from firedrake_amd import forms, mesh as fmesh
    from firedrake_amd.device import DeviceBuffer
    msh = fmesh.UnitCubeMesh(5, degrees=(2,), tile=(4, 4, 2), perturb=0.1)
    V, X = msh.space(2), msh.coord_space
    cells, xcells = np.ascontiguousarray(V.cell_node_map.values_with_halo), np.ascontiguousarray(X.cell_node_map.values_with_halo)
    nn, nx, ne = V.node_set.total_size, X.node_set.total_size, msh.cell_set.size
    x = np.array(msh.coordinates.data_ro)
    kj = forms.poisson_jacobian_kernel(3, 2)
    m, xm = MapKernelArg(arity=10), MapKernelArg(arity=4)
    gjac = GlobalKernel(local_kernel=CStringLocalKernel(code=kj.code, name=kj.name, accesses=(4, 1), dtypes=(np.float64,) * 2),
                        arguments=[MatKernelArg(dims=((1, 1),), maps=(m, m)), DatKernelArg(dim=(3,), map_=xm)])
    fjac = bridge.compile_global_kernel_hip(gjac)
    assert fjac.mode == "ocrs"
    m_d, xm_d, x_d = (DeviceBuffer.from_numpy(a) for a in (cells, xcells, x))
    bridge.register_map(m_d.ptr, ne, 10, toset_sizes=(nn, nn, nn))
    bridge.register_map(xm_d.ptr, ne, 4, toset_sizes=(nx, nx, nx))
    sp = op2.Sparsity((V.node_set ** 1, V.node_set ** 1), [(V.cell_node_map, V.cell_node_map, None)])
    sp._build()
    vals = DeviceBuffer(sp.nz * 8)
    dm = bridge.DeviceMat(sp._rowptr.ptr, sp._colidx.ptr, vals.ptr, nn, sp.nz, rowptr_bytes=_lib.NNZ_BYTES)
    csr = oracle.build_sparsity(nn, nn, [(cells, cells)])
    rng = np.random.default_rng(2)
    pairs = []
    for frac in (5, 3):
        lg = np.arange(nn, dtype=np.int32)
        lg[rng.choice(nn, nn // frac, replace=False)] = -1
        pairs.append((lg, DeviceBuffer.from_numpy(lg)))
    for lg, lg_d in pairs + pairs[:1] + pairs[:1]:
        dm.set_lgmaps(lg_d.ptr, lg_d.ptr)                 # a fresh pair object per call, as the reference does per assemble
        dm.zero()
        fjac(0, ne, dm.handle, x_d.ptr, m_d.ptr, xm_d.ptr)
        csr.values[:] = 0.0
        oracle.par_loop(kj.code, kj.name, 0, ne, [OMat(csr, INC, cells, cells, row_lgmap=lg, col_lgmap=lg), ODat(x, READ, xcells)])
        v = vals.download(np.float64, (sp.nz,))
        assert np.abs(v - csr.values).max() <= 1e-12 * np.abs(csr.values).max()
    (loop,) = fjac.loops.values() if isinstance(fjac.loops, dict) else fjac.loops
    assert len(loop._ocr_geometry(0, ne)["ocr"]._tables) == 2          # one table set per distinct pair, reused by pointer
    dm.free()


# ---- round 3: composed maps, subsets, extruded sets, tensor-product forms and vector blocks through the seam ---------------
ComposedMapKernelArg = _cls("ComposedMapKernelArg")


def test_composed_map_translates_to_one_table_fed_by_the_constituents_pointers():
    """global_kernel.py:73-90 / types/map.py:206-270: the wrapper sees ONE map of the first constituent's arity; the
    reference's arglist carries one pointer per distinct leaf, in first-use order (parloop.py:203-212)."""
    cell_node, facet_cell = MapKernelArg(arity=3), MapKernelArg(arity=1)
    comp = ComposedMapKernelArg(base_maps=(cell_node, facet_cell))
    lk = CStringLocalKernel(code="static void k(double *b, const double *f) { for (int i = 0; i < 3; ++i) b[i] += f[0]; }", name="k",
                            accesses=(4, 1), dtypes=(np.float64,) * 2)
    gk = GlobalKernel(local_kernel=lk, arguments=[DatKernelArg(dim=(1,), map_=comp), DatKernelArg(dim=(1,), map_=facet_cell)])
    fd = bridge.as_fd_global_kernel(gk)
    assert fd.arguments[0].map_.arity == 3 and fd.arguments[1].map_.arity == 1
    assert fd._seam_maps["leaves"] == [cell_node, facet_cell]
    assert list(fd._seam_maps["composed"].values()) == [(cell_node, facet_cell)]
    with pytest.raises(ValueError):
        bridge.as_fd_global_kernel(GlobalKernel(local_kernel=lk, arguments=[
            DatKernelArg(dim=(1,), map_=ComposedMapKernelArg(base_maps=(facet_cell, cell_node))), DatKernelArg(dim=(1,), map_=facet_cell)]))


def test_tensor_form_descriptor_selects_the_matrix_core_wrapper():
    """the Firedrake-side patch attaches ``fdhip_tensor`` (form data it holds in tsfc_interface.py) to the
    local kernel; the translated kernel is a TensorProductLocalKernel and the loop takes the tp_matrix wrapper."""
    from firedrake_amd import forms
    from firedrake_amd.codegen import select_mode
    info = bridge.tensor_form_info("hexahedron", "Q", 4, 8, {"stiffness": 1.0, "mass": 1.0}, "matrix")
    assert info == {"kind": "matrix", "degree": 4, "nq": 5, "alpha": 1.0, "beta": 1.0}
    assert bridge.tensor_form_info("hexahedron", "Q", 3, 6, {"stiffness": 1.0}, "matrix") == \
        {"kind": "matrix", "degree": 3, "nq": 4, "alpha": 1.0, "beta": 0.0}                                    # Q1..Q5: whole panels
    assert bridge.tensor_form_info("hexahedron", "Q", 6, 12, {"stiffness": 1.0}, "matrix") == \
        {"kind": "matrix", "degree": 6, "nq": 7, "alpha": 1.0, "beta": 0.0}                                    # ... and Q6..Q8 since round 6
    assert bridge.tensor_form_info("hexahedron", "Q", 9, 18, {"stiffness": 1.0}, "matrix") is None         # Q9: ordinary wrappers
    assert bridge.tensor_form_info("tetrahedron", "CG", 4, 8, {"stiffness": 1.0}, "matrix") is None
    assert bridge.tensor_form_info("hexahedron", "Q", 4, 8, {"curlcurl": 1.0}, "matrix") is None           # not a term of the descriptor
    adv = bridge.tensor_form_info("hexahedron", "Q", 2, 4, {"stiffness": 0.5, "advection": (1.0, 0.0, -1.0)}, "action")
    assert adv == {"kind": "action", "degree": 2, "nq": 3, "alpha": 0.5, "beta": 0.0, "velocity": (1.0, 0.0, -1.0)}
    dense = forms.helmholtz_q4_hex_jacobian_kernel()
    q4, q1 = MapKernelArg(arity=125, offset=(4,) * 125), MapKernelArg(arity=8, offset=(1,) * 8)
    lk = CStringLocalKernel(code=dense.code, name=dense.name, accesses=(4, 1), dtypes=(np.float64,) * 2, requires_zeroed_output_arguments=True)
    plain = GlobalKernel(local_kernel=lk, arguments=[MatKernelArg(dims=((1, 1),), maps=(q4, q4)), DatKernelArg(dim=(3,), map_=q1)],
                         _extruded=True, _constant_layers=True)
    assert select_mode(bridge.as_fd_global_kernel(plain)) == "direct"        # 125 x 125 without the descriptor: private tensor
    lk.fdhip_tensor = info
    fd = bridge.as_fd_global_kernel(plain)
    assert select_mode(fd) == "tp_matrix" and fd.local_kernel.tp["weights_code"] == dense.tp["weights_code"]
    # a Jacobian with coefficient Functions (variable diffusivity w0, linearisation point u0): the descriptor names their number
    # and the two factors as C expressions; the loop -- Mat, coordinates, w_0, w_1 in TSFC's order -- still takes tp_matrix
    cinfo = bridge.tensor_form_info("hexahedron", "Q", 4, 8, {"stiffness": "1.0 + C[0]", "mass": "1.0 + C[1]*C[1]", "coefficients": 2}, "matrix")
    assert cinfo == {"kind": "matrix", "degree": 4, "nq": 5, "ncoef": 2, "kappa": "1.0 + C[0]", "react": "1.0 + C[1]*C[1]"}
    cdense = forms.coefficient_hex_jacobian_kernel(4, 5)
    clk = CStringLocalKernel(code=cdense.code, name=cdense.name, accesses=(4, 1, 1, 1), dtypes=(np.float64,) * 4, requires_zeroed_output_arguments=True)
    cgk = GlobalKernel(local_kernel=clk, arguments=[MatKernelArg(dims=((1, 1),), maps=(q4, q4)), DatKernelArg(dim=(3,), map_=q1),
                                                    DatKernelArg(dim=(1,), map_=q4), DatKernelArg(dim=(1,), map_=q4)],
                       _extruded=True, _constant_layers=True)
    assert select_mode(bridge.as_fd_global_kernel(cgk)) == "direct"
    clk.fdhip_tensor = cinfo
    cfd = bridge.as_fd_global_kernel(cgk)
    assert select_mode(cfd) == "tp_matrix" and cfd.local_kernel.tp["ncoef"] == 2 and cfd.local_kernel.tp["weights_code"] == cdense.tp["weights_code"]


def test_tensor_form_descriptor_for_vector_spaces_and_coefficient_gradients():
    """The wider descriptors through the seam: elasticity on a VectorFunctionSpace (``value_size`` 3: Mat dims (3, 3), blocked element
    tensor), the Newton Jacobian of a form nonlinear in grad(u) (the coefficient's gradient at the points), and the general form -- the
    Firedrake-side patch hands the point-weight callback itself."""
    from firedrake_amd import forms, tensor
    from firedrake_amd.codegen import select_mode
    q2, q1 = MapKernelArg(arity=27, offset=(2,) * 27), MapKernelArg(arity=8, offset=(1,) * 8)
    einfo = bridge.tensor_form_info("hexahedron", "Q", 2, 4, {"elasticity": (1.0, 1.25)}, "matrix", value_size=3)
    assert einfo == {"kind": "matrix", "degree": 2, "nq": 3, "family": "elasticity", "mu": 1.0, "lam": 1.25, "rho": 0.0, "vdim": 3}
    assert bridge.tensor_form_info("hexahedron", "Q", 2, 4, {"elasticity": (1.0, 1.25)}, "matrix") is None           # a scalar space
    assert bridge.tensor_form_info("hexahedron", "Q", 2, 4, {"stiffness": 1.0}, "matrix", value_size=3) is None      # not a vector form
    dense = forms.elasticity_hex_jacobian_kernel(2, 3)
    lk = CStringLocalKernel(code=dense.code, name=dense.name, accesses=(4, 1), dtypes=(np.float64,) * 2, requires_zeroed_output_arguments=True)
    gk = GlobalKernel(local_kernel=lk, arguments=[MatKernelArg(dims=((3,), (3,)), maps=(q2, q2)), DatKernelArg(dim=(3,), map_=q1)],
                      _extruded=True, _constant_layers=True)
    assert select_mode(bridge.as_fd_global_kernel(gk)) != "tp_matrix"
    lk.fdhip_tensor = einfo
    fd = bridge.as_fd_global_kernel(gk)
    assert select_mode(fd) == "tp_matrix" and fd.local_kernel.tp["vdim"] == 3 and fd.local_kernel.tp["weights_code"] == dense.tp["weights_code"]
    ninfo = bridge.tensor_form_info("hexahedron", "Q", 3, 6, {"nonlinear_diffusion": 1}, "action")
    assert ninfo == {"kind": "action", "degree": 3, "nq": 4, "family": "nonlinear_diffusion", "ncoef": 1, "coef_gradients": True}
    q3 = MapKernelArg(arity=64, offset=(3,) * 64)
    adense = forms.nonlinear_diffusion_hex_action_kernel(3, 4)
    alk = CStringLocalKernel(code=adense.code, name=adense.name, accesses=(4, 1, 1, 1), dtypes=(np.float64,) * 4, requires_zeroed_output_arguments=True)
    agk = GlobalKernel(local_kernel=alk, arguments=[DatKernelArg(dim=(1,), map_=q3), DatKernelArg(dim=(3,), map_=q1), DatKernelArg(dim=(1,), map_=q3),
                                                    DatKernelArg(dim=(1,), map_=q3)], _extruded=True, _constant_layers=True)
    alk.fdhip_tensor = ninfo
    afd = bridge.as_fd_global_kernel(agk)
    assert select_mode(afd) == "tp_action" and afd.local_kernel.tp["coef_gradients"] and afd.local_kernel.tp["weights_code"] == adense.tp["weights_code"]
    # the general road: the callback text itself (NAME stands for the kernel's name)
    ginfo = bridge.tensor_form_info("hexahedron", "Q", 3, 6, {"weights_code": tensor.NONLINEAR_DIFFUSION_WEIGHTS, "coefficients": 1,
                                                             "coefficient_gradients": True}, "action")
    alk.fdhip_tensor = ginfo
    gfd = bridge.as_fd_global_kernel(agk)
    assert select_mode(gfd) == "tp_action" and gfd.local_kernel.tp["weights_code"] == adense.tp["weights_code"]


def _dev(*arrays):
    from firedrake_amd.device import DeviceBuffer
    return [DeviceBuffer.from_numpy(np.ascontiguousarray(a)) for a in arrays]


@pytest.mark.gpu
@pytest.mark.parametrize("undefined", [False, True])
def test_composed_map_through_func(undefined):
    """A Dat accessed through ComposedMap(cell_node, facet_cell): ``func`` receives both constituents' pointers and builds
    the composed table on the device.  With undefined (negative) intermediate entries the composed rows are undefined and
    the loop runs over the active entities only -- a Subset, as types/map.py:289-301 prescribes."""
    rng = np.random.default_rng(3)
    nn, nc, nf = 300, 500, 800
    cell_node = rng.integers(0, nn, size=(nc, 3)).astype(np.int32)
    facet_cell = rng.integers(0, nc, size=(nf, 1)).astype(np.int32)
    if undefined:
        facet_cell[::17, 0] = -1
    f = rng.standard_normal(nf)
    a, b = MapKernelArg(arity=3), MapKernelArg(arity=1)
    code = "static void k(double *y, const double *f) { for (int i = 0; i < 3; ++i) y[i] += f[0]; }"
    gk = GlobalKernel(local_kernel=CStringLocalKernel(code=code, name="k", accesses=(4, 1), dtypes=(np.float64,) * 2), _subset=undefined,
                      arguments=[DatKernelArg(dim=(1,), map_=ComposedMapKernelArg(base_maps=(a, b))), DatKernelArg(dim=(1,))])
    func = bridge.compile_global_kernel_hip(gk)
    y_d, f_d, a_d, b_d = _dev(np.zeros(nn), f, cell_node, facet_cell)
    bridge.register_map(a_d.ptr, nc, 3, toset_sizes=(nn,) * 3)
    bridge.register_map(b_d.ptr, nf, 1, toset_sizes=(nc,) * 3)
    active = np.nonzero(facet_cell[:, 0] >= 0)[0].astype(np.int32)
    if undefined:
        (idx_d,) = _dev(active)
        func(0, len(active), idx_d.ptr, y_d.ptr, f_d.ptr, a_d.ptr, b_d.ptr)
    else:
        func(0, nf, y_d.ptr, f_d.ptr, a_d.ptr, b_d.ptr)
    exp = np.zeros(nn)
    for e in active:
        for i in range(3):
            exp[cell_node[facet_cell[e, 0], i]] += f[e]
    got = y_d.download(np.float64, (nn,))
    assert np.abs(got - exp).max() <= 1e-12 * max(1.0, np.abs(exp).max())
    for buf in (a_d, b_d) + ((idx_d,) if undefined else ()):
        bridge.forget(buf.ptr)                 # what the carriers' finalisers do: the addresses are about to be reused
    assert not func.loops


@pytest.mark.gpu
def test_dg_ds_subset_loop_through_func():
    """Config C4's exterior-facet integral over a ``ds(subdomain)`` Subset (a direct uint32 facet-number Dat addressed by
    the base entity, READ Globals) through ``func(start, end, subset, *args, *maps)`` against the oracle."""
    from firedrake_amd import forms, mesh as fmesh
    from helpers import oracle_run
    m = fmesh.make_quad_mesh(24, perturb=0.1)
    prob = forms.DGAdvectionProblem(m)
    _, ke, _ = forms.dg_advection_kernels()
    ec = np.asarray(m.ext_dq.values_with_halo)[:, 0] // 4
    side = np.nonzero(m.dq_points.reshape(-1, 4, 2)[ec][:, :, 0].min(axis=1) < 0.5 / 24)[0].astype(np.int32)    # facets of cells at x ~ 0
    sub = op2.Subset(m.ext_facet_set, side)
    L = op2.Dat(m.dq_set)
    args = (L(op2.INC, m.ext_dq), m.coordinates(op2.READ, m.ext_q1), prob.q(op2.READ, m.ext_dq), prob.u(op2.READ, m.ext_q1),
            prob.dtc(op2.READ), prob.q_in(op2.READ), m.ext_local_facet(op2.READ))
    ref = oracle_run(ke, sub, *args)[0]
    dq, q1 = MapKernelArg(arity=4), MapKernelArg(arity=4)
    lk = CStringLocalKernel(code=ke.code, name=ke.name, accesses=(4, 1, 1, 1, 1, 1, 1), dtypes=(np.float64,) * 6 + (np.uint32,))
    gk = GlobalKernel(local_kernel=lk, _subset=True,
                      arguments=[DatKernelArg(dim=(1,), map_=dq), DatKernelArg(dim=(2,), map_=q1), DatKernelArg(dim=(1,), map_=dq),
                                 DatKernelArg(dim=(2,), map_=q1), GlobalKernelArg(dim=(1,)), GlobalKernelArg(dim=(1,)), DatKernelArg(dim=(1,))])
    func = bridge.compile_global_kernel_hip(gk)
    nd, nv, nf = m.dq_set.total_size, m.q1_set.total_size, m.ext_facet_set.total_size
    L_d, x_d, q_d, u_d, dt_d, qin_d, lf_d, dq_d, q1_d, idx_d = _dev(
        np.zeros(nd), np.array(m.coordinates.data_ro), np.array(prob.q.data_ro), np.array(prob.u.data_ro), np.array(prob.dtc.data_ro),
        np.array(prob.q_in.data_ro), np.array(m.ext_local_facet.data_ro), np.asarray(m.ext_dq.values_with_halo),
        np.asarray(m.ext_q1.values_with_halo), sub.indices.astype(np.int32))
    bridge.register_map(dq_d.ptr, nf, 4, toset_sizes=(nd,) * 3, values=np.asarray(m.ext_dq.values_with_halo))
    bridge.register_map(q1_d.ptr, nf, 4, toset_sizes=(nv,) * 3)                 # values are downloaded once when not given
    func(0, len(sub.indices), idx_d.ptr, L_d.ptr, x_d.ptr, q_d.ptr, u_d.ptr, dt_d.ptr, qin_d.ptr, lf_d.ptr, dq_d.ptr, q1_d.ptr)
    got = L_d.download(np.float64, (nd,))
    assert np.abs(ref).max() > 0 and np.abs(got - ref.reshape(-1)).max() <= 1e-12 * np.abs(ref).max()
    for buf in (dq_d, q1_d, idx_d):
        bridge.forget(buf.ptr)


@pytest.mark.gpu
@pytest.mark.parametrize("bcs", [False, True])
def test_q4_jacobian_and_action_through_func(bcs):
    """Config C3 through the seam alone: ``func(start, end, layers, mat, coords, map_q4, map_q1)`` over the extruded set,
    the local kernel translated from a stand-in carrying the TSFC-shaped C text plus the ``fdhip_tensor`` descriptor -> the
    fp64-MFMA matrix wrapper and the sum-factorised action wrapper, BC lgmaps swapped in, against the oracle."""
    from firedrake_amd import forms, mesh as fmesh
    from helpers import oracle_run
    n = 3
    m = fmesh.make_extruded_hex_mesh(n, n, 4, perturb=0.1)
    cm, xm = m.cell_node_map, m.coord_map
    nn, nx, ncol = m.node_set.total_size, m.coord_node_set.total_size, m.base_set.size
    sp = op2.Sparsity((m.node_set ** 1, m.node_set ** 1), [(cm, cm, None)])
    mat = op2.Mat(sp)
    lg = None
    if bcs:
        pts = m.node_points
        rlg = np.arange(nn, dtype=np.int32)
        rlg[np.nonzero(((pts < 1e-12) | (pts > 1 - 1e-12)).any(axis=1))[0]] = -1
        lg = (rlg, rlg.copy())
    kj, ka = forms.helmholtz_q4_hex_jacobian_kernel(), forms.helmholtz_q4_hex_action_kernel()
    ref = oracle_run(kj, m.cell_set, mat(op2.INC, (cm, cm), lgmaps=lg), m.coordinates(op2.READ, xm))[0]
    u = np.random.default_rng(4).standard_normal(nn)
    refy = oracle_run(ka, m.cell_set, op2.Dat(m.node_set)(op2.INC, cm), m.coordinates(op2.READ, xm), op2.Dat(m.node_set, u)(op2.READ, cm))[0]
    q4, q1 = MapKernelArg(arity=125, offset=(4,) * 125), MapKernelArg(arity=8, offset=(1,) * 8)
    lkj = CStringLocalKernel(code=kj.code, name=kj.name, accesses=(4, 1), dtypes=(np.float64,) * 2, requires_zeroed_output_arguments=True)
    lkj.fdhip_tensor = bridge.tensor_form_info("hexahedron", "Q", 4, 8, {"stiffness": 1.0, "mass": 1.0}, "matrix")
    lka = CStringLocalKernel(code=ka.code, name=ka.name, accesses=(4, 1, 1), dtypes=(np.float64,) * 3, requires_zeroed_output_arguments=True)
    lka.fdhip_tensor = bridge.tensor_form_info("hexahedron", "Q", 4, 8, {"stiffness": 1.0, "mass": 1.0}, "action")
    flags = dict(_extruded=True, _constant_layers=True)
    fj = bridge.compile_global_kernel_hip(GlobalKernel(local_kernel=lkj, arguments=[MatKernelArg(dims=((1, 1),), maps=(q4, q4)),
                                                                                 DatKernelArg(dim=(3,), map_=q1)], **flags))
    fa = bridge.compile_global_kernel_hip(GlobalKernel(local_kernel=lka, arguments=[DatKernelArg(dim=(1,), map_=q4), DatKernelArg(dim=(3,), map_=q1),
                                                                                 DatKernelArg(dim=(1,), map_=q4)], **flags))
    assert (fj.mode, fa.mode) == ("tp_matrix", "tp_action")
    sp._build()
    from firedrake_amd.device import DeviceBuffer
    vals = DeviceBuffer(sp.nz * 8)
    vals.upload(np.zeros(sp.nz))
    dm = bridge.DeviceMat(sp._rowptr.ptr, sp._colidx.ptr, vals.ptr, nn, sp.nz, rowptr_bytes=_lib.NNZ_BYTES)
    lay_d, x_d, q4_d, q1_d, y_d, u_d = _dev(np.asarray(m.cell_set.layers_array, dtype=np.int32), np.array(m.coordinates.data_ro),
                                           np.asarray(cm.values_with_halo), np.asarray(xm.values_with_halo), np.zeros(nn), u)
    bridge.register_map(q4_d.ptr, ncol, 125, toset_sizes=(nn,) * 3, values=np.asarray(cm.values_with_halo))
    bridge.register_map(q1_d.ptr, ncol, 8, toset_sizes=(nx,) * 3, values=np.asarray(xm.values_with_halo))
    if bcs:
        (lg_d,) = _dev(lg[0])
        dm.set_lgmaps(lg_d.ptr, lg_d.ptr)
    fj(0, ncol, lay_d.ptr, dm.handle, x_d.ptr, q4_d.ptr, q1_d.ptr)
    v = vals.download(np.float64, (sp.nz,))
    assert np.abs(v - ref.values).max() <= 1e-11 * np.abs(ref.values).max()
    fa(0, ncol, lay_d.ptr, y_d.ptr, x_d.ptr, u_d.ptr, q4_d.ptr, q1_d.ptr)
    y = y_d.download(np.float64, (nn,))
    assert np.abs(y - refy).max() <= 1e-11 * np.abs(refy).max()
    dm.free()
    for buf in (lay_d, q4_d, q1_d):
        bridge.forget(buf.ptr)


@pytest.mark.gpu
def test_vector_valued_blocks_through_func():
    """MatSetValuesBlockedLocal (builder.py:573-625) through the seam: vector P1 on tetrahedra, 3 x 3 blocks; the DeviceMat is
    described by its NODE pattern and holds the block-expanded scalar CSR values."""
    from firedrake_amd import mesh as fmesh
    from firedrake_amd.device import DeviceBuffer
    from helpers import oracle_run
    from mixed_cases import vector_p1_elasticity_kernel
    msh = fmesh.UnitCubeMesh(4, perturb=0.1)
    V = msh.space(1)
    cm = V.cell_node_map
    sp = op2.Sparsity((V.node_set ** 3, V.node_set ** 3), [(cm, cm, None)])
    mat = op2.Mat(sp)
    k = vector_p1_elasticity_kernel(3)
    ref = oracle_run(k, msh.cell_set, mat(op2.INC, (cm, cm)), msh.coordinates(op2.READ, cm))[0]
    sp._build()
    nn, ne = V.node_set.total_size, msh.cell_set.size
    vals = DeviceBuffer(sp.nz * 8)
    # (a PETSc build with 32-bit PetscInt hands over int32 row starts: DeviceMat widens them once)
    narrow = DeviceBuffer.from_numpy(np.ascontiguousarray(sp._node_rowptr_host(), dtype=np.int32))
    dm = bridge.DeviceMat(narrow.ptr, sp._node_colidx.ptr, vals.ptr, nn, sp._node_nnz, rbs=3, cbs=3, rowptr_bytes=4)
    m = MapKernelArg(arity=4)
    gk = GlobalKernel(local_kernel=CStringLocalKernel(code=k.code, name=k.name, accesses=(4, 1), dtypes=(np.float64,) * 2),
                      arguments=[MatKernelArg(dims=((3, 3),), maps=(m, m)), DatKernelArg(dim=(3,), map_=m)])
    func = bridge.compile_global_kernel_hip(gk)
    assert func.mode == "ocrs"
    m_d, x_d = _dev(np.asarray(cm.values_with_halo), np.array(msh.coordinates.data_ro))
    bridge.register_map(m_d.ptr, ne, 4, toset_sizes=(nn,) * 3)
    dm.zero()
    func(0, ne, dm.handle, x_d.ptr, m_d.ptr)
    v = vals.download(np.float64, (sp.nz,))
    assert np.abs(v - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
    dm.free()
