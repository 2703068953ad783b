"""CPU: the PyOP2-side seam (firedrake_amd/bridge.py).  PyOP2 cannot be imported here, so its descriptors are played
by stand-ins with the reference's class names and fields (pyop2/global_kernel.py:27-325, local_kernel.py:86-227):
the translated kernel must generate exactly the wrapper the natively built kernel generates."""
import types

import numpy as np
import pytest

from firedrake_amd import bridge, op2
from firedrake_amd.codegen import generate_wrapper


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


def test_function_level_seam_rejects_mat_arguments():
    m = MapKernelArg(arity=3)
    lk = CStringLocalKernel(code="static void k(double *A) {}", name="k", accesses=(4,), dtypes=(np.float64,))
    gk = GlobalKernel(local_kernel=lk, arguments=[MatKernelArg(dims=((1, 1),), maps=(m, m))])
    assert bridge.as_fd_global_kernel(gk).arguments[0].dims == (1, 1)
    with pytest.raises(NotImplementedError):
        bridge.compile_global_kernel_hip(gk)


def test_function_level_seam_builds_the_direct_wrapper():
    from firedrake_amd import _lib
    m = MapKernelArg(arity=3)
    lk = CStringLocalKernel(code=RHS, name="rhs", accesses=(4, 1, 1), dtypes=(np.float64,) * 3)
    gk = GlobalKernel(local_kernel=lk, arguments=[DatKernelArg(dim=(), map_=m), DatKernelArg(dim=(2,), map_=m),
                                                  DatKernelArg(dim=(1,), map_=m)])
    func = bridge.compile_global_kernel_hip(gk)
    assert func.wrapper.src.mode == "direct" and func.wrapper.path.endswith(".hsaco")
    with pytest.raises(ValueError):
        func(0, 2, 1, 2)                                     # three Dat pointers + one map pointer are expected
    if not _lib.gpu_available():
        with pytest.raises(_lib.FDHipError):                 # no device, no fallback
            func(0, 2, 0, 0, 0, 0)


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
