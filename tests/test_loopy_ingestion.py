"""The loopy -> HIP local-kernel route (pyop2/local_kernel.py:210-227): C text in loopy's
dialect goes through the wrapper generator unchanged, on the CPU (host-sim vs oracle vs the reference's golden
vectors) and on the GPU; LoopyLocalKernel lowers a loopy kernel with lp.generate_code_v2 (exercised with a stand-in
``loopy`` module, since the package is not installed here)."""
import sys
import types

import numpy as np
import pytest
from numpy.testing import assert_allclose

from firedrake_amd import op2
import golden_kernels as gk
import loopy_dialect as ld
from helpers import oracle_run

G = gk.GOLD


def _golden_mesh():
    nodes, ele = op2.Set(4, "nodes"), op2.Set(2, "elements")
    m = op2.Map(ele, nodes, 3, gk.ELEM_NODE, "elem_node")
    return nodes, ele, m, op2.Dat(nodes ** 2, gk.COORDS), op2.Dat(nodes, gk.F)


def test_host_only_includes_are_dropped():
    from firedrake_amd.codegen import generate_wrapper
    nodes, ele, m, x, f = _golden_mesh()
    k = op2.loopy_c_kernel(ld.RHS_P1, "form1_cell_integral")
    assert k.requires_zeroed_output_arguments
    pl = op2.LegacyParloop(k, ele, op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    src = generate_wrapper(pl.global_kernel, "direct").source
    assert "petsc.h" not in src and "complex.h" not in src and "#include <stdint.h>" in src


def test_loopy_dialect_kernels_reproduce_reference_goldens_on_host():
    from hostsim import run_direct
    nodes, ele, m, x, f = _golden_mesh()
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    k = op2.loopy_c_kernel(ld.MASS_P1, "form0_cell_integral")
    args = (mat(op2.INC, (m, m)), x(op2.READ, m))
    got = run_direct(op2.LegacyParloop(k, ele, *args))[0]
    assert_allclose(got.todense(), np.array(G["expected_matrix"]), rtol=G["expected_matrix_rtol"], atol=1e-7)
    assert_allclose(got.values, oracle_run(k, ele, *args)[0].values, rtol=1e-14)
    k = op2.loopy_c_kernel(ld.RHS_P1, "form1_cell_integral")
    args = (op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    got = run_direct(op2.LegacyParloop(k, ele, *args))[0]
    assert_allclose(got, G["expected_rhs"], rtol=G["expected_rhs_rtol_quad3"])
    assert_allclose(got, oracle_run(k, ele, *args)[0], rtol=1e-14)


def test_loopy_local_kernel_lowers_with_generate_code_v2(monkeypatch):
    """LoopyLocalKernel asks loopy for the C text and the argument dtypes (local_kernel.py:210-227)."""
    class ArrayArg:
        def __init__(self, dtype):
            self.dtype = np.dtype(dtype)

    class TranslationUnit:                       # duck-typed like loopy's
        def __init__(self, name, args):
            self.callables_table = {name: types.SimpleNamespace(subkernel=types.SimpleNamespace(args=args))}

    fake = types.ModuleType("loopy")
    fake.ArrayArg = ArrayArg
    fake.generate_code_v2 = lambda tu: types.SimpleNamespace(device_code=lambda: ld.MASS_P1)
    monkeypatch.setitem(sys.modules, "loopy", fake)
    tu = TranslationUnit("form0_cell_integral", [ArrayArg("float64"), ArrayArg("float64")])
    k = op2.Kernel(tu, "form0_cell_integral", accesses=[op2.INC, op2.READ])
    assert isinstance(k, op2.LoopyLocalKernel) and k.code == ld.MASS_P1
    assert k.dtypes == (np.dtype("float64"), np.dtype("float64"))
    with pytest.raises(TypeError):
        op2.Kernel(42, "nope")


def test_loopy_local_kernel_without_loopy_fails_loudly(monkeypatch):
    monkeypatch.setitem(sys.modules, "loopy", None)
    with pytest.raises(ImportError, match="loopy"):
        op2.LoopyLocalKernel(object(), "k")


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["auto", "direct"])
def test_loopy_dialect_kernels_on_gpu(mode, monkeypatch):
    from firedrake_amd.configuration import configuration
    from helpers import structured_tri_mesh
    monkeypatch.setitem(configuration, "mode", mode)
    nodes, ele, m, x, f = _golden_mesh()
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    km = op2.loopy_c_kernel(ld.MASS_P1, "form0_cell_integral")
    op2.par_loop(km, ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    assert_allclose(mat.values, np.array(G["expected_matrix"]), rtol=G["expected_matrix_rtol"], atol=1e-7)
    kr = op2.loopy_c_kernel(ld.RHS_P1, "form1_cell_integral")
    b = op2.Dat(nodes)
    op2.par_loop(kr, ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    assert_allclose(b.data_ro, G["expected_rhs"], rtol=G["expected_rhs_rtol_quad3"])
    # a mesh large enough for several plan blocks, against the oracle
    coords, cells = structured_tri_mesh(70, 60, perturb=0.2)
    nodes, ele = op2.Set(len(coords)), op2.Set(len(cells))
    m = op2.Map(ele, nodes, 3, cells)
    x = op2.Dat(nodes ** 2, coords)
    f = op2.Dat(nodes, np.random.default_rng(0).standard_normal(len(coords)))
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(m, m, None)]))
    op2.par_loop(km, ele, mat(op2.INC, (m, m)), x(op2.READ, m))
    ref = oracle_run(km, ele, mat(op2.INC, (m, m)), x(op2.READ, m))[0]
    assert np.abs(mat.csr()[2] - ref.values).max() <= 1e-12 * np.abs(ref.values).max()
    assert abs(mat.csr()[2].sum() - 1.0) < 1e-12          # sum of the mass matrix = |Omega|
    b = op2.Dat(nodes)
    op2.par_loop(kr, ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    ref = oracle_run(kr, ele, op2.Dat(nodes)(op2.INC, m), x(op2.READ, m), f(op2.READ, m))[0]
    assert np.abs(b.data_ro - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
