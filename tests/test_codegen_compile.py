"""CPU: every wrapper shape the code generator can emit cross-compiles for gfx950 with hipcc
(no GPU needed) -- the analogue of the reference's "does the generated C compile" guarantee
(pyop2/compilation.py:527-611)."""
import numpy as np
import pytest

from firedrake_amd import op2
from firedrake_amd.codegen import generate_wrapper, staged_eligible
from firedrake_amd.compilation import compile_hip
import golden_kernels as gk


def _mesh():
    nodes, ele = op2.Set(4), op2.Set(2)
    m = op2.Map(ele, nodes, 3, gk.ELEM_NODE)
    return nodes, ele, m


def _compile(pl, modes=("staged", "direct")):
    out = []
    for mode in modes:
        if mode == "staged" and not staged_eligible(pl.global_kernel):
            continue
        src = generate_wrapper(pl.global_kernel, mode)
        path = compile_hip(src.source, src.symbol + "_" + mode)
        assert path.endswith(".hsaco")
        out.append(mode)
    return out


def test_rhs_staged_and_direct():
    nodes, ele, m = _mesh()
    b, x, f = op2.Dat(nodes), op2.Dat(nodes ** 2, gk.COORDS), op2.Dat(nodes, gk.F)
    pl = op2.LegacyParloop(op2.Kernel(gk.RHS_Q6, "rhs_q6"), ele, b(op2.INC, m), x(op2.READ, m), f(op2.READ, m))
    assert _compile(pl) == ["staged", "direct"]


@pytest.mark.parametrize("scatter", ["table", "search"])
@pytest.mark.parametrize("cdim", [1, 2])
def test_mat_wrappers(scatter, cdim, monkeypatch):
    from firedrake_amd.configuration import configuration
    monkeypatch.setitem(configuration, "mat_scatter", scatter)
    nodes, ele, m = _mesh()
    mat = op2.Mat(op2.Sparsity((nodes ** cdim, nodes ** cdim), [(m, m, None)]))
    x = op2.Dat(nodes ** 2, gk.COORDS)
    k = op2.Kernel(gk.MASS_AFFINE if cdim == 1 else gk.MASS_VEC_AFFINE, "mass_affine" if cdim == 1 else "mass_vec_affine")
    lg = np.array([-1, 1, 2, 3], dtype=np.int32)
    for lgmaps in (None, (lg, lg)):
        pl = op2.LegacyParloop(k, ele, mat(op2.INC, (m, m), lgmaps=lgmaps), x(op2.READ, m))
        assert _compile(pl) == ["staged", "direct"]


def test_integer_minmax_global_subset_permuted():
    it, ind = op2.Set(16), op2.Set(16)
    idm = op2.Map(it, ind, 1, np.arange(16))
    a, b = op2.Dat(ind, dtype=np.int32), op2.Dat(ind, dtype=np.int32)
    _compile(op2.LegacyParloop(op2.Kernel("static void mx(int *a, int *b) { *a = *a < *b ? *b : *a; }", "mx"),
                               it, a(op2.MAX, idm), b(op2.READ, idm)))
    x = op2.Dat(ind, dtype=np.uint32)
    g = op2.Global(1, 0, np.uint32)
    _compile(op2.LegacyParloop(op2.Kernel("static void gi(unsigned int *x, unsigned int *inc) { (*x) = (*x) + 1; (*inc) += (*x); }", "gi"),
                               it, x(op2.RW, idm), g(op2.INC)))
    ss = op2.Subset(it, [1, 3, 5])
    d = op2.Dat(ind)
    _compile(op2.LegacyParloop(op2.Kernel("static void one(double *x) { *x += 1.0; }", "one"), ss, d(op2.INC, idm)))
    m1 = op2.Map(op2.Set(1), op2.Set(4), 4, [1, 2, 3, 0])
    m2 = op2.PermutedMap(m1, [3, 2, 0, 1])
    d1, d2 = op2.Dat(m1.toset, dtype=np.int32), op2.Dat(m1.toset, dtype=np.int32)
    _compile(op2.LegacyParloop(op2.Kernel("void cp(int *to, const int *from) { for (int i = 0; i < 4; i++) to[i] = from[i]; }", "cp"),
                               m1.iterset, d2(op2.WRITE, m2), d1(op2.READ, m1)))


def test_extruded_wrappers():
    base = op2.Set(4)
    ext = op2.ExtrudedSet(base, layers=5)
    nodes = op2.Set(6 * 5)
    cm = op2.Map(ext, nodes, 6, np.arange(24) % 25, offset=[1] * 6)
    x = op2.Dat(nodes ** 2)
    g = op2.Global(1, 0.0)
    k = op2.Kernel("static void vol(double A[1], const double x[12]) { A[0] += x[0]; }", "vol")
    for region in (None, op2.ON_BOTTOM, op2.ON_TOP):
        pl = op2.LegacyParloop(k, ext, g(op2.INC), x(op2.READ, cm), iteration_region=region)
        assert _compile(pl) == ["staged", "direct"]        # staged over the (column, layer) cells through a derived map
    k2 = op2.Kernel("static void volf(double A[1], const double x[24]) { A[0] += x[12]; }", "volf")
    pl = op2.LegacyParloop(k2, ext, g(op2.INC), x(op2.READ, cm), iteration_region=op2.ON_INTERIOR_FACETS)
    assert _compile(pl) == ["staged", "direct"]            # (a derived row holds the nodes of both stacked cells)
    d = op2.Dat(base)
    pl = op2.LegacyParloop(op2.Kernel("static void k1(double *x) { *x += 1.0; }", "k1"), ext, d(op2.INC))
    assert _compile(pl) == ["direct"]
    f = op2.Dat(op2.Set(4 * 4))
    fm = op2.Map(ext, f.dataset.set, 1, np.arange(4) * 4, offset=[1])
    pl = op2.LegacyParloop(op2.Kernel("static void blah(double* x, int layer_arg){ x[0] = layer_arg; }", "blah"), ext,
                           f(op2.WRITE, fm), pass_layer_arg=True)
    assert _compile(pl) == ["direct"]


def test_mixed_wrappers_compile():
    """Mixed arguments are flattened to per-part arguments + an adaptor local kernel (GlobalKernel.flattened)."""
    from mixed_cases import ADDONE_MAT, ADDONE_RHS_VEC, mixed_kernels, reference_mixed_fixture, velocity_pressure_space
    mset, mdat, mvdat, mmap, msparsity = reference_mixed_fixture()
    mat = op2.Mat(msparsity)
    pl = op2.LegacyParloop(op2.Kernel(ADDONE_MAT, "addone_mat"), mmap.iterset, mat(op2.INC, (mmap, mmap)), mdat(op2.READ, mmap))
    assert len(pl.arguments) == 6 and not pl.global_kernel.is_mixed
    assert _compile(pl) == ["staged", "direct"]
    dat = op2.MixedDat(mset ** 2)
    pl = op2.LegacyParloop(op2.Kernel(ADDONE_RHS_VEC, "addone_rhs_vec"), mmap.iterset, dat(op2.INC, mmap), mvdat(op2.READ, mmap))
    assert _compile(pl) == ["staged", "direct"]
    ele, ms, mm, x, vmap = velocity_pressure_space(3, 3, 2)
    mds = op2.MixedDataSet(ms, (2, 1))
    sp = op2.Sparsity((mds, mds), {(i, j): [(rm, cm, None)] for i, rm in enumerate(mm) for j, cm in enumerate(mm)})
    jac, res = mixed_kernels(2)
    lgv, lgp = np.arange(ms[0].total_size, dtype=np.int32), np.arange(ms[1].total_size, dtype=np.int32)
    for lgmaps in (None, [(lgv, lgv), (lgv, lgp), (lgp, lgv), (lgp, lgp)]):
        pl = op2.LegacyParloop(jac, ele, op2.Mat(sp)(op2.INC, (mm, mm), lgmaps=lgmaps), x(op2.READ, vmap))
        assert _compile(pl) == ["staged", "direct"]


def test_periodic_extruded_wrappers_compile():
    from mixed_cases import periodic_column_mesh
    base, ext, nodes, cm = periodic_column_mesh(np.random.default_rng(0))
    x, out = op2.Dat(nodes ** 2), op2.Dat(nodes)
    for region, nf in ((None, 1), (op2.ON_INTERIOR_FACETS, 2)):
        k = op2.Kernel("static void kp(double *o, const double *x) { for (int i = 0; i < %d; ++i) o[i] += x[2*i]; }" % (6 * nf), "kp")
        pl = op2.LegacyParloop(k, ext, out(op2.INC, cm), x(op2.READ, cm), iteration_region=region)
        assert pl.global_kernel._extruded_periodic
        # (the wrap is folded into the derived map: the staged wrapper runs cell regions and interior facets alike)
        assert _compile(pl) == ["staged", "direct"]


def test_row_sliced_wrappers():
    """The row-sliced owner-computes-rows wrappers ("ocrs": contiguous flush, "ocrsp": row-by-row flush under a row order,
    8- and 16-bit column positions) cross-compile, each instantiation of the local kernel without scratch."""
    from firedrake_amd import forms
    from firedrake_amd.codegen import select_mode
    from firedrake_amd.compilation import kernel_resources
    nodes, ele, vs = op2.Set(40), op2.Set(2), op2.Set(8)
    cm, xm = op2.Map(ele, nodes, 10, np.arange(20)), op2.Map(ele, vs, 4, np.arange(8))
    mat = op2.Mat(op2.Sparsity((nodes ** 1, nodes ** 1), [(cm, cm, None)]))
    x = op2.Dat(vs ** 3)
    lg = np.arange(40, dtype=np.int32)
    pl = op2.LegacyParloop(forms.poisson_jacobian_kernel(3, 2), ele, mat(op2.INC, (cm, cm), lgmaps=(lg, lg)), x(op2.READ, xm))
    assert select_mode(pl.global_kernel) == "ocrs"
    for mode in ("ocrs", "ocrsp", "ocrs_k16", "ocrsp_s448"):
        src = generate_wrapper(pl.global_kernel, mode)
        assert "rlg" not in src.source and src.source.count("fdk::poisson_p2_tet_jacobian(") == 10
        path = compile_hip(src.source, src.symbol + "_" + mode)
        res = kernel_resources(path, src.symbol)
        assert res and res.get("scratch", 0) == 0 and res.get("vgprs", 999) <= 128


def test_scratch_directed_unrolling(monkeypatch):
    """A 12x12 element matrix written as four nested loops is not fully unrolled at LLVM's default threshold: the element
    tensor stays in scratch memory.  GlobalKernel.compile notices hipcc's scratch report and rebuilds the wrapper with a
    high -unroll-threshold; the flag sticks to the wrapper source."""
    from firedrake_amd.compilation import kernel_resources
    from firedrake_amd.configuration import configuration
    from mixed_cases import vector_p1_elasticity_kernel
    nodes, ele = op2.Set(40), op2.Set(2)
    cm = op2.Map(ele, nodes, 4, np.arange(8))
    x = op2.Dat(nodes ** 3)

    def build():
        mat = op2.Mat(op2.Sparsity((nodes ** 3, nodes ** 3), [(cm, cm, None)]))
        k = vector_p1_elasticity_kernel(3)
        gk_ = op2.LegacyParloop(k, ele, mat(op2.INC, (cm, cm)), x(op2.READ, cm)).global_kernel
        gk_._compiled.clear()                 # (GlobalKernels are cached per local kernel + argument shapes)
        return gk_
    monkeypatch.setitem(configuration, "unroll_retry_threshold", 0)
    from firedrake_amd.kernel import GlobalKernel
    monkeypatch.setattr(GlobalKernel, "_cache", {})
    cw0 = build().compile("ocrs")
    assert kernel_resources(cw0.path, cw0.src.symbol)["scratch"] > 0 and cw0.src.extra_flags == ()
    monkeypatch.setitem(configuration, "unroll_retry_threshold", 30000)
    monkeypatch.setattr(GlobalKernel, "_cache", {})
    cw1 = build().compile("ocrs")
    assert kernel_resources(cw1.path, cw1.src.symbol)["scratch"] == 0 and "-unroll-threshold=30000" in cw1.src.extra_flags


def test_row_sliced_variants_for_blocks_dof_masks_and_virtual_spaces():
    """The sliced wrapper's other shapes cross-compile without scratch: vector-valued blocks (row length table), per-dof
    lgmaps (row / column bit masks), and a matrix loop over an extruded subset with a direct argument and the layer argument."""
    from firedrake_amd.codegen import select_mode
    from firedrake_amd.compilation import kernel_resources
    from mixed_cases import vector_p1_elasticity_kernel
    nodes, ele = op2.Set(40), op2.Set(2)
    cm = op2.Map(ele, nodes, 4, np.arange(8))
    x = op2.Dat(nodes ** 3)
    lg = np.arange(120, dtype=np.int32)
    for unroll in (False, True):
        mat = op2.Mat(op2.Sparsity((nodes ** 3, nodes ** 3), [(cm, cm, None)]))
        pl = op2.LegacyParloop(vector_p1_elasticity_kernel(3), ele, mat(op2.INC, (cm, cm), lgmaps=(lg, lg), unroll_map=unroll), x(op2.READ, cm))
        assert select_mode(pl.global_kernel) == "ocrs"
        cw = pl.global_kernel.compile("ocrsp")
        assert ("_rmask" in cw.src.source) == unroll and "_rowlen" in cw.src.source
        assert kernel_resources(cw.path, cw.src.symbol)["scratch"] == 0
    base = op2.Set(4)
    ext = op2.ExtrudedSet(base, layers=5)
    en = op2.Set(6 * 5)
    em = op2.Map(ext, en, 6, np.arange(24) % 25, offset=[1] * 6)
    emat = op2.Mat(op2.Sparsity((en ** 1, en ** 1), [(em, em, None)]))
    w = op2.Dat(ext)
    k = op2.Kernel("static void pk(double *A, const double *w, int layer) { for (int i = 0; i < 36; ++i) A[i] += w[0] + layer; }", "pk")
    pl = op2.LegacyParloop(k, op2.Subset(ext, [0, 2]), emat(op2.INC, (em, em)), w(op2.READ), pass_layer_arg=True)
    assert select_mode(pl.global_kernel) == "ocr"             # 6 rows: whole-entity instances; both shapes exist on virtual spaces
    for mode in ("ocrs", "ocr"):
        cw = pl.global_kernel.compile(mode)
        assert "subset_indices[fd_col]" in cw.src.source and kernel_resources(cw.path, cw.src.symbol)["scratch"] == 0


def test_wrapper_shape_selection_table():
    """Which wrapper a matrix loop gets (codegen.select_mode): whole-entity owner-computes-rows for small scalar element
    matrices, row-sliced from 8 scalar rows and for every vector-valued / per-dof-lgmap matrix, both also over subsets and
    extruded sets (interior facets: row-sliced on maps of twice the arity); the direct or the staged wrapper for what neither covers
    (oversized element matrices, a second output)."""
    from firedrake_amd.codegen import select_mode
    from firedrake_amd.configuration import configuration
    assert configuration["ocr_sliced_min_arity"] == 8
    nodes, ele = op2.Set(64), op2.Set(2)
    x = op2.Dat(nodes ** 3)
    lg = np.arange(64, dtype=np.int32)

    def mode(ar, dim=1, iterset=ele, unroll=False, extra=(), **kw):
        m = op2.Map(iterset if not isinstance(iterset, op2.Subset) else iterset.superset, nodes, ar, np.arange(2 * ar) % 64,
                    **({"offset": [1] * ar} if getattr(iterset, "_extruded", False) else {}))
        mat = op2.Mat(op2.Sparsity((nodes ** dim, nodes ** dim), [(m, m, None)]))
        n = ar * dim
        k = op2.Kernel(f"static void k{n}(double *A, const double *x) {{ for (int i = 0; i < {n * n}; ++i) A[i] += x[0]; }}", f"k{n}")
        lgs = (np.arange(64 * dim, dtype=np.int32),) * 2 if unroll else (lg, lg)
        pl = op2.LegacyParloop(k, iterset, mat(op2.INC, (m, m), lgmaps=lgs, unroll_map=unroll), x(op2.READ, m), *extra, **kw)
        return select_mode(pl.global_kernel)
    assert mode(4) == "ocr" and mode(6) == "ocr"                      # P1 tets, P2 triangles
    assert mode(8) == "ocrs" and mode(10) == "ocrs"                   # Q1 hexahedra, P2 tets
    assert mode(3, dim=2) == "ocrs" and mode(4, dim=3) == "ocrs"      # vector-valued blocks: sliced whatever the size
    assert mode(4, dim=3, unroll=True) == "ocrs"                      # per-dof lgmaps
    assert mode(40) == "staged"                                       # 1600 entries: beyond the register budget of both (the staged
    #                                                                   wrapper then scatters the matrix entry by entry)
    ext = op2.ExtrudedSet(op2.Set(2), layers=4)
    assert mode(6, iterset=ext) == "ocr" and mode(8, iterset=ext) == "ocrs"
    assert mode(6, iterset=ext, iteration_region=op2.ON_INTERIOR_FACETS) == "ocrs"   # two stacked cells: maps of twice the arity
    assert mode(6, iterset=op2.Subset(ele, [1])) == "ocr"
    out = op2.Dat(nodes)
    m4 = op2.Map(ele, nodes, 4, np.arange(8))
    assert mode(4, extra=(out(op2.INC, m4),)) in ("staged", "direct")  # a second output: no redundant instances
