"""Compile-time budgets of the benchmark wrappers (hipcc cross-compiles gfx950 without a GPU): the register / scratch / LDS figures
hipcc reports for the code objects the bench runs.  A wrapper that starts spilling, or loses a resident wavefront to registers, is a
performance regression no parity test sees."""
import numpy as np
import pytest

from firedrake_amd import forms
from firedrake_amd.compilation import kernel_resources
from firedrake_amd.kernel import DatKernelArg, GlobalKernel, MapKernelArg, MatKernelArg
from firedrake_amd.op2types import INC, READ

f64 = np.dtype("float64")


def _poisson(dim, degree):
    nd = {(2, 1): 3, (3, 1): 4, (3, 2): 10}[(dim, degree)]
    cm = MapKernelArg(nd)
    xm = cm if degree == 1 else MapKernelArg(dim + 1)
    kres = forms.poisson_residual_kernel(dim, degree).with_signature([INC, READ, READ, READ], [f64] * 4)
    kjac = forms.poisson_jacobian_kernel(dim, degree).with_signature([INC, READ], [f64] * 2)
    res = GlobalKernel(kres, [DatKernelArg((1,), cm), DatKernelArg((dim,), xm), DatKernelArg((1,), cm), DatKernelArg((1,), cm)])
    jac = GlobalKernel(kjac, [MatKernelArg(((1,), (1,)), (cm, cm), lgmaps=True), DatKernelArg((dim,), xm)])
    return res, jac


@pytest.mark.parametrize("degree,mode,max_vgprs", [(1, "ocr", 80), (1, "ocrp", 80), (2, "ocrs", 96), (2, "ocrsp", 96)])
def test_jacobian_wrappers_keep_their_element_tensor_in_registers(degree, mode, max_vgprs):
    """C2 / C5 Jacobians, hinted and derived row orders: no scratch, at most ``max_vgprs`` registers (P1: three 512-lane groups per CU = six wavefronts per SIMD, which 80 registers allow; P2 row-sliced: five wavefronts per SIMD, LDS is what limits it)."""
    _, jac = _poisson(3, degree)
    cw = jac.compile(mode)
    res = kernel_resources(cw.path, cw.src.symbol)
    assert res["scratch"] == 0 and res["vgpr_spill"] == 0
    assert res["vgprs"] + res.get("agprs", 0) <= max_vgprs, res


@pytest.mark.parametrize("degree,max_vgprs,min_occupancy", [(1, 64, 8), (2, 96, 5)])
def test_residual_wrappers_stay_within_their_occupancy_step(degree, max_vgprs, min_occupancy):
    """The staged residuals: P1 at full occupancy without scratch; P2 takes the occupancy-directed variant (kernel.py:
    _occupancy_variant), which may trade up to FDHIP_AUTO_OCCUPANCY_SCRATCH bytes of scratch per lane for a wavefront per SIMD."""
    from firedrake_amd.configuration import configuration
    res_k, _ = _poisson(3, degree)
    cw = res_k.compile("staged")
    res = kernel_resources(cw.path, cw.src.symbol)
    assert res["scratch"] <= (0 if degree == 1 else configuration["auto_occupancy_scratch"]), res
    assert res["vgprs"] <= max_vgprs and res["occupancy"] >= min_occupancy, res


def test_q4_tensor_wrappers_hold_three_wavefronts_per_simd():
    """C3: the MFMA matrix wrapper (64 accumulator registers per lane) and the action fit 168 registers -- three wavefronts per SIMD,
    the occupancy the measured 0.66 of the fp64 MFMA peak was taken at.  The action is COMPILED for three wavefronts (codegen:
    __launch_bounds__(128, 3), profiles/r4n_action_variants.txt) and may park a few registers in scratch for it."""
    from firedrake_amd import mesh as fmesh
    from firedrake_amd.codegen import generate_tensor_wrapper
    from firedrake_amd.compilation import compile_hip
    prob = forms.HelmholtzQ4Problem(fmesh.make_extruded_hex_mesh(1, 1, 4, perturb=0.0), bcs=True)
    for loop in (prob.jac_loop, prob.act_loop):
        src = generate_tensor_wrapper(loop.global_kernel)
        res = kernel_resources(compile_hip(src.source, src.symbol), src.symbol)
        assert res["scratch"] <= (32 if src.mode == "tp_action" else 0) and res["occupancy"] >= 3 and res["vgprs"] <= 168, (src.symbol, res)


def test_wider_tensor_descriptors_compile_without_scratch():
    """The Q3 Newton Jacobian with a coefficient gradient and the (Q2)^3 elasticity matrix -- nine scalar blocks in one workgroup, 144
    accumulator registers, compiled for two wavefronts per SIMD -- stay out of scratch memory; so does the scalar Q4 matrix kernel
    whose index arithmetic the widening touched (unsigned workgroup ids: signed ones cost 20 %, profiles/r5j_ab_c3_tree.txt)."""
    from firedrake_amd import mesh as fmesh
    from firedrake_amd.codegen import generate_tensor_wrapper, tensor_matrix_groups
    from firedrake_amd.compilation import compile_hip
    el = forms.ElasticityHexProblem(fmesh.make_extruded_hex_mesh(1, 1, 2, perturb=0.0), bcs=True)
    nd = forms.NonlinearDiffusionHexProblem(fmesh.make_extruded_hex_mesh(1, 1, 3, perturb=0.0), bcs=True)
    src = generate_tensor_wrapper(el.jac_loop.global_kernel)
    assert tensor_matrix_groups(src.tp, 3) == src.tp["matrix_groups"]                  # fused pairs
    res = kernel_resources(compile_hip(src.source, src.symbol), src.symbol)
    assert res["scratch"] == 0 and res["occupancy"] >= 2 and res["vgprs"] + res.get("agprs", 0) <= 256, res
    src = generate_tensor_wrapper(nd.jac_loop.global_kernel)
    res = kernel_resources(compile_hip(src.source, src.symbol), src.symbol)
    assert res["scratch"] == 0 and res["occupancy"] >= 3, res
    for loop in (el.act_loop, nd.act_loop):
        src = generate_tensor_wrapper(loop.global_kernel)
        res = kernel_resources(compile_hip(src.source, src.symbol), src.symbol)
        assert res["scratch"] == 0, (src.symbol, res)
