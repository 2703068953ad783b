"""The LAPACK callables of generated kernels, `inverse(Aout, A, N)` and `solve(out, A, B, N)` (pyop2/codegen/rep2loopy.py:
108-199, pyop2/codegen/c/inverse.c:20-47, solve.c:18-51): mirrors tests/pyop2/test_callables.py:85-126 -- the same 2x2
fixtures and the same check against numpy.linalg -- with the C text loopy emits for `B[:,:] = inverse(A[:,:])` and
`x[:] = solve(A[:,:], b[:])` written out (loopy is not installed here), plus per-entity matrices that need pivoting.
CPU: the oracle's restatement (oracle/callables.h) and the generated direct wrapper on the host; GPU: csrc/fd_callables.h."""
import numpy as np
import pytest

from firedrake_amd import op2
from helpers import oracle_run

INV = """
static void callable_kernel(double *__restrict__ B, double const *__restrict__ A)
{
  inverse(&(B[0]), &(A[0]), 2);
}
"""
SOLVE = """
static void callable_kernel2(double *__restrict__ x, double const *__restrict__ A, double const *__restrict__ b)
{
  solve(&(x[0]), &(A[0]), &(b[0]), 2);
}
"""
INV5 = """
static void inv5(double *__restrict__ B, double const *__restrict__ A, double *__restrict__ x, double const *__restrict__ b)
{
  inverse(&(B[0]), &(A[0]), 5);
  solve(&(x[0]), &(A[0]), &(b[0]), 5);
}
"""


def _fixtures():
    s = op2.Set(1)
    zero_mat = op2.Dat(s ** (2, 2), [[0.0, 0.0], [0.0, 0.0]], np.float64)
    inv_mat = op2.Dat(s ** (2, 2), [[1.0, 2.0], [3.0, 4.0]], np.float64)
    zero_vec = op2.Dat(s ** (2, 1), [0.0, 0.0], np.float64)
    solve_mat = op2.Dat(s ** (2, 2), [[2.0, 1.0], [-3.0, 2.0]], np.float64)
    solve_vec = op2.Dat(s ** (2, 1), [1.0, 0.0], np.float64)
    return s, zero_mat, inv_mat, zero_vec, solve_mat, solve_vec


def _batch(n=257, seed=3, N=5):
    """n NxN systems; every third one has a zero leading pivot, every fifth a tiny one (partial pivoting matters)."""
    rng = np.random.default_rng(seed)
    A = rng.standard_normal((n, N, N)) + 3.0 * np.eye(N)
    A[::3, 0, 0] = 0.0
    A[::5, 1, 1] = 1e-14
    b = rng.standard_normal((n, N))
    return A, b


def test_reference_fixtures_on_the_oracle():
    s, zero_mat, inv_mat, zero_vec, solve_mat, solve_vec = _fixtures()
    got = oracle_run(op2.Kernel(INV, "callable_kernel"), s, zero_mat(op2.WRITE), inv_mat(op2.READ))[0]
    assert np.allclose(np.linalg.inv(np.array(inv_mat.data_ro).reshape(2, 2)), got.reshape(2, 2))
    got = oracle_run(op2.Kernel(SOLVE, "callable_kernel2"), s, zero_vec(op2.WRITE), solve_mat(op2.READ), solve_vec(op2.READ))[0]
    assert np.allclose(np.linalg.solve(np.array(solve_mat.data_ro).reshape(2, 2), np.array(solve_vec.data_ro).reshape(2)), got.reshape(2))


def test_pivoting_batch_on_the_oracle_and_the_host_wrapper():
    from hostsim import run_direct
    A, b = _batch()
    s = op2.Set(len(A))
    dA, db = op2.Dat(s ** (5, 5), A, np.float64), op2.Dat(s ** 5, b, np.float64)
    dB, dx = op2.Dat(s ** (5, 5), None, np.float64), op2.Dat(s ** 5, None, np.float64)
    k = op2.Kernel(INV5, "inv5")
    args = (dB(op2.WRITE), dA(op2.READ), dx(op2.WRITE), db(op2.READ))
    ref = oracle_run(k, s, *args)
    assert np.abs(ref[0].reshape(-1, 5, 5) - np.linalg.inv(A)).max() <= 1e-9 * np.abs(np.linalg.inv(A)).max()
    assert np.abs(ref[2].reshape(-1, 5) - np.linalg.solve(A, b[..., None])[..., 0]).max() <= 1e-9 * np.abs(ref[2]).max()
    got = run_direct(op2.LegacyParloop(k, s, *args))
    assert np.abs(got[0].reshape(-1) - ref[0].reshape(-1)).max() <= 1e-12 * np.abs(ref[0]).max()
    assert np.abs(got[2].reshape(-1) - ref[2].reshape(-1)).max() <= 1e-12 * np.abs(ref[2]).max()


@pytest.mark.gpu
def test_reference_fixtures_and_batch_on_the_gpu():
    s, zero_mat, inv_mat, zero_vec, solve_mat, solve_vec = _fixtures()
    op2.par_loop(op2.Kernel(INV, "callable_kernel"), s, zero_mat(op2.WRITE), inv_mat(op2.READ))
    assert np.allclose(np.linalg.inv(np.array(inv_mat.data_ro).reshape(2, 2)), np.array(zero_mat.data_ro).reshape(2, 2))
    op2.par_loop(op2.Kernel(SOLVE, "callable_kernel2"), s, zero_vec(op2.WRITE), solve_mat(op2.READ), solve_vec(op2.READ))
    assert np.allclose(np.linalg.solve(np.array(solve_mat.data_ro).reshape(2, 2), np.array(solve_vec.data_ro).reshape(2)),
                       np.array(zero_vec.data_ro).reshape(2))
    for N in (5, 9):                     # unrolled register version / out-of-line version with run-time loops
        A, b = _batch(4099, N=N)
        s = op2.Set(len(A))
        dA, db = op2.Dat(s ** (N, N), A, np.float64), op2.Dat(s ** N, b, np.float64)
        dB, dx = op2.Dat(s ** (N, N), None, np.float64), op2.Dat(s ** N, None, np.float64)
        k = op2.Kernel(INV5.replace(", 5)", f", {N})").replace("inv5", f"inv{N}"), f"inv{N}")
        args = (dB(op2.WRITE), dA(op2.READ), dx(op2.WRITE), db(op2.READ))
        ref = oracle_run(k, s, *args)
        op2.par_loop(k, s, *args)
        assert np.abs(np.array(dB.data_ro).reshape(-1) - ref[0].reshape(-1)).max() <= 1e-11 * np.abs(ref[0]).max()
        assert np.abs(np.array(dx.data_ro).reshape(-1) - ref[2].reshape(-1)).max() <= 1e-11 * np.abs(ref[2]).max()
