"""Jacobi-preconditioned conjugate gradients on the device, written against the product's own pointwise API
(op2.Mat.mult = fd_csr_spmv, Mat.get_diagonal, Dat.inner / axpy and direct parloops).  Test infrastructure: the reference's
regression tests solve their variational problems with PETSc (tests/firedrake/regression/test_helmholtz.py:50:
``solve(a == L, sol, solver_parameters={'ksp_type': 'cg'})``); solvers are outside this repository's scope,
so the threshold tests bring this small Krylov loop.  Serial (one rank)."""
from firedrake_amd import op2

_K = {}


def _kernel(name, code):
    if name not in _K:
        _K[name] = op2.Kernel(code, name)
    return _K[name]


def cg(A, b, x, rtol=1e-13, maxiter=5000):
    """Solve A x = b (A symmetric positive definite op2.Mat, b and x op2.Dats on its row space); x holds the initial guess.
    Returns (iterations, final relative residual)."""
    ds = b.dataset
    s = ds.set
    r, z, p, Ap, dinv = (op2.Dat(ds) for _ in range(5))
    A.get_diagonal(dinv)
    op2.par_loop(_kernel("cg_recip", "static void cg_recip(double *d) { d[0] = 1.0 / d[0]; }"), s, dinv(op2.RW))
    precond = _kernel("cg_precond", "static void cg_precond(double *z, const double *r, const double *d) { z[0] = r[0] * d[0]; }")
    xpay = _kernel("cg_xpay", "static void cg_xpay(double *p, const double *z, const double *beta) { p[0] = z[0] + beta[0] * p[0]; }")
    A.mult(x, Ap)
    b.copy(r)
    r.axpy(-1.0, Ap)
    bnorm = b.norm or 1.0
    op2.par_loop(precond, s, z(op2.WRITE), r(op2.READ), dinv(op2.READ))
    z.copy(p)
    rz = r.inner(z)
    it, res = 0, r.norm / bnorm
    while it < maxiter and res > rtol:
        A.mult(p, Ap)
        alpha = rz / p.inner(Ap)
        x.axpy(alpha, p)
        r.axpy(-alpha, Ap)
        op2.par_loop(precond, s, z(op2.WRITE), r(op2.READ), dinv(op2.READ))
        rz_new = r.inner(z)
        beta = op2.Global(1, rz_new / rz)
        op2.par_loop(xpay, s, p(op2.RW), z(op2.READ), beta(op2.READ))
        rz = rz_new
        it += 1
        res = r.norm / bnorm
    return it, res
