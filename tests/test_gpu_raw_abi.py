"""GPU: drive libfdhip.so through nothing but its C ABI (ctypes), the way a non-Python host would:
allocate, upload, look up a wrapper kernel, launch with the positional (start, end, *args) convention,
time with events, download.  Also builds a plan and a CSR pattern from raw device pointers."""
import ctypes

import numpy as np
import pytest

from firedrake_amd import _lib

pytestmark = pytest.mark.gpu


def _dev(arr):
    p = ctypes.c_void_p()
    _lib.call("fd_malloc", ctypes.byref(p), arr.nbytes)
    _lib.call("fd_memcpy_h2d", p, arr.ctypes.data, arr.nbytes, None)
    _lib.call("fd_stream_sync", None)
    return p


def test_axpy_builtin_through_raw_abi():
    n = 100003
    x, y, a = np.arange(n, dtype=np.float64), np.ones(n), np.array([2.5])
    dx, dy, da = _dev(x), _dev(y), _dev(a)
    k = ctypes.c_void_p()
    _lib.call("fd_kernel_builtin", b"wrap_fd_axpy", ctypes.byref(k))
    e0, e1 = ctypes.c_void_p(), ctypes.c_void_p()
    _lib.call("fd_event_create", ctypes.byref(e0))
    _lib.call("fd_event_create", ctypes.byref(e1))
    args = (ctypes.c_void_p * 3)(dy, dx, da)
    _lib.call("fd_event_record", e0, None)
    _lib.call("fd_kernel_launch", k, 7, n - 5, args, 3, 256, 256, -1, 0, None)     # sub-range [7, n-5)
    _lib.call("fd_event_record", e1, None)
    _lib.call("fd_event_sync", e1)
    ms = ctypes.c_float()
    _lib.call("fd_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    assert ms.value >= 0
    out = np.empty(n)
    _lib.call("fd_memcpy_d2h", out.ctypes.data, dy, out.nbytes, None)
    ref = y.copy()
    ref[7:n - 5] += 2.5 * x[7:n - 5]
    assert np.array_equal(out, ref)
    with pytest.raises(_lib.FDHipError):
        _lib.call("fd_kernel_builtin", b"wrap_does_not_exist", ctypes.byref(k))
    for p in (dx, dy, da):
        _lib.call("fd_free", p)


def test_plan_and_csr_from_raw_pointers():
    rng = np.random.default_rng(0)
    ncell, nnode = 5000, 1500
    m = ((np.arange(ncell)[:, None] * nnode // ncell + rng.integers(0, 25, size=(ncell, 3))) % nnode).astype(np.int32)
    dm = _dev(m)
    plan = ctypes.c_void_p()
    _lib.call("fd_plan_create", dm, 3, 0, ncell, 512, None, ctypes.byref(plan))
    nb, mx, ll = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int64()
    _lib.call("fd_plan_info", plan, ctypes.byref(nb), ctypes.byref(mx), ctypes.byref(ll))
    assert nb.value == (ncell + 511) // 512
    assert mx.value == max(len(np.unique(m[b:b + 512])) for b in range(0, ncell, 512))
    rm = (ctypes.c_void_p * 1)(dm)
    nent, ar = (ctypes.c_int32 * 1)(ncell), (ctypes.c_int32 * 1)(3)
    nl = (ctypes.c_int32 * 1)(0)
    none = (ctypes.c_void_p * 1)(None)
    rp, ci, nnz = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64()
    _lib.call("fd_csr_from_maps", nnode, nnode, 1, 1, rm, rm, nent, ar, ar, nl, none, none,
              ctypes.byref(rp), ctypes.byref(ci), ctypes.byref(nnz), None)
    rowptr = np.empty(nnode + 1, dtype=_lib.NNZ_DTYPE)               # fd_nnz_t row starts
    _lib.call("fd_memcpy_d2h", rowptr.ctypes.data, rp, rowptr.nbytes, None)
    pairs = {(int(r), int(c)) for row in m for r in row for c in row} | {(i, i) for i in range(nnode)}
    assert nnz.value == len(pairs) and rowptr[-1] == nnz.value
    _lib.call("fd_plan_free", plan)
    for p in (dm, rp, ci):
        _lib.call("fd_free", p)
