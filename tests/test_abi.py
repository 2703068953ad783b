"""CPU: the C-ABI library loads and exports every symbol include/fdhip.h declares."""
import os
import re

from firedrake_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "fdhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fd_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 35
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/fdhip.h but not exported"


def test_python_signatures_cover_header():
    assert sorted(_lib.SIGNATURES) == _declared()


def test_version_and_error_string():
    lib = _lib.load()
    assert lib.fd_version() >= 100
    assert isinstance(lib.fd_last_error(), (bytes, type(None)))


def test_no_cpu_fallback():
    """Without a GPU the compute path must fail loudly."""
    import numpy as np
    import pytest
    from firedrake_amd import op2
    if _lib.gpu_available():
        pytest.skip("GPU present")
    s = op2.Set(4)
    d = op2.Dat(s, np.arange(4.0))
    k = op2.Kernel("static void k(double *x) { x[0] += 1.0; }", "k_nofallback")
    with pytest.raises(_lib.FDHipError):
        op2.par_loop(k, s, d(op2.RW))


def test_fd_kernel_create_compiles_into_the_cache(tmp_path):
    """The in-library JIT (fd_kernel_create: hash -> hipcc --genco -> rename into the cache -> load).  Without a GPU the
    compile and cache steps run (hipcc cross-compiles gfx950) and the load step fails loudly."""
    import ctypes
    import pytest
    src = ('#include "fd_wrapper.h"\\nextern "C" __global__ void wrap_jit_probe(int start, int end, double *y, const double *x) '
           '{ for (int i = start + blockIdx.x*blockDim.x + threadIdx.x; i < end; i += gridDim.x*blockDim.x) y[i] += 2.0*x[i]; }\\n')
    h = ctypes.c_void_p()
    lib = _lib.load()
    rc = lib.fd_kernel_create(src.encode(), b"wrap_jit_probe", str(tmp_path).encode(), None, ctypes.byref(h))
    objs = [f for f in os.listdir(tmp_path) if f.endswith(".hsaco")]
    assert len(objs) == 1 and objs[0].startswith("wrap_jit_probe_c")          # compiled and cached either way
    assert not [f for f in os.listdir(tmp_path) if f.endswith(".tmp") or ".tmp." in f]
    if _lib.gpu_available():
        assert rc == 0 and h.value
        import numpy as np
        from firedrake_amd.device import DeviceBuffer
        x, y = np.arange(1000.0), np.ones(1000)
        dx, dy = DeviceBuffer.from_numpy(x), DeviceBuffer.from_numpy(y)
        arr = (ctypes.c_void_p * 2)(dy.ptr, dx.ptr)
        _lib.call("fd_kernel_launch", h.value, 0, 1000, arr, 2, 256, 256, -1, 0, None)
        assert np.array_equal(dy.download(np.float64, (1000,)), 1.0 + 2.0 * x)
        # second call: served from the cache
        h2 = ctypes.c_void_p()
        assert lib.fd_kernel_create(src.encode(), b"wrap_jit_probe", str(tmp_path).encode(), None, ctypes.byref(h2)) == 0
    else:
        assert rc != 0 and b"hipModuleLoad" in lib.fd_last_error()
    bad = ctypes.c_void_p()
    assert lib.fd_kernel_create(b"this is not HIP", b"wrap_bad", str(tmp_path).encode(), None, ctypes.byref(bad)) != 0
    assert b"hipcc failed" in lib.fd_last_error()
