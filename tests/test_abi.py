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
