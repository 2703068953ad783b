"""python tools/leak_suite.py tests/test_gpu_x.py ...: every module run three times in ONE process; free device memory after passes 2 and 3 must equal that after pass 1 (what pass 1 keeps is per kernel: loaded code objects, the runtime's scratch)."""
import ctypes, gc, os, sys
sys.path.insert(0, os.getcwd())
import pytest
from firedrake_amd import _lib
hip = ctypes.CDLL("libamdhip64.so")
def free_mb():
    _lib.call("fd_device_sync")
    f = ctypes.c_size_t(); t = ctypes.c_size_t(); hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t)); return f.value / 2**20
mods = sys.argv[1:]
_lib.require_gpu()
print("start", round(free_mb()))
for m in mods:
    row = []
    for rep in range(3):
        rc = pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", "-x", m, "--no-header", "-q"])
        gc.collect()
        row.append(round(free_mb(), 1))
    print(f"LEAK {m}: free after pass 1/2/3 = {row}  delta(2->3) = {row[2]-row[1]:.1f} MB  rc={rc}", flush=True)
