"""python tools/cycle_suite.py tests/test_gpu_x.py ...: every module run with the cycle collector OFF; afterwards a collection with DEBUG_SAVEALL
lists what only the collector could free -- objects that hold device memory (buffers, plans, maps, tensors) must not be among it."""
import collections, gc, os, sys
sys.path.insert(0, os.getcwd())
import pytest
from firedrake_amd import _lib
_lib.require_gpu()
for m in sys.argv[1:]:
    gc.collect()
    gc.disable()
    rc = pytest.main(["-q", "-m", "gpu", "-p", "no:cacheprovider", m, "--no-header", "-q"])
    gc.set_debug(gc.DEBUG_SAVEALL)
    gc.collect()
    hist = collections.Counter(type(o).__name__ for o in gc.garbage)
    interesting = {k: v for k, v in hist.items() if k in ("DeviceBuffer", "Plan", "Map", "Mat", "Sparsity", "Dat", "OcrPlan", "RowOrder", "LocalityOrder", "LegacyParloop", "Parloop", "MatPlan", "VirtualSpace", "Set", "ExtrudedSet", "Subset", "Global", "CompiledWrapper", "CapturedStep", "Halo")}
    print(f"CYCLES {m}: rc={rc} garbage objects of interest: {interesting}", flush=True)
    gc.set_debug(0)
    gc.garbage.clear()
    gc.enable()
