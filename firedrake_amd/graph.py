"""hipGraph capture of an assembly step (include/fdhip.h: fd_graph_*).

For launch-bound sizes (BASELINE config C1: 8192 cells) the reference spends its time in per-parloop Python
and ctypes overhead (pyop2/parloop.py:203-232).  A step whose plans and device mirrors
are already built consists only of stream-ordered work (memsets, wrapper kernels, BC kernels), so it can be
recorded once and replayed with a single host call."""
import ctypes
import gc

from . import _lib


class CapturedStep:
    def __init__(self, fn, warmup=2):
        """``fn()`` must only enqueue device work once warmed up (no host<->device copies, no allocation)."""
        for _ in range(warmup):
            fn()                      # builds plans / tables / uploads data
        gc.collect()                  # (finalisers that release device memory run now rather than inside the capture; one that does
        _lib.call("fd_device_sync")   #  arrive there is parked by the library until fd_graph_end: hipFree would invalidate the capture)
        h = ctypes.c_void_p()
        _lib.call("fd_graph_begin", ctypes.byref(h))
        self.h = None
        try:
            fn()
        finally:
            try:
                _lib.call("fd_graph_end", h)
            except _lib.FDHipError:
                _lib.load().fd_graph_free(h)      # (a capture that did not end in a graph: its stream goes with it)
                raise
        self.h = h.value

    def __call__(self):
        _lib.call("fd_graph_launch", self.h, None)

    def sync(self):
        _lib.call("fd_graph_sync", self.h)

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_graph_free(self.h)
        except Exception:
            pass
