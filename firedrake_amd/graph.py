"""hipGraph capture of an assembly step (include/fdhip.h: fd_graph_*).

For launch-bound sizes (BASELINE config C1: 8192 cells) the reference spends its time in per-parloop Python
and ctypes overhead (pyop2/parloop.py:203-232).  A step whose plans and device mirrors
are already built consists only of stream-ordered work (memsets, wrapper kernels, BC kernels), so it can be
recorded once and replayed with a single host call."""
import ctypes
import gc

from . import _lib, device


class CapturedStep:
    def __init__(self, fn, warmup=2, ordered=True):
        """``fn()`` must only enqueue device work once warmed up (no host<->device copies, no allocation).  ``ordered``: replays go to
        the stream eager launches go to (fd_graph_launch_default), so a replay is ordered with the parloops and copies before and
        after it like any other launch; False = the graph's own stream (host reads still wait for it, eager launches do not)."""
        self.ordered = bool(ordered)
        for _ in range(warmup):
            fn()                      # builds plans / tables / uploads data
        gc.collect()                  # (finalisers that release device memory run now rather than inside the capture; one that does
        _lib.call("fd_device_sync")   #  arrive there is parked by the library until fd_graph_end: hipFree would invalidate the capture)
        h = ctypes.c_void_p()
        _lib.call("fd_graph_begin", ctypes.byref(h))
        self.h = None
        from . import op2types
        op2types._capture_log = log = ({}, {})
        try:
            fn()
        finally:
            op2types._capture_log = None
            try:
                _lib.call("fd_graph_end", h)
            except _lib.FDHipError:
                _lib.load().fd_graph_free(h)      # (a capture that did not end in a graph: its stream goes with it)
                raise
        self.h = h.value
        # the graph holds ADDRESSES: what they point at stays alive as long as the graph does (the closure holds the tensors, maps
        # and plans of the step; the carriers it touched are held by name)
        self._fn = fn
        self._written = list(log[1].values())
        self._touched = list({**log[0], **log[1]}.values())

    def __call__(self):
        """Replay.  The coherence bookkeeping of the carriers ran when the step was recorded, so it is redone here: a Dat / Global the
        step touches whose HOST copy was written since (``dat.data[...] = ...``) is uploaded first (synchronously, to the same address),
        and the host copy of everything the step writes is stale afterwards (``data_ro`` waits for the replay and downloads again)."""
        if not self.ordered and any(not c._dev_valid or c._rw_handed for c in self._touched):
            device.wait_pending_graph()      # (an upload must not overtake the previous replay that may still read the buffer)
        for c in self._touched:
            c._dev_ptr(False)
        if self.ordered:
            _lib.call("fd_graph_launch_default", self.h)
        else:
            _lib.call("fd_graph_launch", self.h, None)
            device._pending_graph[0] = self  # (host reads of device memory wait for this replay: device.wait_pending_graph)
        for c in self._written:
            c._dev_valid = True
            c._host_valid = False
            c.dat_version += 1

    def sync(self):
        _lib.call("fd_graph_sync", self.h)
        if device._pending_graph[0] is self:
            device._pending_graph[0] = None

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_graph_free(self.h)
        except Exception:
            pass
