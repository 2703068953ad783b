"""Device buffers: HBM allocations that stand in for the numpy buffers whose raw pointers the
reference passes to its wrapper (pyop2/types/dat.py:94-96, map.py:57-59, glob.py:32-33)."""
import ctypes

import numpy as np

from . import _lib


_pending_graph = [None]        # the graph.CapturedStep whose last replay nobody has waited for yet


def wait_pending_graph():
    """A replayed step runs on the graph's own (non-blocking) stream, which the synchronous copies of the NULL stream do not wait
    for: every host read of device memory (``DeviceBuffer.download``, the Dats' ``data_ro``) waits for the last replay first."""
    g = _pending_graph[0]
    if g is not None:
        _pending_graph[0] = None
        g.sync()


class DeviceBuffer:
    """An owned hipMalloc allocation (fd_malloc / fd_free)."""

    def __init__(self, nbytes: int):
        _lib.require_gpu()
        p = ctypes.c_void_p()
        try:
            _lib.call("fd_malloc", ctypes.byref(p), max(int(nbytes), 8))
        except _lib.FDHipError as exc:
            if "out of memory" not in str(exc).lower():
                raise
            # device memory held by unreachable objects that sit in reference cycles comes back with a collection: once, then again
            import gc
            gc.collect()
            _lib.call("fd_malloc", ctypes.byref(p), max(int(nbytes), 8))
        self.ptr = p.value
        self.nbytes = int(nbytes)
        self._owned = True

    @classmethod
    def wrap(cls, ptr: int, nbytes: int, owner=None, owned=True):
        """Adopt device memory allocated inside libfdhip (released with fd_free on GC), or -- ``owned=False`` --
        borrow memory that belongs to somebody else (the function-level seam, bridge.py)."""
        b = cls.__new__(cls)
        b.ptr, b.nbytes, b._owned = ptr, int(nbytes), bool(owned)
        return b

    @classmethod
    def from_numpy(cls, arr: np.ndarray):
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        b.upload(arr)
        return b

    def upload(self, arr: np.ndarray, stream=None):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= max(self.nbytes, 8)
        if arr.nbytes:
            _lib.call("fd_memcpy_h2d", self.ptr, arr.ctypes.data, arr.nbytes, stream)
            _lib.call("fd_stream_sync", stream)   # host buffer may be a temporary

    def download(self, dtype, shape):
        out = np.empty(shape, dtype=dtype)
        wait_pending_graph()
        if out.nbytes:
            _lib.call("fd_memcpy_d2h", out.ctypes.data, self.ptr, out.nbytes, None)
        return out

    def zero(self, stream=None):
        _lib.call("fd_memset", self.ptr, 0, self.nbytes, stream)

    def __del__(self):
        try:
            if getattr(self, "_owned", False) and self.ptr:
                _lib.load().fd_free(self.ptr)
                self.ptr = None
        except Exception:
            pass


def synchronize():
    _lib.call("fd_device_sync")


class Event:
    def __init__(self):
        p = ctypes.c_void_p()
        _lib.call("fd_event_create", ctypes.byref(p))
        self.h = p.value

    def record(self, stream=None):
        _lib.call("fd_event_record", self.h, stream)

    def sync(self):
        _lib.call("fd_event_sync", self.h)

    def elapsed_ms(self, later: "Event") -> float:
        ms = ctypes.c_float()
        _lib.call("fd_event_elapsed_ms", self.h, later.h, ctypes.byref(ms))
        return ms.value

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_event_destroy(self.h)
        except Exception:
            pass



class Stream:
    """A non-blocking HIP stream of the library (fd_stream_create).  The reference runs its parloops one after another on the host
    (pyop2/parloop.py:243-260); here a parloop is stream-ordered device work, and two loops that write different tensors and only
    READ what they share -- ``assemble(F)`` and ``assemble(J)`` of one Newton step -- can be put on two streams and share the
    device: the latency-bound staged residual fills the issue slots the LDS-bound Jacobian leaves idle.  ``with side.fork(): ...``
    runs the enclosed launches on ``side`` after everything enqueued so far; ``side.join()`` makes the launching stream wait for
    them.  Nothing here synchronises the host.  Inside a captured step (graph.CapturedStep) a fork becomes a second branch of the
    graph: at launch-bound sizes (config C1) the two assemblies' kernel chains run side by side."""

    _current = None      # the stream NULL-stream calls of the library resolve to (None = the HIP null stream)

    def __init__(self):
        p = ctypes.c_void_p()
        _lib.call("fd_stream_create", ctypes.byref(p))
        self.h = p.value
        self._fork, self._done = Event(), Event()

    def sync(self):
        _lib.call("fd_stream_sync", self.h)

    def wait(self, event: "Event"):
        _lib.call("fd_stream_wait_event", self.h, event.h)

    def fork(self):
        return _Forked(self)

    def join(self):
        """The launching stream waits for the work enqueued inside the last ``fork()`` block."""
        _lib.call("fd_stream_wait_event", None, self._done.h)           # (NULL = the stream launches go to right now)

    def __del__(self):
        try:
            if self.h:
                _lib.load().fd_stream_destroy(self.h)
                self.h = None
        except Exception:
            pass


class _Forked:
    def __init__(self, side):
        self.side = side

    def __enter__(self):
        s = self.side
        self.prev = Stream._current
        p = ctypes.c_void_p()
        _lib.call("fd_stream_get_default", ctypes.byref(p))
        self.prev_handle = p.value           # (not always prev.h: inside a capture the launching stream is the capturing one)
        s._fork.record()                     # on the launching stream: everything enqueued so far ...
        s.wait(s._fork)                      # ... precedes the side stream's work
        _lib.call("fd_stream_set_default", s.h)
        Stream._current = s
        return s

    def __exit__(self, *exc):
        s = self.side
        s._done.record()                     # (on the side stream: it is still the default)
        _lib.call("fd_stream_set_default", self.prev_handle)
        Stream._current = self.prev
        return False
