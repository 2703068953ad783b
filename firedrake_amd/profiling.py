"""Per-parloop tracing hook (pyop2/profiling.py, pyop2/parloop.py:219-232).

The reference wraps every ``Parloop._compute`` in a PETSc log event ``Parloop_<iterset>_<kernel>`` and logs the kernel's flop
count to it (``PETSc.Log.logFlops(part.size * num_flops)``) so that ``-log_view`` attributes time and flops to each loop.
Here the same region names become **roctx ranges** (``fd_trace_range_push/pop`` -> ``roctxRangePushA/Pop``: ``rocprofv3
--marker-trace`` draws them above the wrapper kernels they launched) and a host-side table of calls, entities and flops per
region (``summary()``), the counterpart of the ``-log_view`` rows.  Off unless ``FDHIP_TRACE=1`` (configuration["trace"]):
a push/pop pair costs two ctypes calls per parloop."""
from contextlib import contextmanager

from . import _lib
from .configuration import configuration

_events = {}          # region name -> [calls, entities, flops]
_roctx = None


def _ranges_available():
    global _roctx
    if _roctx is None:
        try:
            _roctx = bool(_lib.load().fd_trace_available())
        except Exception:
            _roctx = False
    return _roctx


@contextmanager
def timed_region(name):
    """pyop2/profiling.py ``timed_region``: a named range around the launches of one parloop part."""
    if not configuration["trace"]:
        yield
        return
    pushed = _ranges_available() and _lib.load().fd_trace_range_push(name.encode()) == 0
    try:
        yield
    finally:
        if pushed:
            _lib.load().fd_trace_range_pop()


def log_flops(name, entities, flops):
    """PETSc.Log.logFlops analogue: attribute ``flops`` (and ``entities`` iterations) to region ``name``."""
    if configuration["trace"]:
        e = _events.setdefault(name, [0, 0, 0.0])
        e[0] += 1
        e[1] += int(entities)
        e[2] += float(flops)


def summary():
    """{region: {"calls", "entities", "flops"}} since the last ``reset()``."""
    return {k: {"calls": v[0], "entities": v[1], "flops": v[2]} for k, v in _events.items()}


def reset():
    _events.clear()
