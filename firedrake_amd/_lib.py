"""ctypes binding of libfdhip.so (the C ABI in include/fdhip.h).

This is the thin layer the north star asks for: Python host code reaches the HIP
kernels through plain C entry points, the way pyop2/global_kernel.py:443-456 reaches
its JIT-compiled wrapper through ``ctypes.CDLL``.  There is no CPU fallback: if the
library (or a GPU) is missing, calls raise :class:`FDHipError`.
"""
import ctypes
import os
from ctypes import (POINTER, byref, c_char_p, c_double, c_float, c_int, c_int32, c_int64,
                    c_size_t, c_uint16, c_void_p)

import numpy as _np

# include/fdhip.h: fd_nnz_t -- the type of CSR / accumulator row starts (64-bit: patterns beyond 2^31 entries)
NNZ_DTYPE = _np.dtype(_np.int64)
NNZ_BYTES = NNZ_DTYPE.itemsize

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libfdhip.so")


class FDHipError(RuntimeError):
    """Raised for every non-zero status from libfdhip.so (cf. pyop2 CompilationError /
    the reference's practice of surfacing failures as Python exceptions)."""


_lib = None

# name -> (restype, argtypes); must list EVERY symbol declared in include/fdhip.h
SIGNATURES = {
    "fd_version": (c_int, []),
    "fd_last_error": (c_char_p, []),
    "fd_device_count": (c_int, [POINTER(c_int)]),
    "fd_set_device": (c_int, [c_int]),
    "fd_device_info": (c_int, [c_int, c_char_p, c_size_t, POINTER(c_int), POINTER(c_size_t), POINTER(c_int)]),
    "fd_malloc": (c_int, [POINTER(c_void_p), c_size_t]),
    "fd_free": (c_int, [c_void_p]),
    "fd_memset": (c_int, [c_void_p, c_int, c_size_t, c_void_p]),
    "fd_memcpy_h2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_memcpy_d2h": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_memcpy_d2d": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "fd_stream_create": (c_int, [POINTER(c_void_p)]),
    "fd_stream_destroy": (c_int, [c_void_p]),
    "fd_stream_sync": (c_int, [c_void_p]),
    "fd_stream_set_default": (c_int, [c_void_p]),
    "fd_stream_get_default": (c_int, [POINTER(c_void_p)]),
    "fd_stream_wait_event": (c_int, [c_void_p, c_void_p]),
    "fd_device_sync": (c_int, []),
    "fd_event_create": (c_int, [POINTER(c_void_p)]),
    "fd_event_destroy": (c_int, [c_void_p]),
    "fd_event_record": (c_int, [c_void_p, c_void_p]),
    "fd_event_sync": (c_int, [c_void_p]),
    "fd_event_elapsed_ms": (c_int, [c_void_p, c_void_p, POINTER(c_float)]),
    "fd_trace_available": (c_int, []),
    "fd_trace_range_push": (c_int, [c_char_p]),
    "fd_trace_range_pop": (c_int, []),
    "fd_graph_begin": (c_int, [POINTER(c_void_p)]),
    "fd_graph_end": (c_int, [c_void_p]),
    "fd_graph_launch": (c_int, [c_void_p, c_void_p]),
    "fd_graph_launch_default": (c_int, [c_void_p]),
    "fd_graph_sync": (c_int, [c_void_p]),
    "fd_graph_free": (c_int, [c_void_p]),
    "fd_kernel_load": (c_int, [c_char_p, c_char_p, POINTER(c_void_p)]),
    "fd_kernel_create": (c_int, [c_char_p, c_char_p, c_char_p, c_char_p, POINTER(c_void_p)]),
    "fd_kernel_builtin": (c_int, [c_char_p, POINTER(c_void_p)]),
    "fd_kernel_free": (c_int, [c_void_p]),
    "fd_kernel_launch": (c_int, [c_void_p, c_int32, c_int32, POINTER(c_void_p), c_int, c_int, c_int, c_int,
                                 c_size_t, c_void_p]),
    "fd_kd_order": (c_int, [c_void_p, c_int, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_int32, POINTER(c_int32), c_void_p]),
    "fd_leaf_labels": (c_int, [c_void_p, c_int64, c_void_p, c_int32, c_void_p, c_void_p]),
    "fd_group_entities": (c_int, [c_void_p, c_int, c_int32, c_int32, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    "fd_invert_permutation": (c_int, [c_void_p, c_int32, c_void_p, c_void_p]),
    "fd_row_order_tables": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_row_entry_positions": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_row_entry_positions_masked": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_first_touch_order": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_comm_available": (c_int, []),
    "fd_comm_unique_id": (c_int, [c_void_p]),
    "fd_comm_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "fd_comm_info": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int)]),
    "fd_comm_free": (c_int, [c_void_p]),
    "fd_comm_allreduce": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p]),
    "fd_halo_create": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]),
    "fd_halo_free": (c_int, [c_void_p]),
    "fd_halo_g2l_begin": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fd_halo_g2l_end": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "fd_halo_l2g_begin": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fd_halo_l2g_end": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "fd_halo_wire_buffers": (c_int, [c_void_p, c_void_p, c_int, POINTER(c_void_p), POINTER(c_int64), POINTER(c_void_p),
                                     POINTER(c_int64)]),
    "fd_dat_fill_range": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p]),
    "fd_plan_create": (c_int, [c_void_p, c_int, c_int32, c_int32, c_int, c_void_p, POINTER(c_void_p)]),
    "fd_plan_create_blocks": (c_int, [c_void_p, c_int, c_void_p, c_int32, c_void_p, POINTER(c_void_p)]),
    "fd_ocrplan_create_ordered": (c_int, [c_void_p, c_int, c_int32, c_int32, c_void_p, c_int32, c_int, c_void_p, c_int32, c_void_p,
                                          c_void_p, POINTER(c_void_p)]),
    "fd_ocr_node_words": (c_int, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_int, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_ocr_node_diag": (c_int, [c_void_p, c_int64, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_ocr_pack_records": (c_int, [c_int64, c_int, POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), c_void_p, c_int, c_int, c_int,
                                    c_int, c_int, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fd_ocr_row_runs": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, POINTER(c_int32),
                                POINTER(c_int32), c_void_p]),
    "fd_plan_set_lane_order": (c_int, [c_void_p, c_int, c_void_p]),
    "fd_plan_block_starts": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int32)]),
    "fd_plan_info": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "fd_plan_arrays": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "fd_plan_free": (c_int, [c_void_p]),
    "fd_matplan_create": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int64, c_void_p, POINTER(c_void_p)]),
    "fd_matplan_zero_list": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_int64), POINTER(c_int64)]),
    "fd_matplan_info": (c_int, [c_void_p, POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int64)]),
    "fd_matplan_arrays": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "fd_matplan_free": (c_int, [c_void_p]),
    "fd_ocrplan_create": (c_int, [c_void_p, c_int, c_int32, c_int32, c_void_p, c_int32, c_int, c_void_p, POINTER(c_void_p)]),
    "fd_ocrplan_info": (c_int, [c_void_p, POINTER(c_int64), POINTER(c_int32)]),
    "fd_ocrplan_arrays": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    "fd_ocrplan_free": (c_int, [c_void_p]),
    "fd_ocrplan_create_sliced": (c_int, [c_void_p, c_int, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int, c_void_p,
                                         POINTER(c_void_p)]),
    "fd_ocrplan_create_paired": (c_int, [c_void_p, c_int, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_int, c_int, c_void_p,
                                         c_void_p, POINTER(c_void_p)]),
    "fd_ocrplan_pair_counts": (c_int, [c_void_p, c_int, c_int32, c_int32, c_void_p, c_int32, c_void_p, c_int32, c_void_p, c_void_p]),
    "fd_ocr_pack_records_rows": (c_int, [c_int64, c_int, POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32), c_void_p, c_int, c_int, c_int,
                                         c_int, c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "fd_ocrplan_sliced_arrays": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64)]),
    "fd_ocrplan_sliced_tables": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                         c_int, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "fd_gather_rows": (c_int, [c_void_p, c_int, c_void_p, c_int64, c_void_p, c_void_p]),
    "fd_csr_elem_row_offsets": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int, c_int, c_int, c_void_p, c_void_p]),
    "fd_csr_from_maps": (c_int, [c_int32, c_int32, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                 POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                 POINTER(c_void_p), POINTER(c_void_p),
                                 POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "fd_csr_from_maps_ex": (c_int, [c_int32, c_int32, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p),
                                    POINTER(c_int32), POINTER(c_int32), POINTER(c_int32), POINTER(c_int32),
                                    POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int32), POINTER(c_int32),
                                    POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                    POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "fd_csr_expand_blocks": (c_int, [c_int32, c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p),
                                     POINTER(c_void_p), c_void_p]),
    "fd_csr_elem_offsets": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_int, c_int, c_int, c_void_p, c_void_p,
                                    c_void_p, c_void_p]),
    "fd_csr_diag_positions": (c_int, [c_void_p, c_void_p, c_void_p, c_int32, c_void_p, c_void_p]),
    "fd_csr_set_at": (c_int, [c_void_p, c_void_p, c_int32, c_double, c_void_p]),
    "fd_csr_set_diagonal": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_double, c_void_p]),
    "fd_csr_zero_rows": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int32, c_double, c_void_p]),
    "fd_csr_spmv": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_csr_split_mpiaij": (c_int, [c_int32, c_void_p, c_void_p, c_int32, c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64),
                                    POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_int64), c_void_p]),
    "fd_csr_split_values": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_csr_get_diagonal": (c_int, [c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "fd_dat_set_rows": (c_int, [c_void_p, c_int, c_void_p, c_int32, c_double, c_void_p]),
    "fd_dat_axpby": (c_int, [c_void_p, c_double, c_void_p, c_double, c_int64, c_void_p]),
    "fd_dat_copy_rows": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int32, c_void_p]),
}


def load():
    """dlopen libfdhip.so (after torch, so both share one HIP runtime) and type every symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FDHipError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "or `make -C firedrake_amd/csrc`")
    if os.environ.get("FDHIP_SKIP_TORCH", "0") != "1":      # single-process runs that never touch torch.distributed
        try:
            import torch  # noqa: F401  (loads libamdhip64 first; our NEEDED entry then binds to the same copy)
        except Exception:
            pass
    lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        msg = load().fd_last_error()
        raise FDHipError(f"libfdhip status {status}: {msg.decode() if msg else '?'}")


_TRACE = os.environ.get("FDHIP_PROFILE_CALLS", "0") == "2"       # 2: also name every C-ABI call before it runs
_PROFILE = os.environ.get("FDHIP_PROFILE_CALLS", "0") in ("1", "2")
call_times = {}          # FDHIP_PROFILE_CALLS=1: C-ABI entry point -> [calls, seconds incl. the device work it queued]


def call(name, *args):
    if _TRACE:        # debugging aid: name every C-ABI call before it runs (with AMD_SERIALIZE_KERNEL=3 the last line is the culprit)
        import sys
        print(f"[fdhip r{os.environ.get('RANK', '0')}] {name}", file=sys.stderr, flush=True)
    if _PROFILE and name != "fd_device_sync":
        # setup profiling aid: every call is followed by a device synchronisation, so its time includes the kernels it queued
        import time
        lib = load()
        t0 = time.perf_counter()
        check(getattr(lib, name)(*args))
        lib.fd_device_sync()
        e = call_times.setdefault(name, [0, 0.0])
        e[0] += 1
        e[1] += time.perf_counter() - t0
        return
    check(getattr(load(), name)(*args))


def profile_report(reset=True):
    """FDHIP_PROFILE_CALLS=1: 'name calls seconds' lines, most expensive first."""
    rows = sorted(call_times.items(), key=lambda kv: -kv[1][1])
    out = "\n".join(f"{k:<32} {v[0]:>6} {v[1]:9.4f} s" for k, v in rows)
    if reset:
        call_times.clear()
    return out


_gpu_ok = None


def require_gpu():
    """Fail loudly when there is no MI355X to run on (no CPU fallback anywhere)."""
    global _gpu_ok
    if _gpu_ok is None:
        n = c_int(0)
        try:
            st = load().fd_device_count(byref(n))
        except FDHipError:
            raise
        _gpu_ok = (st == 0 and n.value > 0)
    if not _gpu_ok:
        raise FDHipError("no HIP device visible: the firedrake_amd compute path needs an MI355X "
                         "(there is deliberately no CPU fallback)")
    return True


def gpu_available():
    try:
        return require_gpu()
    except FDHipError:
        return False
