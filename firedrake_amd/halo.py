"""Halo exchange of shared DoFs as an RCCL neighbour all-to-all over xGMI, driven through the C ABI.

Concrete counterpart of firedrake/halo.py:87-172 (a PetscSF over DoFs, ``bcastBegin/End`` for owner->ghost with
MPI.REPLACE and ``reduceBegin/End`` for ghost->owner with SUM/MIN/MAX) behind the abstract interface of
pyop2/types/halo.py:4-56.  One process per GPU; the per-neighbour send/receive node lists come from the mesh partitioner
(mesh.HaloLists).  The exchange itself lives in libfdhip.so (csrc/fd_comm.hip: ``fd_halo_create`` /
``fd_halo_{g2l,l2g}_{begin,end}``): device-resident index lists, persistent packed buffers, pack kernel -> ONE grouped
``ncclSend``/``ncclRecv`` exchange on a side stream -> unpack kernel (=, +=, min, max), ordered by events so the host
never waits and the transfer overlaps the core-entity kernel exactly where the reference overlaps MPI with
``_compute(core_part)`` (pyop2/parloop.py:250-253).  Messages are O(1 MB) per neighbour: latency-bound,
hence one grouped exchange per Dat and no ring collective.

Wire selection (``Halo.wire``):

``rccl``   the library's own communicator (``fd_comm_create`` = ncclCommInitRank; the 128-byte id is broadcast through
           ``torch.distributed``, which is only the process launcher/rendezvous here).  Default whenever
           ``torch.distributed`` runs with the ``nccl`` backend.
``host``   non-NCCL backends (gloo: the CPU-launched multi-rank tests, ranks sharing one GPU): the same C-ABI pack /
           unpack and persistent device buffers, the packed rows bounced through host tensors.
(The CPU-only protocol tests replace this class by a host restatement of the exchange, tests/host_halo.py, installed
through ``set_halo_factory`` -- nothing in this module runs without a device.)
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

from .op2types import INC, MAX, MIN, WRITE

_OPS = {WRITE: 0, INC: 1, MIN: 2, MAX: 3}
DTYPE_CODE = {np.dtype("float64"): 0, np.dtype("float32"): 1, np.dtype("int32"): 2, np.dtype("uint32"): 3,
              np.dtype("int64"): 4, np.dtype("uint64"): 5}


def _dist():
    import torch.distributed as dist
    return dist


def world_size():
    try:
        dist = _dist()
        return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
    except ImportError:
        return 1


def attach_halo(space):
    """Give the node Set of ``space`` a Halo if the mesh is partitioned."""
    h = space.halo
    if h is None or h.nranks <= 1:
        space.node_set.halo = None
        return
    space.node_set.halo = _factory["halo"](h)


_factory = {"halo": None}


def set_halo_factory(cls):
    """Install the class ``attach_halo`` instantiates (default: ``Halo``).  Test hook: the CPU-only multi-rank protocol
    tests plug in a host restatement of the exchange (tests/host_halo.py)."""
    _factory["halo"] = cls or Halo


# ---- the process-wide RCCL communicator of the library ---------------------------------------------------------------
_comm = {"handle": None, "tried": False, "why": None}


def communicator():
    """The library's RCCL communicator over all ranks (created on first use, collectively), or None when the wire is
    not RCCL (non-NCCL torch backend, FDHIP_HALO_WIRE=host, or librccl.so missing -- the reason is kept in
    ``communicator_status()``)."""
    if _comm["tried"]:
        return _comm["handle"]
    _comm["tried"] = True
    dist = _dist()
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        _comm["why"] = "single process"
        return None
    want = os.environ.get("FDHIP_HALO_WIRE", "auto")
    if want == "host" or (want == "auto" and dist.get_backend() != "nccl"):
        _comm["why"] = f"wire=host (torch.distributed backend {dist.get_backend()})"
        return None
    from . import _lib
    lib = _lib.load()
    rank, nranks = dist.get_rank(), dist.get_world_size()
    ok = bool(lib.fd_comm_available())
    # every rank must take the same branch: agree on availability first
    flags = [None] * nranks
    dist.all_gather_object(flags, ok)
    if not all(flags):
        _comm["why"] = "librccl.so could not be bound on every rank: " + (lib.fd_last_error() or b"").decode()
        return None
    uid = (ctypes.c_ubyte * 128)()
    box = [None]
    if rank == 0:
        # a failure on rank 0 is broadcast too: every rank must leave this collective the same way
        try:
            _lib.call("fd_comm_unique_id", uid)
            box = [bytes(uid)]
        except _lib.FDHipError as exc:
            box = [str(exc)]
    dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], bytes):
        _comm["why"] = f"ncclGetUniqueId failed on rank 0: {box[0]}"
        raise _lib.FDHipError(_comm["why"])
    buf = (ctypes.c_ubyte * 128).from_buffer_copy(box[0])
    h = ctypes.c_void_p()
    _lib.call("fd_comm_create", buf, rank, nranks, ctypes.byref(h))
    _comm["handle"], _comm["why"] = h.value, "rccl"
    return h.value


def communicator_status():
    return _comm["why"]


class Halo:
    """pyop2/types/halo.py interface over the C-ABI exchange."""

    def __init__(self, lists):
        self.lists = lists
        self.rank, self.nranks = lists.rank, lists.nranks
        self._h = None
        self._host_idx = {}
        self._pending = {}

    # -- helpers
    def _neighbours(self):
        return sorted(set(self.lists.send) | set(self.lists.recv))

    def _lists(self, r):
        e = np.zeros(0, dtype=np.int32)
        return (np.ascontiguousarray(self.lists.send.get(r, e), dtype=np.int32),
                np.ascontiguousarray(self.lists.recv.get(r, e), dtype=np.int32))

    @property
    def wire(self):
        self._handle()
        return "rccl" if self._comm else "host"

    def _handle(self):
        """fd_halo_t of this Halo (created on first use; collective when the wire is RCCL)."""
        if self._h is None:
            from . import _lib
            self._comm = communicator()
            nb = self._neighbours()
            self._nb = nb
            pairs = [self._lists(r) for r in nb]
            n = len(nb)
            peers = (ctypes.c_int32 * max(n, 1))(*nb)
            sp = (ctypes.c_void_p * max(n, 1))(*[p[0].ctypes.data for p in pairs])
            rp = (ctypes.c_void_p * max(n, 1))(*[p[1].ctypes.data for p in pairs])
            ns = (ctypes.c_int32 * max(n, 1))(*[len(p[0]) for p in pairs])
            nr = (ctypes.c_int32 * max(n, 1))(*[len(p[1]) for p in pairs])
            h = ctypes.c_void_p()
            _lib.call("fd_halo_create", self._comm, n, peers, sp, ns, rp, nr, ctypes.byref(h))
            self._h = h.value
            self._nsend = [len(p[0]) for p in pairs]
            self._nrecv = [len(p[1]) for p in pairs]
        return self._h

    def __del__(self):
        try:
            if self._h:
                from . import _lib
                _lib.load().fd_halo_free(self._h)
        except Exception:
            pass

    @staticmethod
    def _code(dat):
        try:
            return DTYPE_CODE[np.dtype(dat.dtype)]
        except KeyError:
            raise TypeError(f"halo exchange is not implemented for Dats of type {dat.dtype}")

    # -- device exchange through the C ABI
    def _begin(self, dat, direction, op):
        from . import _lib
        h, code = self._handle(), self._code(dat)
        ptr = dat._dev_ptr(False)                              # begin only reads the Dat
        if direction == 0:
            _lib.call("fd_halo_g2l_begin", h, ptr, dat.cdim, code, None)
        else:
            _lib.call("fd_halo_l2g_begin", h, ptr, dat.cdim, code, op, None)
        if self._comm:
            return
        # external wire: bounce the packed rows through the host (gloo and friends)
        import torch
        dist = _dist()
        sb, rb, ns, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
        _lib.call("fd_halo_wire_buffers", h, ptr, direction, ctypes.byref(sb), ctypes.byref(ns), ctypes.byref(rb), ctypes.byref(nr))
        row = dat.cdim * np.dtype(dat.dtype).itemsize
        out_counts = self._nsend if direction == 0 else self._nrecv
        in_counts = self._nrecv if direction == 0 else self._nsend
        sendh = np.empty(ns.value * row, dtype=np.uint8)
        if sendh.nbytes:
            _lib.call("fd_memcpy_d2h", sendh.ctypes.data, sb.value, sendh.nbytes, None)     # synchronises the pack
        recvh = np.empty(nr.value * row, dtype=np.uint8)
        ops, so, ro = [], 0, 0
        for r, no, ni in zip(self._nb, out_counts, in_counts):
            if no:
                ops.append(dist.P2POp(dist.isend, torch.from_numpy(sendh[so * row:(so + no) * row]), r))
            if ni:
                ops.append(dist.P2POp(dist.irecv, torch.from_numpy(recvh[ro * row:(ro + ni) * row]), r))
            so += no
            ro += ni
        reqs = dist.batch_isend_irecv(ops) if ops else []
        self._pending[(id(dat), direction)] = (reqs, sendh, recvh, rb.value)

    def _end(self, dat, direction, op):
        from . import _lib
        h, code = self._handle(), self._code(dat)
        if not self._comm:
            reqs, sendh, recvh, rbuf = self._pending.pop((id(dat), direction))
            for q in reqs:
                q.wait()
            if recvh.nbytes:
                _lib.call("fd_memcpy_h2d", rbuf, recvh.ctypes.data, recvh.nbytes, None)
                _lib.call("fd_stream_sync", None)
        ptr = dat._dev_ptr(True)
        if direction == 0:
            _lib.call("fd_halo_g2l_end", h, ptr, dat.cdim, code, None)
        else:
            _lib.call("fd_halo_l2g_end", h, ptr, dat.cdim, code, op, None)

    # -- pyop2 Halo interface
    def global_to_local_begin(self, dat, insert_mode):
        """owner -> ghost broadcast (firedrake/halo.py:125-131)."""
        self._begin(dat, 0, 0)

    def global_to_local_end(self, dat, insert_mode):
        self._end(dat, 0, 0)

    def local_to_global_begin(self, dat, insert_mode):
        """ghost -> owner reduction with SUM/MIN/MAX (firedrake/halo.py:141-172)."""
        self._begin(dat, 1, _OPS[insert_mode])

    def local_to_global_end(self, dat, insert_mode):
        self._end(dat, 1, _OPS[insert_mode])

    def fill_ghosts(self, dat, access_mode):
        """pyop2/types/dat.py:631-636: before an INC/MIN/MAX loop the ghost region is set to the identity of the
        access mode (0 / largest / lowest value OF THE DAT'S DTYPE, ``dtype_limits``) so the reverse reduction only
        carries this loop's contributions."""
        n0, n1 = dat.dataset.size, dat.dataset.total_size
        if n1 == n0:
            return
        from . import _lib
        code = self._code(dat)                                 # raises before any memory is touched
        kind = {INC: 0, MIN: 1, MAX: 2}[access_mode]
        _lib.call("fd_dat_fill_range", dat._dev_ptr(True), n0 * dat.cdim, (n1 - n0) * dat.cdim, code, kind, None)


def allreduce_global(glob, access, comm=None):
    """pyop2/parloop.py:411-442: MPI_Iallreduce of INC/MIN/MAX Globals.  With the RCCL wire the reduction runs on the
    Global's device buffer (``fd_comm_allreduce``, stream ordered, no host round trip); otherwise through
    ``torch.distributed`` on the host copy."""
    try:
        dist = _dist()
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
            return
    except ImportError:
        return
    op = {INC: 1, MIN: 2, MAX: 3}[access]
    c = communicator()
    code = DTYPE_CODE.get(np.dtype(glob.dtype))
    if c and code is not None:
        from . import _lib
        _lib.call("fd_comm_allreduce", c, glob._dev_ptr(True), int(np.prod(glob._host.shape)), code, op, None)
        return
    import torch
    rop = {1: dist.ReduceOp.SUM, 2: dist.ReduceOp.MIN, 3: dist.ReduceOp.MAX}[op]
    host = glob._to_host()
    t = torch.from_numpy(np.ascontiguousarray(host.reshape(-1)).copy())
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t, op=rop)
    glob._host_rw()[...] = t.cpu().numpy().reshape(host.shape)


_factory["halo"] = Halo
