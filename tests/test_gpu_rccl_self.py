"""GPU: the RCCL wire of csrc/fd_comm.hip on ONE device -- a communicator of one rank whose halo neighbour is the rank
itself (the situation of a periodic, unpartitioned direction).  Everything the multi-GPU exchange uses runs for real:
librccl.so bound at run time, ncclCommInitRank, the grouped ncclSend/ncclRecv on the side stream, the event ordering
against the compute stream, typed pack/unpack with REPLACE / SUM / MIN / MAX, several Dats in flight at once, and the
in-place ncclAllReduce of a Global.  (Two ranks cannot share a GPU under RCCL; the N > 1 run is
tests/test_multirank_gloo.py::test_partitioned_assembly_over_rccl, which needs >= 2 devices.)"""
import ctypes

import numpy as np
import pytest

from firedrake_amd import _lib
from firedrake_amd.device import DeviceBuffer

pytestmark = pytest.mark.gpu

DT = {np.dtype("float64"): 0, np.dtype("float32"): 1, np.dtype("int32"): 2, np.dtype("uint32"): 3, np.dtype("int64"): 4,
      np.dtype("uint64"): 5}


@pytest.fixture(scope="module")
def comm():
    lib = _lib.load()
    if not lib.fd_comm_available():
        pytest.fail("librccl.so could not be bound: " + lib.fd_last_error().decode())
    uid = (ctypes.c_ubyte * 128)()
    _lib.call("fd_comm_unique_id", uid)
    h = ctypes.c_void_p()
    _lib.call("fd_comm_create", uid, 0, 1, ctypes.byref(h))
    r, n = ctypes.c_int(), ctypes.c_int()
    _lib.call("fd_comm_info", h.value, ctypes.byref(r), ctypes.byref(n))
    assert (r.value, n.value) == (0, 1)
    yield h.value
    _lib.call("fd_comm_free", h.value)


def _halo(comm, send, recv):
    send, recv = (np.ascontiguousarray(a, dtype=np.int32) for a in (send, recv))
    peers = (ctypes.c_int32 * 1)(0)
    sp, rp = (ctypes.c_void_p * 1)(send.ctypes.data), (ctypes.c_void_p * 1)(recv.ctypes.data)
    ns, nr = (ctypes.c_int32 * 1)(len(send)), (ctypes.c_int32 * 1)(len(recv))
    h = ctypes.c_void_p()
    _lib.call("fd_halo_create", comm, 1, peers, sp, ns, rp, nr, ctypes.byref(h))
    return h.value


@pytest.mark.parametrize("dtype,cdim", [(np.float64, 1), (np.float64, 3), (np.float32, 2), (np.int32, 1), (np.uint32, 2), (np.int64, 1)])
def test_forward_and_reverse_exchange_through_rccl(comm, dtype, cdim):
    rng = np.random.default_rng(0)
    n, nh = 5000, 700
    perm = rng.permutation(n)
    send, recv = perm[:nh], perm[nh:2 * nh]            # "owned" nodes and the "ghost" copies they feed
    h = _halo(comm, send, recv)
    code = DT[np.dtype(dtype)]
    a = (rng.standard_normal((n, cdim)) * 100).astype(dtype) if np.dtype(dtype).kind == "f" else rng.integers(0, 1000, (n, cdim)).astype(dtype)
    d = DeviceBuffer.from_numpy(a)
    # forward: ghosts <- owners
    _lib.call("fd_halo_g2l_begin", h, d.ptr, cdim, code, None)
    _lib.call("fd_halo_g2l_end", h, d.ptr, cdim, code, None)
    exp = a.copy()
    exp[recv] = a[send]
    got = d.download(dtype, (n, cdim))
    assert np.array_equal(got, exp)
    # reverse with SUM / MIN / MAX: owners <- op(owners, ghosts)
    for op, f in ((1, lambda x, y: x + y), (2, np.minimum), (3, np.maximum)):
        cur = d.download(dtype, (n, cdim))
        _lib.call("fd_halo_l2g_begin", h, d.ptr, cdim, code, op, None)
        _lib.call("fd_halo_l2g_end", h, d.ptr, cdim, code, op, None)
        exp = cur.copy()
        exp[send] = f(cur[send], cur[recv])
        assert np.array_equal(d.download(dtype, (n, cdim)), exp)
    _lib.call("fd_halo_free", h)


def test_two_dats_in_flight_and_persistent_buffers(comm):
    """Parloop.global_to_local_begin posts every Dat before any end (parloop.py:354-363): two exchanges in flight, then
    the same halo reused for more steps (persistent packed buffers)."""
    rng = np.random.default_rng(1)
    n = 3000
    send, recv = np.arange(0, 400), np.arange(2000, 2400)
    h = _halo(comm, send, recv)
    a, b = rng.standard_normal((n, 1)), rng.standard_normal((n, 3))
    da, db = DeviceBuffer.from_numpy(a), DeviceBuffer.from_numpy(b)
    for step in range(3):
        _lib.call("fd_halo_g2l_begin", h, da.ptr, 1, 0, None)
        _lib.call("fd_halo_g2l_begin", h, db.ptr, 3, 0, None)
        with pytest.raises(_lib.FDHipError):
            _lib.call("fd_halo_g2l_begin", h, da.ptr, 1, 0, None)         # already in flight
        _lib.call("fd_halo_g2l_end", h, db.ptr, 3, 0, None)
        _lib.call("fd_halo_g2l_end", h, da.ptr, 1, 0, None)
        a[recv], b[recv] = a[send], b[send]
        assert np.array_equal(da.download(np.float64, a.shape), a) and np.array_equal(db.download(np.float64, b.shape), b)
        a[send] += 1.0
        da.upload(a)
    with pytest.raises(_lib.FDHipError):
        _lib.call("fd_halo_g2l_end", h, da.ptr, 1, 0, None)               # no matching begin
    _lib.call("fd_halo_free", h)


def test_allreduce_and_ghost_fill(comm):
    v = np.array([1.5, -2.0, 7.25])
    d = DeviceBuffer.from_numpy(v)
    for op in (1, 2, 3):
        _lib.call("fd_comm_allreduce", comm, d.ptr, 3, 0, op, None)       # one rank: the identity, through ncclAllReduce
    assert np.array_equal(d.download(np.float64, (3,)), v)
    for dtype, hi, lo in ((np.float64, np.finfo(np.float64).max, np.finfo(np.float64).min), (np.int32, 2 ** 31 - 1, -2 ** 31),
                          (np.uint32, 2 ** 32 - 1, 0), (np.float32, np.finfo(np.float32).max, np.finfo(np.float32).min)):
        buf = DeviceBuffer.from_numpy(np.full(40, 3, dtype=dtype))
        for kind, val in ((0, 0), (1, hi), (2, lo)):
            _lib.call("fd_dat_fill_range", buf.ptr, 10, 20, DT[np.dtype(dtype)], kind, None)
            got = buf.download(dtype, (40,))
            assert (got[:10] == 3).all() and (got[30:] == 3).all() and (got[10:30] == np.array(val).astype(dtype)).all()


def _halo_multi(comm, pairs):
    """halo with several neighbour entries (all of them this rank): pairs = [(send, recv), ...]"""
    pairs = [tuple(np.ascontiguousarray(a, dtype=np.int32) for a in pr) for pr in pairs]
    n = len(pairs)
    peers = (ctypes.c_int32 * n)(*([0] * n))
    sp = (ctypes.c_void_p * n)(*[p[0].ctypes.data for p in pairs])
    rp = (ctypes.c_void_p * n)(*[p[1].ctypes.data for p in pairs])
    ns = (ctypes.c_int32 * n)(*[len(p[0]) for p in pairs])
    nr = (ctypes.c_int32 * n)(*[len(p[1]) for p in pairs])
    h = ctypes.c_void_p()
    _lib.call("fd_halo_create", comm, n, peers, sp, ns, rp, nr, ctypes.byref(h))
    return h.value


@pytest.mark.parametrize("op", [1, 2, 3])
def test_owned_nodes_shared_with_several_neighbours(comm, op):
    """An owned node that two (here: three) neighbours hold as a ghost -- the plane of a slab one cell thick, the edges and
    corners of a block partition -- receives one contribution per neighbour in the reverse exchange; all of them must land
    (a single non-atomic launch over the concatenated send lists lost some)."""
    rng = np.random.default_rng(5)
    n, nh = 40000, 6000
    shared = rng.permutation(n // 2)[:nh]                         # owned nodes every "neighbour" keeps a ghost copy of
    ghosts = [n // 2 + k * nh + np.arange(nh) for k in range(3)]  # three disjoint ghost sets
    h = _halo_multi(comm, [(shared, g) for g in ghosts])
    a = rng.standard_normal((n, 2))
    d = DeviceBuffer.from_numpy(a)
    for rep in range(3):
        cur = d.download(np.float64, a.shape)
        _lib.call("fd_halo_l2g_begin", h, d.ptr, 2, 0, op, None)
        _lib.call("fd_halo_l2g_end", h, d.ptr, 2, 0, op, None)
        exp = cur.copy()
        for g in ghosts:
            exp[shared] = {1: np.add, 2: np.minimum, 3: np.maximum}[op](exp[shared], cur[g])
        got = d.download(np.float64, a.shape)
        assert np.allclose(got, exp, rtol=0, atol=1e-13), np.abs(got - exp).max()
    # the fused node-major combine (one launch, no atomics) applies a node's contributions in neighbour order: BITWISE the sums
    # taken neighbour by neighbour on the host
    start = d.download(np.float64, a.shape)
    _lib.call("fd_halo_l2g_begin", h, d.ptr, 2, 0, op, None)
    _lib.call("fd_halo_l2g_end", h, d.ptr, 2, 0, op, None)
    seq = start.copy()
    for g in ghosts:
        seq[shared] = {1: np.add, 2: np.minimum, 3: np.maximum}[op](seq[shared], start[g])
    assert np.array_equal(d.download(np.float64, a.shape), seq)
    # forward: every ghost copy receives its owner's value
    _lib.call("fd_halo_g2l_begin", h, d.ptr, 2, 0, None)
    _lib.call("fd_halo_g2l_end", h, d.ptr, 2, 0, None)
    got = d.download(np.float64, a.shape)
    for g in ghosts:
        assert np.array_equal(got[g], got[shared])
    _lib.call("fd_halo_free", h)


def test_halo_create_rejects_lists_the_kernels_cannot_handle(comm):
    with pytest.raises(_lib.FDHipError, match="two receive lists"):
        _halo_multi(comm, [([1, 2], [10, 11]), ([3, 4], [11, 12])])       # a ghost with two owners
    with pytest.raises(_lib.FDHipError, match="repeats a node"):
        _halo_multi(comm, [([1, 1], [10, 11])])


@pytest.mark.parametrize("op", [1, 2, 3])
def test_corner_of_a_block_partition_rccl_wire_equals_host_wire(comm, op):
    """The interior corner of a 2 x 2 x 2 block partition: SEVEN neighbours (three faces, three edges, one corner), the corner
    node owned here and held as a ghost by all seven, edge nodes by three, face nodes by one.  The reverse exchange (ghost -> owner
    SUM / MIN / MAX, firedrake/halo.py:141-172) runs once over the RCCL wire (the rank is its own seven neighbours) and once over the
    HOST wire -- the path the gloo tests and the one-device rehearsals take: fd_halo_create without a communicator, the packed rows
    carried from the send to the receive buffer by the caller (fd_halo_wire_buffers) -- and the fused node-major combine must give
    the same BITS on both, equal to the contributions applied in neighbour order."""
    m = 12                                              # owned lattice m^3; neighbours on the high side of every axis
    idx = np.arange(m ** 3).reshape(m, m, m)
    hi = m - 1
    # owned nodes each neighbour holds as ghosts: faces x/y/z = hi, edges, the corner
    sends = [idx[hi, :, :].ravel(), idx[:, hi, :].ravel(), idx[:, :, hi].ravel(),
             idx[hi, hi, :].ravel(), idx[hi, :, hi].ravel(), idx[:, hi, hi].ravel(), idx[hi, hi, hi].ravel()]
    n_owned = m ** 3
    # the ghost copies this rank holds of "their" nodes (as the self-neighbour, the same counts): a private ghost range per neighbour
    recvs, off = [], n_owned
    for s_ in sends:
        recvs.append(np.arange(off, off + len(s_)))
        off += len(s_)
    n = off
    rng = np.random.default_rng(11)
    a = rng.standard_normal((n, 2))
    pairs = list(zip(sends, recvs))
    results = {}
    for wire in ("rccl", "host"):
        h = _halo_multi(comm if wire == "rccl" else None, pairs)
        d = DeviceBuffer.from_numpy(a)
        _lib.call("fd_halo_l2g_begin", h, d.ptr, 2, 0, op, None)
        if wire == "host":
            sb, rb, ns, nr = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int64(), ctypes.c_int64()
            _lib.call("fd_halo_wire_buffers", h, d.ptr, 1, ctypes.byref(sb), ctypes.byref(ns), ctypes.byref(rb), ctypes.byref(nr))
            assert ns.value == nr.value == sum(len(s_) for s_ in sends)
            _lib.call("fd_device_sync")
            _lib.call("fd_memcpy_d2d", rb.value, sb.value, ns.value * 2 * 8, None)      # what neighbour k sent is what it receives
            _lib.call("fd_device_sync")
        _lib.call("fd_halo_l2g_end", h, d.ptr, 2, 0, op, None)
        results[wire] = d.download(np.float64, a.shape)
        _lib.call("fd_halo_free", h)
    f = {1: np.add, 2: np.minimum, 3: np.maximum}[op]
    seq = a.copy()
    for s_, r_ in pairs:                                 # contributions in neighbour order
        seq[s_] = f(seq[s_], a[r_])
    assert np.array_equal(results["rccl"], results["host"])
    assert np.array_equal(results["rccl"], seq)
    corner = idx[hi, hi, hi]
    assert not np.array_equal(seq[corner], a[corner])     # (seven contributions landed on the corner)
