"""Host restatement of the halo exchange for the CPU-only multi-rank protocol tests (tests/mp_worker.py).

Test scaffolding, not product code: same pyop2 Halo interface (pyop2/types/halo.py:4-56) as firedrake_amd.halo.Halo, but
the pack / wire / combine are torch index ops on the host arrays of the Dats and ``torch.distributed`` (gloo)
point-to-point messages -- what firedrake/halo.py:125-172 does through a PetscSF.  Installed with
``firedrake_amd.halo.set_halo_factory(HostHalo)`` before the meshes are built; the oracle plays the kernel."""
import numpy as np

from firedrake_amd.op2types import INC, MAX, MIN, WRITE

_OPS = {WRITE: 0, INC: 1, MIN: 2, MAX: 3}


class HostHalo:
    wire = "hostsim"

    def __init__(self, lists):
        self.lists = lists
        self.rank, self.nranks = lists.rank, lists.nranks
        self._idx_cache = {}
        self._pending = {}

    def _neighbours(self):
        return sorted(set(self.lists.send) | set(self.lists.recv))

    def _idx(self, kind, r):
        import torch
        t = self._idx_cache.get((kind, r))
        if t is None:
            arr = (self.lists.send if kind == "send" else self.lists.recv)[r]
            t = torch.from_numpy(np.ascontiguousarray(arr, dtype=np.int64))
            self._idx_cache[(kind, r)] = t
        return t

    def _begin(self, dat, send_kind, recv_kind, tag):
        import torch
        import torch.distributed as dist
        ops, recvs, keep = [], [], []
        hview = dat._to_host()
        t = torch.from_numpy(hview.reshape(hview.shape[0], -1))
        for r in self._neighbours():
            sl = (self.lists.send if send_kind == "send" else self.lists.recv).get(r)
            rl = (self.lists.send if recv_kind == "send" else self.lists.recv).get(r)
            if sl is not None and len(sl):
                sbuf = t[self._idx(send_kind, r)].contiguous()
                keep.append(sbuf)
                ops.append(dist.P2POp(dist.isend, sbuf, r))
            if rl is not None and len(rl):
                rbuf = torch.empty((len(rl), t.shape[1]), dtype=t.dtype)
                recvs.append((r, rbuf))
                ops.append(dist.P2POp(dist.irecv, rbuf, r))
        reqs = dist.batch_isend_irecv(ops) if ops else []
        self._pending[(id(dat), tag)] = (reqs, recvs, keep)

    def _end(self, dat, recv_kind, op, tag):
        import torch
        reqs, recvs, keep = self._pending.pop((id(dat), tag))
        for q in reqs:
            q.wait()
        hview = dat._host_rw()
        t = torch.from_numpy(hview.reshape(hview.shape[0], -1))
        for r, buf in recvs:            # neighbour by neighbour: an owned node may be shared with several of them
            li = self._idx(recv_kind, r)
            if op == 0:
                t[li] = buf
            elif op == 1:
                t[li] += buf
            elif op == 2:
                t[li] = torch.minimum(t[li], buf)
            else:
                t[li] = torch.maximum(t[li], buf)

    def global_to_local_begin(self, dat, insert_mode):
        self._begin(dat, "send", "recv", "g2l")

    def global_to_local_end(self, dat, insert_mode):
        self._end(dat, "recv", 0, "g2l")

    def local_to_global_begin(self, dat, insert_mode):
        self._begin(dat, "recv", "send", "l2g")

    def local_to_global_end(self, dat, insert_mode):
        self._end(dat, "send", _OPS[insert_mode], "l2g")

    def fill_ghosts(self, dat, access_mode):
        n0, n1 = dat.dataset.size, dat.dataset.total_size
        if n1 == n0:
            return
        dt = np.dtype(dat.dtype)
        lim = np.finfo(dt) if dt.kind == "f" else np.iinfo(dt)
        dat._host_rw()[n0:] = {INC: 0, MIN: lim.max, MAX: lim.min}[access_mode]
