// fd_comm.hip -- halo exchange and Global reductions inside the C ABI (include/fdhip.h: fd_comm_*, fd_halo_*).
//
// Counterpart of firedrake/halo.py:87-172: the reference hands every exchange to a PetscSF (bcastBegin/End with
// MPI.REPLACE for owner -> ghost, reduceBegin/End with SUM/MIN/MAX for ghost -> owner); here one fd_halo_t holds the
// per-neighbour index lists on the device, persistent packed buffers, and a side stream on which the wire transfer runs
// as ONE grouped RCCL neighbour exchange (ncclGroupStart; ncclSend/ncclRecv per neighbour; ncclGroupEnd) over xGMI:
//
//   *_begin:  pack kernel on the compute stream -> event -> side stream waits -> grouped send/recv -> event
//   *_end:    compute stream waits for that event -> unpack kernel (=, +=, min, max) on the compute stream
//
// so the host never blocks inside a step and the transfer overlaps whatever the compute stream runs between begin and
// end (the core-entity kernel, pyop2/parloop.py:250-253).  Messages are O(1 MB) per neighbour: latency
// bound, hence one grouped exchange and no ring collective.
//
// RCCL is bound at run time (dlopen of librccl.so, reusing a copy the process already loaded -- PyTorch ships one), so
// libfdhip.so itself has no link-time dependency on it and single-GPU runs never touch it.  A halo created WITHOUT a
// communicator still packs/unpacks into its persistent buffers; the caller then carries the wire itself between
// fd_halo_wire_buffers() and *_end (the gloo/host-bounce path of the CPU-launched multi-rank tests).
#include "fd_common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <cstring>
#include <vector>

namespace {

struct Rccl {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;
};

Rccl *rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return &r;
    tried = true;
    const char *names[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (r.lib) break; }   // a copy already in the process
    if (!r.lib)
        for (const char *n : names) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    if (!r.lib) {
        const char *abs_[] = {"/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : abs_) { r.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (r.lib) break; }
    }
    if (!r.lib) { r.why = std::string("cannot load librccl.so: ") + (dlerror() ? dlerror() : "?"); return &r; }
#define FD_SYM(field, name)                                                         \
    r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, name));              \
    if (!r.field) { r.why = std::string("librccl.so lacks ") + name; r.lib = nullptr; return &r; }
    FD_SYM(GetUniqueId, "ncclGetUniqueId")
    FD_SYM(CommInitRank, "ncclCommInitRank")
    FD_SYM(CommDestroy, "ncclCommDestroy")
    FD_SYM(GroupStart, "ncclGroupStart")
    FD_SYM(GroupEnd, "ncclGroupEnd")
    FD_SYM(Send, "ncclSend")
    FD_SYM(Recv, "ncclRecv")
    FD_SYM(AllReduce, "ncclAllReduce")
    FD_SYM(GetErrorString, "ncclGetErrorString")
#undef FD_SYM
    return &r;
}

#define FD_NCCL(call)                                                                               \
    do {                                                                                            \
        ncclResult_t _r = (call);                                                                   \
        if (_r != ncclSuccess) {                                                                    \
            fd::set_error(std::string(#call) + " failed: " + rccl()->GetErrorString(_r));           \
            return -2;                                                                              \
        }                                                                                           \
    } while (0)

// dtype codes of the C ABI (include/fdhip.h FD_F64 ...): element size, RCCL type
constexpr int NDTYPE = 6;
const size_t ITEMSIZE[NDTYPE] = {8, 4, 4, 4, 8, 8};
const ncclDataType_t NCCL_TYPE[NDTYPE] = {ncclFloat64, ncclFloat32, ncclInt32, ncclUint32, ncclInt64, ncclUint64};

template <class T> __global__ void pack_rows_t(const T *__restrict__ dat, int cdim, const int32_t *__restrict__ idx, int64_t n,
                                               T *__restrict__ buf) {
    const int64_t total = n * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t / cdim;
        buf[t] = dat[(int64_t)idx[k] * cdim + (t - k * cdim)];
    }
}

// One launch covers index lists that never repeat a node: the concatenated RECV lists of a forward exchange (a ghost has
// one owner), or ONE neighbour's SEND list of a reverse exchange.  An owned node shared with several neighbours (a slab
// one cell thick, the edges and corners of a block partition) appears once per neighbour in the concatenated SEND list,
// so exchange_end() launches the reverse combine neighbour by neighbour -- stream order serialises the += / min / max on
// such a node (the old per-neighbour PetscSF reduce of firedrake/halo.py:141-172 has the same semantics).
template <class T> __global__ void unpack_rows_t(T *__restrict__ dat, int cdim, const int32_t *__restrict__ idx, int64_t n,
                                                 const T *__restrict__ buf, int op) {
    const int64_t total = n * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t / cdim;
        T *p = &dat[(int64_t)idx[k] * cdim + (t - k * cdim)];
        const T v = buf[t];
        switch (op) {
            case 0: *p = v; break;
            case 1: *p += v; break;
            case 2: *p = (v < *p) ? v : *p; break;
            default: *p = (v > *p) ? v : *p; break;
        }
    }
}

// Reverse combine when owned nodes ARE shared between neighbours, in ONE launch and without atomics: node-major.  comb_node = the
// distinct owned nodes of the concatenated send lists, comb_pos[comb_ptr[d] .. comb_ptr[d+1]) = the rows of the receive buffer
// that belong to node d in neighbour order -- so a node's contributions are combined in the same order the per-neighbour
// launches applied them (bitwise the same sums), by one lane per (node, component).
template <class T> __global__ void combine_rows_t(T *__restrict__ dat, int cdim, const int32_t *__restrict__ comb_node,
                                                  const int32_t *__restrict__ comb_ptr, const int32_t *__restrict__ comb_pos, int64_t nd,
                                                  const T *__restrict__ buf, int op) {
    const int64_t total = nd * cdim;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t d = t / cdim;
        const int c = (int)(t - d * cdim);
        T *p = &dat[(int64_t)comb_node[d] * cdim + c];
        T acc = *p;
        for (int32_t q = comb_ptr[d]; q < comb_ptr[d + 1]; ++q) {
            const T v = buf[(int64_t)comb_pos[q] * cdim + c];
            switch (op) {
                case 0: acc = v; break;
                case 1: acc += v; break;
                case 2: acc = (v < acc) ? v : acc; break;
                default: acc = (v > acc) ? v : acc; break;
            }
        }
        *p = acc;
    }
}

template <class T> struct Lim;
template <> struct Lim<double> { static constexpr double hi = 1.7976931348623157e308, lo = -1.7976931348623157e308; };
template <> struct Lim<float> { static constexpr float hi = 3.402823466e38f, lo = -3.402823466e38f; };
template <> struct Lim<int32_t> { static constexpr int32_t hi = 2147483647, lo = -2147483647 - 1; };
template <> struct Lim<uint32_t> { static constexpr uint32_t hi = 4294967295u, lo = 0u; };
template <> struct Lim<int64_t> { static constexpr int64_t hi = 9223372036854775807LL, lo = -9223372036854775807LL - 1; };
template <> struct Lim<uint64_t> { static constexpr uint64_t hi = 18446744073709551615ULL, lo = 0ULL; };

template <class T> __global__ void fill_range_t(T *__restrict__ p, int64_t n, int kind) {
    const T v = kind == 0 ? (T)0 : (kind == 1 ? Lim<T>::hi : Lim<T>::lo);
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) p[t] = v;
}

inline int grid_for(int64_t n) { int64_t g = (n + 255) / 256; if (g < 1) g = 1; if (g > 8192) g = 8192; return (int)g; }

template <class F> int by_dtype(int dtype, F &&f) {
    switch (dtype) {
        case 0: return f((double *)nullptr);
        case 1: return f((float *)nullptr);
        case 2: return f((int32_t *)nullptr);
        case 3: return f((uint32_t *)nullptr);
        case 4: return f((int64_t *)nullptr);
        case 5: return f((uint64_t *)nullptr);
    }
    fd::set_error("dtype code must be 0..5 (f64, f32, i32, u32, i64, u64)");
    return -1;
}

}  // namespace

struct fd_comm_s {
    ncclComm_t comm = nullptr;
    int rank = 0, nranks = 1;
    hipStream_t stream = nullptr;       // the side stream all transfers of this communicator run on
};

struct fd_halo_slot {
    const void *key = nullptr;          // Dat the slot is busy for (nullptr = free)
    int dir = 0;                        // 0 = forward (owner -> ghost), 1 = reverse (ghost -> owner)
    void *sbuf = nullptr, *rbuf = nullptr;
    size_t scap = 0, rcap = 0;
    hipEvent_t packed = nullptr, done = nullptr;
};

struct fd_halo_s {
    fd_comm_t comm = nullptr;
    std::vector<int> peer;
    std::vector<int64_t> nsend, nrecv, soff, roff;     // rows per neighbour and their offsets in the concatenated lists
    int32_t *send_idx = nullptr, *recv_idx = nullptr;  // concatenated per-neighbour lists on the device
    int64_t tsend = 0, trecv = 0;
    bool send_disjoint = true;                         // no owned node is sent to two neighbours: one reverse-combine launch
    int32_t *comb_node = nullptr, *comb_ptr = nullptr, *comb_pos = nullptr;   // node-major combine tables (shared nodes only)
    int64_t ncomb = 0;
    std::vector<fd_halo_slot> slots;
};

namespace {

int slot_for(fd_halo_t h, const void *dat, int dir, bool acquire, fd_halo_slot **out) {
    for (auto &s : h->slots)
        if (s.key == dat && s.dir == dir) {
            if (acquire) FD_FAIL("halo exchange of this Dat in this direction is already in flight");
            *out = &s;
            return 0;
        }
    if (!acquire) FD_FAIL("fd_halo_*_end without a matching *_begin for this Dat");
    for (auto &s : h->slots)
        if (s.key == nullptr) { s.key = dat; s.dir = dir; *out = &s; return 0; }
    h->slots.emplace_back();
    fd_halo_slot &s = h->slots.back();
    FD_HIP(hipEventCreateWithFlags(&s.packed, hipEventDisableTiming));
    FD_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
    s.key = dat; s.dir = dir;
    *out = &s;
    return 0;
}

int grow(void **buf, size_t *cap, size_t need) {
    if (need <= *cap) return 0;
    if (*buf) FD_HIP(hipFree(*buf));
    *buf = nullptr; *cap = 0;
    FD_HIP(hipMalloc(buf, need));
    *cap = need;
    return 0;
}

// dir 0: pack the SEND lists, receive into buffers laid out like the RECV lists; dir 1: the other way round
int exchange_begin(fd_halo_t h, void *dat, int cdim, int dtype, int dir, fd_stream_t s_) {
    if (!h) FD_FAIL("null halo");
    if (dtype < 0 || dtype >= NDTYPE) FD_FAIL("dtype code must be 0..5");
    if (cdim <= 0) FD_FAIL("cdim must be positive");
    hipStream_t s = fd::st(s_);
    fd_halo_slot *sl = nullptr;
    if (int rc = slot_for(h, dat, dir, true, &sl)) return rc;
    const size_t row = (size_t)cdim * ITEMSIZE[dtype];
    const int64_t nout = dir == 0 ? h->tsend : h->trecv, nin = dir == 0 ? h->trecv : h->tsend;
    const int32_t *oidx = dir == 0 ? h->send_idx : h->recv_idx;
    if (int rc = grow(&sl->sbuf, &sl->scap, (size_t)nout * row)) { sl->key = nullptr; return rc; }
    if (int rc = grow(&sl->rbuf, &sl->rcap, (size_t)nin * row)) { sl->key = nullptr; return rc; }
    if (nout > 0) {
        int rc = by_dtype(dtype, [&](auto *tag) {
            using T = std::remove_pointer_t<decltype(tag)>;
            hipLaunchKernelGGL(pack_rows_t<T>, dim3(grid_for(nout * cdim)), dim3(256), 0, s, (const T *)dat, cdim, oidx, nout, (T *)sl->sbuf);
            return 0;
        });
        if (rc) { sl->key = nullptr; return rc; }
        FD_CHECK_LAUNCH();
    }
    if (!h->comm) return 0;             // external wire: the caller moves sbuf -> peer rbuf (fd_halo_wire_buffers)
    Rccl *R = rccl();
    hipStream_t cs = h->comm->stream;
    FD_HIP(hipEventRecord(sl->packed, s));
    FD_HIP(hipStreamWaitEvent(cs, sl->packed, 0));
    const auto &no = dir == 0 ? h->nsend : h->nrecv, &ni = dir == 0 ? h->nrecv : h->nsend;
    const auto &oo = dir == 0 ? h->soff : h->roff, &io = dir == 0 ? h->roff : h->soff;
    // a failure inside the group must still close it and release the slot, or every later exchange of this Dat would
    // find the NCCL group open and the slot "in flight"
    ncclResult_t bad = R->GroupStart();
    const bool opened = bad == ncclSuccess;
    for (size_t k = 0; opened && bad == ncclSuccess && k < h->peer.size(); ++k) {
        if (no[k] > 0)
            bad = R->Send((const char *)sl->sbuf + (size_t)oo[k] * row, (size_t)no[k] * cdim, NCCL_TYPE[dtype], h->peer[k], h->comm->comm, cs);
        if (bad == ncclSuccess && ni[k] > 0)
            bad = R->Recv((char *)sl->rbuf + (size_t)io[k] * row, (size_t)ni[k] * cdim, NCCL_TYPE[dtype], h->peer[k], h->comm->comm, cs);
    }
    if (opened) { const ncclResult_t e = R->GroupEnd(); if (bad == ncclSuccess) bad = e; }
    if (bad != ncclSuccess) {
        sl->key = nullptr;
        fd::set_error(std::string("RCCL neighbour exchange failed: ") + R->GetErrorString(bad));
        return -2;
    }
    hipError_t he = hipEventRecord(sl->done, cs);
    if (he != hipSuccess) { sl->key = nullptr; FD_HIP(he); }
    return 0;
}

int exchange_end(fd_halo_t h, void *dat, int cdim, int dtype, int dir, int op, fd_stream_t s_) {
    if (!h) FD_FAIL("null halo");
    if (dtype < 0 || dtype >= NDTYPE) FD_FAIL("dtype code must be 0..5");
    if (op < 0 || op > 3) FD_FAIL("op must be 0 (replace), 1 (sum), 2 (min) or 3 (max)");
    hipStream_t s = fd::st(s_);
    fd_halo_slot *sl = nullptr;
    if (int rc = slot_for(h, dat, dir, false, &sl)) return rc;
    if (h->comm) FD_HIP(hipStreamWaitEvent(s, sl->done, 0));
    const int64_t nin = dir == 0 ? h->trecv : h->tsend;
    const int32_t *iidx = dir == 0 ? h->recv_idx : h->send_idx;
    int rc = 0;
    if (nin > 0) {
        const size_t row = (size_t)cdim * ITEMSIZE[dtype];
        rc = by_dtype(dtype, [&](auto *tag) {
            using T = std::remove_pointer_t<decltype(tag)>;
            if (dir == 0 || h->peer.size() <= 1 || h->send_disjoint) {
                hipLaunchKernelGGL(unpack_rows_t<T>, dim3(grid_for(nin * cdim)), dim3(256), 0, s, (T *)dat, cdim, iidx, nin, (const T *)sl->rbuf, op);
                return 0;
            }
            if (h->comb_node) {                                 // shared owned nodes: node-major combine, one launch
                hipLaunchKernelGGL(combine_rows_t<T>, dim3(grid_for(h->ncomb * cdim)), dim3(256), 0, s, (T *)dat, cdim, h->comb_node, h->comb_ptr,
                                   h->comb_pos, h->ncomb, (const T *)sl->rbuf, op);
                return 0;
            }
            for (size_t k = 0; k < h->peer.size(); ++k) {       // (no combine tables: one neighbour after the other)
                const int64_t n = h->nsend[k];
                if (n > 0)
                    hipLaunchKernelGGL(unpack_rows_t<T>, dim3(grid_for(n * cdim)), dim3(256), 0, s, (T *)dat, cdim, iidx + h->soff[k], n,
                                       (const T *)((const char *)sl->rbuf + (size_t)h->soff[k] * row), op);
            }
            return 0;
        });
        if (!rc) { hipError_t e = hipGetLastError(); if (e != hipSuccess) { fd::set_error(hipGetErrorString(e)); rc = (int)e; } }
    }
    sl->key = nullptr;                  // the buffers stay allocated for the next step; stream order protects their reuse
    return rc;
}

}  // namespace

extern "C" {

int fd_comm_available(void) {
    Rccl *R = rccl();
    if (!R->lib) { fd::set_error(R->why); return 0; }
    return 1;
}

int fd_comm_unique_id(unsigned char *id128) {
    Rccl *R = rccl();
    if (!R->lib) FD_FAIL(R->why);
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    FD_NCCL(R->GetUniqueId(&id));
    std::memcpy(id128, &id, sizeof(id));
    return 0;
}

int fd_comm_create(const unsigned char *id128, int rank, int nranks, fd_comm_t *out) {
    Rccl *R = rccl();
    if (!R->lib) FD_FAIL(R->why);
    if (!id128 || rank < 0 || rank >= nranks) FD_FAIL("fd_comm_create: bad arguments");
    ncclUniqueId id;
    std::memcpy(&id, id128, sizeof(id));
    auto *c = new fd_comm_s;
    c->rank = rank; c->nranks = nranks;
    ncclResult_t r = R->CommInitRank(&c->comm, nranks, id, rank);
    if (r != ncclSuccess) {
        // the first multi-device run must not be wasted on a bare error code: say who, where and what to look at
        int dev = -1, ndev = 0;
        (void)hipGetDevice(&dev); (void)hipGetDeviceCount(&ndev);
        const char *ipc = getenv("HSA_ENABLE_IPC_MODE_LEGACY");
        fd::set_error(std::string("ncclCommInitRank failed on rank ") + std::to_string(rank) + " of " + std::to_string(nranks) + " (HIP device " +
                      std::to_string(dev) + " of " + std::to_string(ndev) + " visible): " + R->GetErrorString(r) +
                      "; HSA_ENABLE_IPC_MODE_LEGACY=" + (ipc ? ipc : "<unset: this driver needs 0>") +
                      "; every rank must call fd_comm_create with the same 128-byte id and its own device selected (fd_set_device) BEFORE the call"
                      "; NCCL_DEBUG=INFO prints RCCL's own account");
        delete c;
        return -2;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { R->CommDestroy(c->comm); delete c; FD_HIP(e); }
    *out = c;
    return 0;
}

int fd_comm_free(fd_comm_t c) {
    if (!c) return 0;
    if (c->stream) { (void)hipStreamSynchronize(c->stream); (void)hipStreamDestroy(c->stream); }
    if (c->comm) rccl()->CommDestroy(c->comm);
    delete c;
    return 0;
}

int fd_comm_info(fd_comm_t c, int *rank, int *nranks) {
    if (!c) FD_FAIL("null communicator");
    if (rank) *rank = c->rank;
    if (nranks) *nranks = c->nranks;
    return 0;
}

// In-place all-reduce of `count` elements on the device: the MPI_Iallreduce of INC/MIN/MAX Globals
// (pyop2/parloop.py:411-442).  Ordered after the work queued on `s`, and `s` waits for the result.
int fd_comm_allreduce(fd_comm_t c, void *buf, int64_t count, int dtype, int op, fd_stream_t s_) {
    if (!c) FD_FAIL("null communicator");
    if (dtype < 0 || dtype >= NDTYPE) FD_FAIL("dtype code must be 0..5");
    if (op < 1 || op > 3) FD_FAIL("op must be 1 (sum), 2 (min) or 3 (max)");
    if (count <= 0) return 0;
    Rccl *R = rccl();
    hipStream_t s = fd::st(s_);
    hipEvent_t ev;
    FD_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    FD_HIP(hipEventRecord(ev, s));
    FD_HIP(hipStreamWaitEvent(c->stream, ev, 0));
    const ncclRedOp_t rop = op == 1 ? ncclSum : (op == 2 ? ncclMin : ncclMax);
    ncclResult_t r = R->AllReduce(buf, buf, (size_t)count, NCCL_TYPE[dtype], rop, c->comm, c->stream);
    if (r != ncclSuccess) { (void)hipEventDestroy(ev); fd::set_error(std::string("ncclAllReduce failed: ") + R->GetErrorString(r)); return -2; }
    FD_HIP(hipEventRecord(ev, c->stream));
    FD_HIP(hipStreamWaitEvent(s, ev, 0));
    FD_HIP(hipEventDestroy(ev));
    return 0;
}

int fd_halo_create(fd_comm_t comm, int nneigh, const int32_t *peers, const int32_t *const *send_idx_host, const int32_t *nsend,
                   const int32_t *const *recv_idx_host, const int32_t *nrecv, fd_halo_t *out) {
    if (nneigh < 0 || !out) FD_FAIL("fd_halo_create: bad arguments");
    auto *h = new fd_halo_s;
    h->comm = comm;
    std::vector<int32_t> sall, rall;
    for (int k = 0; k < nneigh; ++k) {
        // a rank may be its own neighbour (a periodic direction that is not partitioned): ncclSend/ncclRecv to self inside the group
        if (comm && (peers[k] < 0 || peers[k] >= comm->nranks)) { delete h; FD_FAIL("fd_halo_create: bad neighbour rank"); }
        h->peer.push_back(peers[k]);
        h->nsend.push_back(nsend[k]); h->nrecv.push_back(nrecv[k]);
        h->soff.push_back((int64_t)sall.size()); h->roff.push_back((int64_t)rall.size());
        sall.insert(sall.end(), send_idx_host[k], send_idx_host[k] + nsend[k]);
        rall.insert(rall.end(), recv_idx_host[k], recv_idx_host[k] + nrecv[k]);
    }
    h->tsend = (int64_t)sall.size(); h->trecv = (int64_t)rall.size();
    {   // validate what the kernels rely on: a list never repeats a node; ghosts have one owner; shared owned nodes are noted
        std::vector<int32_t> t(rall);
        std::sort(t.begin(), t.end());
        if (std::adjacent_find(t.begin(), t.end()) != t.end()) { delete h; FD_FAIL("fd_halo_create: a ghost node appears in two receive lists"); }
        for (int k = 0; k < nneigh; ++k) {
            t.assign(send_idx_host[k], send_idx_host[k] + nsend[k]);
            std::sort(t.begin(), t.end());
            if (std::adjacent_find(t.begin(), t.end()) != t.end()) { delete h; FD_FAIL("fd_halo_create: a send list repeats a node"); }
        }
        t = sall;
        std::sort(t.begin(), t.end());
        h->send_disjoint = std::adjacent_find(t.begin(), t.end()) == t.end();
    }
    std::vector<int32_t> cnode, cptr, cpos;
    if (!h->send_disjoint) {
        // node-major tables of the reverse combine: positions of the concatenated send list grouped by node, neighbour order kept
        std::vector<int32_t> ord(sall.size());
        for (size_t i = 0; i < ord.size(); ++i) ord[i] = (int32_t)i;
        std::stable_sort(ord.begin(), ord.end(), [&](int32_t a, int32_t b) { return sall[a] < sall[b]; });
        for (size_t i = 0; i < ord.size(); ++i) {
            if (i == 0 || sall[ord[i]] != sall[ord[i - 1]]) { cnode.push_back(sall[ord[i]]); cptr.push_back((int32_t)i); }
            cpos.push_back(ord[i]);
        }
        cptr.push_back((int32_t)ord.size());
        h->ncomb = (int64_t)cnode.size();
    }
    hipError_t e = hipMalloc((void **)&h->send_idx, std::max<size_t>(sall.size() * 4, 8));
    if (e == hipSuccess) e = hipMalloc((void **)&h->recv_idx, std::max<size_t>(rall.size() * 4, 8));
    if (e == hipSuccess && !cnode.empty()) {
        e = hipMalloc((void **)&h->comb_node, cnode.size() * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&h->comb_ptr, cptr.size() * 4);
        if (e == hipSuccess) e = hipMalloc((void **)&h->comb_pos, cpos.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(h->comb_node, cnode.data(), cnode.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->comb_ptr, cptr.data(), cptr.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(h->comb_pos, cpos.data(), cpos.size() * 4, hipMemcpyHostToDevice);
    }
    if (e == hipSuccess && !sall.empty()) e = hipMemcpy(h->send_idx, sall.data(), sall.size() * 4, hipMemcpyHostToDevice);
    if (e == hipSuccess && !rall.empty()) e = hipMemcpy(h->recv_idx, rall.data(), rall.size() * 4, hipMemcpyHostToDevice);
    if (e != hipSuccess) { (void)hipFree(h->send_idx); (void)hipFree(h->recv_idx); (void)hipFree(h->comb_node); (void)hipFree(h->comb_ptr);
                           (void)hipFree(h->comb_pos); delete h; FD_HIP(e); }
    *out = h;
    return 0;
}

int fd_halo_free(fd_halo_t h) {
    if (!h) return 0;
    for (auto &s : h->slots) {
        if (s.sbuf) (void)fd::release(s.sbuf);
        if (s.rbuf) (void)fd::release(s.rbuf);
        if (s.packed) (void)hipEventDestroy(s.packed);
        if (s.done) (void)hipEventDestroy(s.done);
    }
    (void)fd::release(h->send_idx); (void)fd::release(h->recv_idx);
    (void)fd::release(h->comb_node); (void)fd::release(h->comb_ptr); (void)fd::release(h->comb_pos);
    delete h;
    return 0;
}

int fd_halo_g2l_begin(fd_halo_t h, void *dat, int cdim, int dtype, fd_stream_t s) { return exchange_begin(h, dat, cdim, dtype, 0, s); }
int fd_halo_g2l_end(fd_halo_t h, void *dat, int cdim, int dtype, fd_stream_t s) { return exchange_end(h, dat, cdim, dtype, 0, 0, s); }
int fd_halo_l2g_begin(fd_halo_t h, void *dat, int cdim, int dtype, int op, fd_stream_t s) {
    if (op < 1 || op > 3) FD_FAIL("fd_halo_l2g_begin: op must be 1 (sum), 2 (min) or 3 (max)");
    return exchange_begin(h, dat, cdim, dtype, 1, s);
}
int fd_halo_l2g_end(fd_halo_t h, void *dat, int cdim, int dtype, int op, fd_stream_t s) {
    if (op < 1 || op > 3) FD_FAIL("fd_halo_l2g_end: op must be 1 (sum), 2 (min) or 3 (max)");
    return exchange_end(h, dat, cdim, dtype, 1, op, s);
}

// The packed buffers of the exchange in flight for `dat` (external wire: a halo created without a communicator).
// Rows of neighbour k start at row offset *_off[k] of the list the buffer mirrors.
int fd_halo_wire_buffers(fd_halo_t h, const void *dat, int dir, void **send_buf, int64_t *send_rows, void **recv_buf, int64_t *recv_rows) {
    if (!h) FD_FAIL("null halo");
    fd_halo_slot *sl = nullptr;
    if (int rc = slot_for(h, dat, dir, false, &sl)) return rc;
    if (send_buf) *send_buf = sl->sbuf;
    if (recv_buf) *recv_buf = sl->rbuf;
    if (send_rows) *send_rows = dir == 0 ? h->tsend : h->trecv;
    if (recv_rows) *recv_rows = dir == 0 ? h->trecv : h->tsend;
    return 0;
}

// Fill `count` elements starting at element `first` with the identity of an access mode: kind 0 = zero (INC),
// 1 = largest value (MIN), 2 = lowest value (MAX) of the dtype -- the ghost fill of pyop2/types/dat.py:631-636.
int fd_dat_fill_range(void *dat, int64_t first, int64_t count, int dtype, int kind, fd_stream_t s_) {
    if (count <= 0) return 0;
    if (kind < 0 || kind > 2) FD_FAIL("fd_dat_fill_range: kind must be 0, 1 or 2");
    hipStream_t s = fd::st(s_);
    int rc = by_dtype(dtype, [&](auto *tag) {
        using T = std::remove_pointer_t<decltype(tag)>;
        hipLaunchKernelGGL(fill_range_t<T>, dim3(grid_for(count)), dim3(256), 0, s, (T *)dat + first, count, kind);
        return 0;
    });
    if (rc) return rc;
    FD_CHECK_LAUNCH();
    return 0;
}

}  // extern "C"
