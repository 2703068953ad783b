"""TEST INFRASTRUCTURE ONLY: run the wrappers emitted by firedrake_amd/codegen.py on the host.

``run_direct``: the generated HIP source of a DIRECT-mode wrapper is compiled by g++ against
tests/hostsim/fd_wrapper.h (a sequential one-lane stand-in for csrc/fd_wrapper.h) and called with host pointers in
the kernel's own parameter order (``WrapperSource.layout``): maps, extruded offsets, layer bounds, subsets, lgmap
masking, CSR search.  ``run_staged`` / ``run_ocr``: STAGED and OWNER-COMPUTES-ROWS wrappers compiled against
tests/hostsim/mt/fd_wrapper.h and executed with one OS thread per lane (real barriers, a shared LDS buffer, CAS
atomics), on plan tables built by the numpy restatements in helpers.py.  Everything is compared with the oracle by the
callers.  This is not a fallback: nothing under firedrake_amd/ can reach it, and the product path still raises
without a GPU.
"""
import ctypes
import hashlib
import os
import subprocess

import numpy as np

from helpers import oracle_pattern
from firedrake_amd.codegen import generate_wrapper
from firedrake_amd.configuration import configuration
from firedrake_amd.parloop import DatParloopArg, GlobalParloopArg, MatParloopArg

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "hostsim", "_build")


def _compile(source, name):
    os.makedirs(_BUILD, exist_ok=True)
    key = hashlib.sha1(source.encode()).hexdigest()[:16]
    so = os.path.join(_BUILD, f"{name}_{key}.so")
    if not os.path.exists(so):
        src = os.path.join(_BUILD, f"{name}_{key}.{os.getpid()}.cpp")
        with open(src, "w") as f:
            f.write(source)
        cmd = ["g++", "-O1", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-std=gnu++17", "-w", "-I", os.path.join(_HERE, "hostsim"),
               "-o", so + f".{os.getpid()}.tmp", src, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hostsim compile failed:\n" + r.stderr[-4000:])
        os.replace(so + f".{os.getpid()}.tmp", so)
    return ctypes.CDLL(so)


def _compile_mt(source, name):
    os.makedirs(_BUILD, exist_ok=True)
    key = hashlib.sha1(source.encode()).hexdigest()[:16]
    so = os.path.join(_BUILD, f"{name}_mt_{key}.so")
    if not os.path.exists(so):
        src = os.path.join(_BUILD, f"{name}_mt_{key}.{os.getpid()}.cpp")
        with open(src, "w") as f:
            f.write(source)
        # -Bsymbolic: the wrapper's own definition wins over a same-named kernel handle exported by a loaded libfdhip.so
        cmd = ["g++", "-O1", "-fPIC", "-shared", "-Wl,-Bsymbolic", "-std=gnu++20", "-pthread", "-w", "-I", os.path.join(_HERE, "hostsim", "mt"),
               "-o", so + f".{os.getpid()}.tmp", src, "-lm"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hostsim (mt) compile failed:\n" + r.stderr[-4000:])
        os.replace(so + f".{os.getpid()}.tmp", so)
    return ctypes.CDLL(so)


# plan-ordered copies of READ Dats (Parloop._plan_copy): True -> the host-sim hands the wrapper the rows gathered over the plan's
# node list (the streaming branch of the staging phase), False -> a null pointer (the gather branch).  Tests flip it.
PLAN_COPIES = False


def _plan_copy_arg(pl, desc, plans, ptr):
    if not PLAN_COPIES:
        return ctypes.c_void_p(0)
    d = pl.arguments[desc[1]].data
    d = getattr(d, "_parent", d)
    host = d._host if d._host_valid else d._to_host()
    rows = np.asarray(host).reshape(host.shape[0], -1)
    return ptr(np.ascontiguousarray(rows[np.asarray(plans[desc[2]][1], dtype=np.int64)]))



def run_staged(pl, epb=48, order=None, direct_noreuse=False):
    """Execute Parloop ``pl`` (Dat / Global arguments only) with the STAGED wrapper on the host: one OS thread per
    lane, workgroups one after the other, real barriers and atomics (tests/hostsim/mt/fd_wrapper.h).  The
    block-localisation plans come from the numpy restatement in helpers.py (itself checked against the device's
    plans by the -m gpu suite), stored in lane order like the device's.  Returns COPIES like run_direct."""
    import re
    from firedrake_amd.codegen import mode_variant, staged_eligible
    from helpers import lane_slot_to_entity, plan_ref
    gk = pl.global_kernel
    assert staged_eligible(gk) and not any(isinstance(pa, MatParloopArg) for pa in pl.arguments)
    # subsets / extruded sets: the plans live on derived maps over the virtual (position x layer) space
    virt = pl._virtual(staged=True)
    start, end = 0, (virt.size(pl.iterset.size) if virt else pl.iterset.size)
    maps, base_maps = [], []
    for pa in pl.arguments:
        for m in getattr(pa, "maps", ()):
            if all(m._base() is not q for q in base_maps):
                base_maps.append(m._base())
                maps.append(pl._plan_map(m._base(), staged=True))
    base = generate_wrapper(gk, "stagedo" if order is not None else "staged")
    T = base.block_threads
    plans = {}
    for mi in base.staged_maps:
        rows = np.asarray(maps[mi].values_with_halo)
        if order is not None:                                   # "stagedo": plans over the rows gathered in the entity order
            rows = rows[np.asarray(order)]
        blk, lst, lm = plan_ref(rows, start, end, epb)
        if base.lane_threads:                                   # lane order: slot k*T + t <- k-th entity of lane t's run
            for b0 in range(start, end, epb):
                b1 = min(end, b0 + epb)
                lm[b0 - start:b1 - start] = lm[b0 - start:b1 - start][lane_slot_to_entity(b1 - b0, T)]
        plans[mi] = (blk, lst, np.ascontiguousarray(lm), int(np.diff(blk).max()) if len(blk) > 1 else 0)
    bstart = np.array(list(range(start, end, epb)) + [end], dtype=np.int32)
    nblocks = len(bstart) - 1
    # direct_noreuse: maps whose node lists show no reuse inside any block go straight from / to global memory ("_d" variants), the
    # choice Parloop._staged_geometry makes from the device plans
    direct = []
    if direct_noreuse:
        def affine(mi):                                     # map[e][i] == arity*e + i (what "_d" variants rely on)
            v = np.asarray(maps[mi].values_with_halo)[start:end]
            return np.array_equal(v, np.arange(start, end)[:, None] * v.shape[1] + np.arange(v.shape[1])[None, :])
        direct = [mi for mi in base.staged_maps if len(plans[mi][1]) == (end - start) * maps[mi].arity and order is None and affine(mi)]
        direct = direct[:max(len(base.staged_maps) - 1, 0)]
    kept = [mi for mi in base.staged_maps if mi not in direct]
    src = generate_wrapper(gk, mode_variant("stagedo" if order is not None else "staged", 1, [plans[mi][3] for mi in kept], direct_maps=direct))
    assert src.staged_maps == kept and (not direct or "_d" in src.mode)
    text = src.source.replace("extern __shared__ __align__(16) unsigned char fd_lds[];", "unsigned char *fd_lds = fd_sim::lds;")
    # driver: the kernel's own parameter list, run as nblocks workgroups of T lanes
    sig = re.search(r'extern "C" __global__[^\n]*void %s\((.*)\)\n' % src.symbol, text).group(1)
    names = [p.split()[-1].lstrip("*") for p in sig.split(", ")]
    text += ('\nextern "C" void sim_run(int fd_nblocks, int fd_nthreads, %s)\n{\n  fd_sim::run(fd_nblocks, fd_nthreads, [&] { %s(%s); });\n}\n'
             % (sig, src.symbol, ", ".join(names)))
    lib = _compile_mt(text, src.symbol)
    lib.sim_run.restype = None
    outs, keep = {}, []

    def ptr(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return ctypes.c_void_p(a.ctypes.data)

    cargs = [ctypes.c_int(nblocks), ctypes.c_int(T), ctypes.c_int(start), ctypes.c_int(end)]
    for desc in src.layout:
        kind = desc[0]
        if kind in ("virt_col", "virt_layer"):
            cargs.append(ptr(np.ascontiguousarray(virt.col if kind == "virt_col" else virt.layer, dtype=np.int32)))
        elif kind == "layers":
            cargs.append(ptr(np.asarray(pl.iterset.layers_array, dtype=np.int32)))
        elif kind == "subset":
            cargs.append(ptr(np.asarray(pl.iterset.indices, dtype=np.int32)))
        elif kind == "arg":
            pa = pl.arguments[desc[1]]
            host = pa.data._host if pa.data._host_valid else pa.data._to_host()
            a = np.array(host, copy=True)
            outs[desc[1]] = a
            cargs.append(ctypes.c_void_p(a.ctypes.data))
        elif kind == "map":
            cargs.append(ptr(np.asarray(base_maps[desc[1]].values_with_halo, dtype=np.int32)))
        elif kind == "bstart":
            cargs.append(ptr(bstart))
        elif kind == "order":
            cargs.append(ptr(np.asarray(order, dtype=np.int32)))
        elif kind.startswith("plan_") and kind != "plan_copy" and desc[1] in direct:
            raise AssertionError("a direct map has no plan parameters")
        elif kind == "plan_blkoff":
            cargs.append(ptr(plans[desc[1]][0]))
        elif kind == "plan_list":
            cargs.append(ptr(plans[desc[1]][1]))
        elif kind == "plan_lmap":
            cargs.append(ptr(plans[desc[1]][2]))
        elif kind == "plan_maxnd":
            cargs.append(ctypes.c_longlong(plans[desc[1]][3]))
        elif kind == "plan_copy":
            cargs.append(_plan_copy_arg(pl, desc, plans, ptr))
        else:
            raise AssertionError(f"hostsim (staged) cannot provide {kind}")
    lib.sim_run(*cargs)
    return [outs.get(k) for k in range(len(pl.arguments))]


def pack_records_ref(lmaps, lbits, kidx, nr, nc, kbits, diag, words, extra=None, ebits=0, sentinel=False):
    """numpy restatement of fd_ocr_pack_records: per instance the local-map rows of the staged maps (lbits[m] bits per entry)
    and the nr x nc row offsets (kbits bits, diagonal left out when ``diag``), back to back from bit 0 of ``words`` 32-bit words."""
    kidx = np.asarray(kidx).reshape(-1, nr * nc)
    ninst = len(kidx)
    out = np.zeros((max(ninst, 1), words), dtype=np.uint32)
    for t in range(ninst):
        acc, off = 0, 0
        for lm, lb in zip(lmaps, lbits):
            for v in np.asarray(lm).reshape(ninst, -1)[t]:
                assert int(v) < (1 << lb)
                acc |= int(v) << off
                off += lb
        for i in range(nr):
            for j in range(nc):
                if diag and i == j:
                    continue
                v = int(kidx[t, i * nc + j])
                if sentinel:                 # "dropped" (all ones of the source type) -> all ones of the field
                    v = (1 << kbits) - 1 if v == (1 << (8 * kidx.dtype.itemsize)) - 1 else v
                    assert v == (1 << kbits) - 1 or v < (1 << kbits) - 1
                assert v < (1 << kbits)
                acc |= v << off
                off += kbits
        for v in ([] if extra is None else np.atleast_1d(np.asarray(extra)[t])):      # (one slot per row of a paired instance)
            v = int(v)
            if sentinel:
                v = (1 << ebits) - 1 if v == 0xffff else v
            assert v < (1 << ebits)
            acc |= v << off
            off += ebits
        for w in range(words):
            out[t, w] = (acc >> (32 * w)) & 0xffffffff
    return out


NNZ = np.int64            # include/fdhip.h: fd_nnz_t (row starts of the value array and of the block accumulators)


def row_entry_positions_masked_ref(gpos, colidx, clg):
    """numpy restatement of fd_row_entry_positions_masked: places of entries in masked columns (clg[col] < 0) read -2 - place."""
    gpos = np.asarray(gpos, dtype=np.int32).copy()
    live = gpos >= 0
    masked = np.zeros(len(gpos), dtype=bool)
    masked[live] = np.asarray(clg)[np.asarray(colidx)[gpos[live]]] < 0
    gpos[masked] = -2 - gpos[masked]
    return gpos


def row_runs_ref(prowptr, gstart, rb):
    """numpy restatement of fd_ocr_row_runs: (grun per entry, brun per block, rdelta per run, most runs in a block)."""
    npos = len(gstart)
    disp = np.asarray(gstart, dtype=np.int64) - np.asarray(prowptr[:npos], dtype=np.int64)
    flag = np.ones(npos, dtype=np.int64)
    flag[1:] = disp[1:] != disp[:-1]
    flag[[int(r) for r in rb[:-1] if r < npos]] = 1
    runidx = np.cumsum(flag) - 1
    nruns = int(runidx[-1]) + 1 if npos else 0
    brun = np.array([int(runidx[r]) if r < npos else nruns for r in rb], dtype=np.int32)
    rdelta = np.zeros(max(nruns, 1), dtype=NNZ)
    rdelta[runidx[flag == 1]] = disp[flag == 1]
    grun = np.zeros(max(int(prowptr[npos]), 1), dtype=np.uint8)
    for b in range(len(rb) - 1):
        for p_ in range(int(rb[b]), int(rb[b + 1])):
            assert runidx[p_] - brun[b] <= 255
            grun[int(prowptr[p_]):int(prowptr[p_ + 1])] = runidx[p_] - brun[b]
    return grun, brun, rdelta, int(np.diff(brun).max()) if len(rb) > 1 else 0


def run_ocr(pl, rows_per_block=24, zero_pending=True, order=None, records=False, fixed_point=None, flush_colmask=False):
    """Execute a matrix-assembly Parloop ``pl`` with the OWNER-COMPUTES-ROWS wrapper on the host (one OS thread per
    lane, tests/hostsim/mt/fd_wrapper.h).  The plan tables come from the numpy restatements in helpers.py; the CSR
    pattern from the oracle.  Returns the OracleCSR holding the assembled values."""
    import re
    from firedrake_amd.codegen import mode_variant, ocr_eligible
    from helpers import first_touch_ref, ocr_plan_ref, plan_ref_blocks
    gk = pl.global_kernel
    assert ocr_eligible(gk)
    (k, mpa), = [(k, pa) for k, pa in enumerate(pl.arguments) if isinstance(pa, MatParloopArg)]
    # subsets / extruded sets: the plan lives on the derived maps over the virtual (position x layer) space
    virt = pl._virtual(staged=True)
    base_maps, maps = [], []
    for pa in pl.arguments:
        for m in getattr(pa, "maps", ()):
            if all(m._base() is not q for q in base_maps):
                base_maps.append(m._base())
                maps.append(pl._plan_map(m._base(), staged=True))
    base = generate_wrapper(gk, "ocr")
    T = base.block_threads
    csr = oracle_pattern(mpa.data.sparsity)
    rmap, cmap = (pl._plan_map(m._base(), staged=True) for m in mpa.maps)
    nent = virt.size(pl.iterset.size) if virt else pl.iterset.size
    nrows = rmap.toset.size
    rb = np.array(list(range(0, nrows, rows_per_block)) + [nrows], dtype=np.int32)
    plist = pinv = prowptr = None
    if order is not None:
        # "ocrp": row blocks are ranges of row positions under the first-touch order of the entity order
        plist, pinv = first_touch_ref(np.asarray(rmap.values_with_halo), order, nrows)
        plen = np.diff(csr.rowptr)[:nrows][plist]
        prowptr = np.concatenate([[0], np.cumsum(plen)]).astype(np.int32)
    inst_off, inst_ent, kidx = ocr_plan_ref(np.asarray(rmap.values_with_halo), np.asarray(cmap.values_with_halo), nent, rb,
                                            csr.rowptr, csr.colidx, pinv=pinv)
    plans = {}
    for mi in base.staged_maps:
        blk, lst, lm = plan_ref_blocks(np.asarray(maps[mi].values_with_halo)[inst_ent], inst_off)
        plans[mi] = (blk, lst, np.ascontiguousarray(lm), int(np.diff(blk).max()) if len(blk) > 1 else 0)
    max_nnz = int(np.diff((prowptr if order is not None else csr.rowptr)[rb]).max())
    rec = None
    if records:
        from firedrake_amd.codegen import record_layout
        rec = record_layout([maps[mi].arity for mi in base.staged_maps], [plans[mi][3] for mi in base.staged_maps], rmap.arity,
                            cmap.arity, int(np.diff(csr.rowptr).max()), mpa.maps[0]._base() is mpa.maps[1]._base())
    # flush_colmask: "ocrpm" -- the column lgmap folded into the flush's place table (fd_row_entry_positions_masked)
    assert not flush_colmask or (order is not None and mpa.lgmaps and fixed_point is None)
    src = generate_wrapper(gk, mode_variant(("ocrpm" if flush_colmask else "ocrp") if order is not None else "ocr", 1,
                                            [plans[mi][3] for mi in base.staged_maps], rec) + ("_fx" if fixed_point is not None else ""))
    text = src.source.replace("extern __shared__ __align__(16) unsigned char fd_lds[];", "unsigned char *fd_lds = fd_sim::lds;")
    sig = re.search(r'extern "C" __global__[^\n]*void %s\((.*)\)\n' % src.symbol, text).group(1)
    names = [p.split()[-1].lstrip("*") for p in sig.split(", ")]
    text += ('\nextern "C" void sim_run(int fd_nblocks, int fd_nthreads, %s)\n{\n  fd_sim::run(fd_nblocks, fd_nthreads, [&] { %s(%s); });\n}\n'
             % (sig, src.symbol, ", ".join(names)))
    lib = _compile_mt(text, src.symbol)
    lib.sim_run.restype = None
    keep = []

    def ptr(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return ctypes.c_void_p(a.ctypes.data)

    if not zero_pending:
        csr.values[...] = 1.0                     # accumulate on top of existing values
    elif flush_colmask:
        csr.values[...] = 7.0                     # a pending zero() is no memset: the loop overwrites every entry of its rows
    cargs = [ctypes.c_int(len(rb) - 1), ctypes.c_int(T), ctypes.c_int(0), ctypes.c_int(len(inst_ent))]
    for desc in src.layout:
        kind = desc[0]
        if kind in ("virt_col", "virt_layer"):
            cargs.append(ptr(np.ascontiguousarray(virt.col if kind == "virt_col" else virt.layer, dtype=np.int32)))
        elif kind == "layers":
            cargs.append(ptr(np.asarray(pl.iterset.layers_array, dtype=np.int32)))
        elif kind == "subset":
            cargs.append(ptr(np.asarray(pl.iterset.indices, dtype=np.int32)))
        elif kind == "arg":
            pa = pl.arguments[desc[1]]
            if isinstance(pa, MatParloopArg):
                cargs.append(ctypes.c_void_p(csr.values.ctypes.data))
            else:
                host = pa.data._host if pa.data._host_valid else pa.data._to_host()
                cargs.append(ptr(np.array(host, copy=True)))
        elif kind == "map":
            cargs.append(ptr(np.asarray(base_maps[desc[1]].values_with_halo, dtype=np.int32)))
        elif kind == "bstart":
            cargs.append(ptr(inst_off))
        elif kind == "ocr_inst_ent":
            cargs.append(ptr(inst_ent))
        elif kind == "plan_blkoff":
            cargs.append(ptr(plans[desc[1]][0]))
        elif kind == "plan_list":
            cargs.append(ptr(plans[desc[1]][1]))
        elif kind == "plan_lmap":
            cargs.append(ptr(plans[desc[1]][2]))
        elif kind == "plan_maxnd":
            cargs.append(ctypes.c_longlong(plans[desc[1]][3]))
        elif kind == "ocr_rblk":
            cargs.append(ptr(rb))
        elif kind == "ocr_rowptr":
            cargs.append(ptr(np.ascontiguousarray(csr.rowptr, dtype=NNZ)))
        elif kind == "ocr_kidx":
            cargs.append(ptr(kidx))
        elif kind == "ocr_maxnnz":
            cargs.append(ctypes.c_longlong(max_nnz))
        elif kind == "ocr_maxnown":
            cargs.append(ctypes.c_longlong(int(np.diff(rb).max())))
        elif kind == "ocr_flags":
            cargs.append(ctypes.c_longlong(1 if zero_pending else 0))
        elif kind == "ocr_prowptr":
            cargs.append(ptr(np.ascontiguousarray(prowptr, dtype=NNZ)))
        elif kind == "ocr_nstart":
            ns = np.zeros(max(nrows, 1), dtype=NNZ)
            ns[plist] = prowptr[:-1]
            cargs.append(ptr(ns))
        elif kind == "ocr_gstart":
            cargs.append(ptr(np.ascontiguousarray(csr.rowptr[plist], dtype=NNZ)))
        elif kind == "ocr_srow":
            # numpy restatement of fd_ocr_node_words: per (block, staged node) 1 + offset of the node's row in the block
            # accumulator (0 = not owned / dropped by the row lgmap), bit 31 = column dropped by the column lgmap
            blk_, lst_ = plans[desc[2]][0], plans[desc[2]][1]
            lg = mpa.lgmaps or (None, None)
            nst = None
            if order is not None:
                nst = np.zeros(max(nrows, 1), dtype=np.int64)
                nst[plist] = prowptr[:-1]
            words = np.zeros(max(len(lst_), 1), dtype=np.uint32)
            for b in range(len(rb) - 1):
                n0, n1 = int(rb[b]), int(rb[b + 1])
                r0 = int(prowptr[n0]) if order is not None else int(csr.rowptr[n0])
                nnzb = (int(prowptr[n1]) if order is not None else int(csr.rowptr[n1])) - r0
                for i in range(int(blk_[b]), int(blk_[b + 1])):
                    g, p_ = int(lst_[i]), -1
                    if order is not None:
                        if 0 <= g < nrows and 0 <= nst[g] - r0 < nnzb:
                            p_ = int(nst[g]) - r0
                    elif n0 <= g < n1:
                        p_ = int(csr.rowptr[g]) - r0
                    w = p_ + 1 if (p_ >= 0 and not (lg[0] is not None and lg[0][g] < 0)) else 0
                    if lg[1] is not None and lg[1][g] < 0:
                        w |= 0x80000000
                    if len(desc) > 3 and 0 <= g < len(csr.rowptr) - 1:
                        # fd_ocr_node_diag: place of the diagonal entry inside the node's CSR row, bits 20..27
                        row = csr.colidx[csr.rowptr[g]:csr.rowptr[g + 1]]
                        hit = np.nonzero(row == g)[0]
                        if len(hit):
                            w |= int(hit[0]) << 20
                    words[i] = w
            cargs.append(ptr(words))
        elif kind == "ocr_rec":
            lbits, kbits, diag, words_ = rec
            cargs.append(ptr(pack_records_ref([plans[mi][2] for mi in base.staged_maps], lbits, kidx, rmap.arity, cmap.arity,
                                              kbits, diag, words_)))
        elif kind == "ocr_gpos":
            # place of every accumulator entry (rows in position order) in the CSR value array
            gp = np.full(max(int(prowptr[-1]), 1), -1, dtype=np.int32)
            for p_, r in enumerate(plist):
                gp[prowptr[p_]:prowptr[p_] + plen[p_]] = np.arange(csr.rowptr[r], csr.rowptr[r + 1])
            if flush_colmask:
                gp = row_entry_positions_masked_ref(gp, csr.colidx, np.asarray(mpa.lgmaps[1]))
            cargs.append(ptr(gp))
        elif kind == "ocr_npos":
            cargs.append(ctypes.c_longlong(nrows))
        elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
            cargs.append(ptr(np.asarray(mpa.lgmaps[0 if kind == "mat_row_lgmap" else 1], dtype=np.int32)))
        elif kind == "plan_copy":
            cargs.append(_plan_copy_arg(pl, desc, plans, ptr))
        elif kind == "fx_scale":
            # one fx_block_t {S, 1/S, lim_hi, low_hi, L, pad} per row block: ``fixed_point`` = True (no scale yet: every block runs
            # fp64 and writes its record), a limit exponent L for all blocks (|x| < 2^L, S = 2^(50 - L)), or the record array of an
            # earlier run
            dt = np.dtype([("S", "<f8"), ("invS", "<f8"), ("lim", "<u4"), ("low", "<u4"), ("L", "<i4"), ("pad", "<u4")])
            if isinstance(fixed_point, np.ndarray):
                rec_ = fixed_point.copy()
            else:
                rec_ = np.zeros(len(rb) - 1, dtype=dt)
                if fixed_point is not True:
                    L = int(fixed_point)
                    rec_["S"], rec_["invS"], rec_["L"] = 2.0 ** (50 - L), 2.0 ** (L - 50), L
                    rec_["lim"], rec_["low"] = (L + 1023) << 20, (L - 6 + 1023) << 20
            csr.fx_records = rec_
            cargs.append(ptr(rec_) if False else ctypes.c_void_p(rec_.ctypes.data))
        elif kind == "fx_stat":
            csr.fx_stat = np.zeros(4, dtype=np.uint32)
            cargs.append(ctypes.c_void_p(csr.fx_stat.ctypes.data))
        else:
            raise AssertionError(f"hostsim (ocr) cannot provide {kind}")
    lib.sim_run(*cargs)
    return csr


def run_ocrs(pl, nnz_per_block=96, zero_pending=True, order=None, run_flush=False, records=False, pairs=False):
    """Execute a matrix-assembly Parloop ``pl`` with the ROW-SLICED owner-computes-rows wrapper on the host (one OS thread
    per lane).  Plan tables from helpers.ocrs_plan_ref, CSR pattern from the oracle.  Returns the OracleCSR."""
    import re
    from firedrake_amd.codegen import _ocr_shape, mode_variant
    from helpers import choose_groups_ref, first_touch_ref, ocrs_pair_counts_ref, ocrs_paired_plan_ref, ocrs_plan_ref, plan_ref_blocks
    gk = pl.global_kernel
    assert _ocr_shape(gk, mats_on_virtual=True, allow_unroll=True) is not None
    (k, mpa), = [(k, pa) for k, pa in enumerate(pl.arguments) if isinstance(pa, MatParloopArg)]
    # subsets / extruded sets: the plan lives on the derived maps over the virtual (position x layer) space
    virt = pl._virtual(staged=True)
    base_maps, maps = [], []
    for pa in pl.arguments:
        for m in getattr(pa, "maps", ()):
            if all(m._base() is not q for q in base_maps):
                base_maps.append(m._base())
                maps.append(pl._plan_map(m._base(), staged=True))
    base = generate_wrapper(gk, "ocrs")
    T = base.block_threads
    csr = oracle_pattern(mpa.data.sparsity)             # scalar CSR the values live in (blocks expanded)
    sp = mpa.data.sparsity
    B = sp.dsets[0].cdim * sp.dsets[1].cdim
    ncsr = csr
    if B > 1:                                             # the plan works on the NODE pattern
        import oracle
        ncsr = oracle.build_sparsity(sp.dsets[0].set.total_size, sp.dsets[1].set.total_size,
                                     [(r.values_with_halo, c.values_with_halo) for r, c, _ in sp._pairs], set_diag=sp._has_diagonal)
        assert not pl.iterset._extruded
    rmap, cmap = (pl._plan_map(m._base(), staged=True) for m in mpa.maps)
    nent = virt.size(pl.iterset.size) if virt else pl.iterset.size
    nrows = rmap.toset.size
    plist = pinv = None
    acc = ncsr.rowptr
    acc_by_node = ncsr.rowptr
    if order is not None:
        plist, pinv = first_touch_ref(np.asarray(rmap.values_with_halo), order, nrows)
        acc = np.concatenate([[0], np.cumsum(np.diff(ncsr.rowptr)[:nrows][plist])]).astype(np.int32)
        acc_by_node = np.zeros(max(nrows, 1), dtype=np.int32)
        acc_by_node[plist] = acc[:-1]
    targets = np.arange(0, int(acc[nrows]) + nnz_per_block, nnz_per_block)
    rb = np.unique(np.concatenate([np.searchsorted(acc[:nrows + 1], targets, side="left"), [0, nrows]]))
    rb = rb[rb <= nrows].astype(np.int32)
    lg = mpa.lgmaps
    per_dof = (sp.dsets[0].cdim, sp.dsets[1].cdim) if (lg is not None and gk.arguments[k].unroll) else None
    groups = None
    if pairs:
        # two rows per instance ("_g" variants): the groups from the rows' co-ownership counts, as Parloop._ocrs_geometry picks them
        assert B == 1 and per_dof is None and records
        groups = choose_groups_ref(ocrs_pair_counts_ref(np.asarray(rmap.values_with_halo), 0, nent, rb, pinv=pinv))
        res = ocrs_paired_plan_ref(
            np.asarray(rmap.values_with_halo), np.asarray(cmap.values_with_halo), 0, nent, rb, ncsr.rowptr, ncsr.colidx, acc_by_node, acc,
            groups, pinv=pinv, rlg=None if lg is None else np.asarray(lg[0]), clg=None if lg is None else np.asarray(lg[1]))
        res = res + (None,)
    else:
        res = ocrs_plan_ref(
            np.asarray(rmap.values_with_halo), np.asarray(cmap.values_with_halo), 0, nent, rb, ncsr.rowptr, ncsr.colidx, acc_by_node, acc,
            pinv=pinv, rlg=None if lg is None else np.asarray(lg[0]), clg=None if lg is None else np.asarray(lg[1]), per_dof=per_dof)
    inst_off, inst_ent, chunk_role, valid, slot, kk, rowlen = res[:7]
    plans = {}
    for mi in base.staged_maps:
        blk, lst, lm = plan_ref_blocks(np.asarray(maps[mi].values_with_halo)[inst_ent], inst_off)
        plans[mi] = (blk, lst, np.ascontiguousarray(lm), int(np.diff(blk).max()) if len(blk) > 1 else 0)
    max_nnz = int(np.diff(acc[rb]).max())
    run_tabs = None
    if order is not None and run_flush:
        run_tabs = row_runs_ref(acc, ncsr.rowptr[plist], rb)
    rec = None
    if records:
        from firedrake_amd.codegen import sliced_record_layout
        rec = sliced_record_layout([maps[mi].arity for mi in base.staged_maps], [plans[mi][3] for mi in base.staged_maps], cmap.arity,
                                   int(np.diff(ncsr.rowptr).max()), max_nnz, rows=2 if groups else 1)
    src = generate_wrapper(gk, mode_variant(("ocrspr" if run_tabs else "ocrsp") if order is not None else "ocrs", 1,
                                            [plans[mi][3] for mi in base.staged_maps], rec, groups=groups))
    text = src.source.replace("extern __shared__ __align__(16) unsigned char fd_lds[];", "unsigned char *fd_lds = fd_sim::lds;")
    sig = re.search(r'extern "C" __global__[^\n]*void %s\((.*)\)\n' % src.symbol, text).group(1)
    names = [p.split()[-1].lstrip("*") for p in sig.split(", ")]
    text += ('\nextern "C" void sim_run(int fd_nblocks, int fd_nthreads, %s)\n{\n  fd_sim::run(fd_nblocks, fd_nthreads, [&] { %s(%s); });\n}\n'
             % (sig, src.symbol, ", ".join(names)))
    lib = _compile_mt(text, src.symbol)
    lib.sim_run.restype = None
    keep = []

    def ptr(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return ctypes.c_void_p(a.ctypes.data)

    if not zero_pending:
        csr.values[...] = 1.0
    cargs = [ctypes.c_int(len(rb) - 1), ctypes.c_int(T), ctypes.c_int(0), ctypes.c_int(len(inst_ent))]
    for desc in src.layout:
        kind = desc[0]
        if kind in ("virt_col", "virt_layer"):
            cargs.append(ptr(np.ascontiguousarray(virt.col if kind == "virt_col" else virt.layer, dtype=np.int32)))
        elif kind == "layers":
            cargs.append(ptr(np.asarray(pl.iterset.layers_array, dtype=np.int32)))
        elif kind == "subset":
            cargs.append(ptr(np.asarray(pl.iterset.indices, dtype=np.int32)))
        elif kind == "arg":
            pa = pl.arguments[desc[1]]
            if isinstance(pa, MatParloopArg):
                cargs.append(ctypes.c_void_p(csr.values.ctypes.data))
            else:
                host = pa.data._host if pa.data._host_valid else pa.data._to_host()
                cargs.append(ptr(np.array(host, copy=True)))
        elif kind == "map":
            cargs.append(ptr(np.asarray(base_maps[desc[1]].values_with_halo, dtype=np.int32)))
        elif kind == "bstart":
            cargs.append(ptr(inst_off))
        elif kind == "ocr_inst_ent":
            cargs.append(ptr(inst_ent))
        elif kind == "ocr_rec":
            lbits, kbits, sbits, words_ = rec
            cargs.append(ptr(pack_records_ref([plans[mi][2] for mi in base.staged_maps], lbits, np.asarray(kk).reshape(len(inst_ent), -1),
                                              2 if groups else 1, cmap.arity, kbits, False,
                                              words_, extra=np.asarray(slot), ebits=sbits, sentinel=True)))
        elif kind == "ocrs_chunk_role":
            cargs.append(ptr(chunk_role))
        elif kind == "plan_blkoff":
            cargs.append(ptr(plans[desc[1]][0]))
        elif kind == "plan_list":
            cargs.append(ptr(plans[desc[1]][1]))
        elif kind == "plan_lmap":
            cargs.append(ptr(plans[desc[1]][2]))
        elif kind == "plan_maxnd":
            cargs.append(ctypes.c_longlong(plans[desc[1]][3]))
        elif kind == "ocr_rblk":
            cargs.append(ptr(rb))
        elif kind in ("ocr_rowptr", "ocr_prowptr"):
            cargs.append(ptr(np.ascontiguousarray(acc, dtype=NNZ)))
        elif kind == "ocr_gstart":
            cargs.append(ptr(np.ascontiguousarray(ncsr.rowptr[plist], dtype=NNZ)))
        elif kind in ("ocr_grun", "ocr_brun", "ocr_rdelta"):
            cargs.append(ptr(run_tabs[{"ocr_grun": 0, "ocr_brun": 1, "ocr_rdelta": 2}[kind]]))
        elif kind == "plan_copy":
            cargs.append(_plan_copy_arg(pl, desc, plans, ptr))
        elif kind == "ocr_gpos":
            # (scalar matrices only: place of every accumulator entry, rows in position order)
            cargs.append(ptr(np.concatenate([np.arange(ncsr.rowptr[r], ncsr.rowptr[r + 1]) for r in plist] + [np.zeros(0, np.int64)]).astype(np.int32)))
        elif kind == "ocrs_slot":
            cargs.append(ptr(slot))
        elif kind == "ocrs_rowlen":
            cargs.append(ptr(rowlen))
        elif kind == "ocrs_rmask":
            cargs.append(ptr(res[7]))
        elif kind == "ocrs_cmask":
            cargs.append(ptr(res[8]))
        elif kind == "ocrs_kk":
            cargs.append(ptr(kk))
        elif kind == "ocr_maxnnz":
            cargs.append(ctypes.c_longlong(max_nnz))
        elif kind == "ocr_flags":
            cargs.append(ctypes.c_longlong(1 if zero_pending else 0))
        else:
            raise AssertionError(f"hostsim (ocrs) cannot provide {kind}")
    lib.sim_run(*cargs)
    return csr


def run_direct(pl, part=None):
    """Execute Parloop ``pl`` over ``part`` = (offset, size) (default: all owned entities) with the direct
    wrapper on the host.  Returns one array per argument: COPIES of Dat/Global data after the loop, or an
    OracleCSR holding the assembled values for a Mat.  The carriers are not modified."""
    old = configuration["mat_scatter"]
    configuration["mat_scatter"] = "search"          # the element->nz table is built on the device
    try:
        src = generate_wrapper(pl.global_kernel, "direct")
    finally:
        configuration["mat_scatter"] = old
    lib = _compile(src.source, src.symbol)
    fn = getattr(lib, src.symbol)
    fn.restype = None
    offset, size = part if part is not None else (0, pl.iterset.size)
    maps = []
    for pa in pl.arguments:
        for m in getattr(pa, "maps", ()):
            if all(m._base() is not q for q in maps):
                maps.append(m._base())
    outs, keep = {}, []

    def ptr(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return ctypes.c_void_p(a.ctypes.data)

    cargs = [ctypes.c_int(offset), ctypes.c_int(offset + size)]
    for desc in src.layout:
        kind = desc[0]
        if kind in ("virt_col", "virt_layer"):
            cargs.append(ptr(np.ascontiguousarray(virt.col if kind == "virt_col" else virt.layer, dtype=np.int32)))
        elif kind == "layers":
            cargs.append(ptr(np.asarray(pl.iterset.layers_array, dtype=np.int32)))
        elif kind == "subset":
            cargs.append(ptr(np.asarray(pl.iterset.indices, dtype=np.int32)))
        elif kind == "arg":
            pa = pl.arguments[desc[1]]
            if isinstance(pa, MatParloopArg):
                csr = oracle_pattern(pa.data.sparsity)
                outs[desc[1]] = csr
                cargs.append(ctypes.c_void_p(csr.values.ctypes.data))
            elif isinstance(pa, (DatParloopArg, GlobalParloopArg)):
                host = pa.data._host if pa.data._host_valid else pa.data._to_host()     # a DatView: its parent's rows
                a = np.array(host, copy=True)
                outs[desc[1]] = a
                cargs.append(ctypes.c_void_p(a.ctypes.data))
            else:
                cargs.append(ctypes.c_void_p(0))
        elif kind == "map":
            cargs.append(ptr(np.asarray(maps[desc[1]].values_with_halo, dtype=np.int32)))
        elif kind == "bstart":
            cargs.append(ctypes.c_void_p(0))
        elif kind == "mat_rowptr":
            rp64 = np.ascontiguousarray(outs[desc[1]].rowptr, dtype=NNZ)
            keep.append(rp64)
            cargs.append(ctypes.c_void_p(rp64.ctypes.data))
        elif kind == "mat_colidx":
            cargs.append(ctypes.c_void_p(outs[desc[1]].colidx.ctypes.data))
        elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
            pa = pl.arguments[desc[1]]
            cargs.append(ptr(np.asarray(pa.lgmaps[0 if kind == "mat_row_lgmap" else 1], dtype=np.int32)))
        else:
            raise AssertionError(f"hostsim cannot provide {kind}")
    fn(*cargs)
    return [outs.get(k) for k in range(len(pl.arguments))]


def run_tensor(pl, initial=None):
    """Execute a tensor-product Parloop ``pl`` (codegen modes tp_action / tp_matrix) on the host: the generated wrapper and the
    REAL device templates of firedrake_amd/csrc/fd_tensor.h, compiled against the stand-in header (one OS thread per lane;
    the fp64 MFMA is restated there from the operand layout the templates rely on).  Returns one entry per argument like
    run_direct: copies of the Dats after the loop, an OracleCSR for the Mat.  ``initial``: values of the Mat before the loop (default
    zeros)."""
    import re
    from firedrake_amd.codegen import generate_tensor_wrapper
    from firedrake_amd.tensor import gll_gauss_tables
    gk = pl.global_kernel
    src = generate_tensor_wrapper(gk)
    csrc = os.path.join(os.path.dirname(_HERE), "firedrake_amd", "csrc")
    with open(os.path.join(csrc, "fd_tensor.h")) as f:
        templates = f.read().replace('#include "fd_wrapper.h"', "")
    text = src.source.replace('#include "fd_tensor.h"', '#include "fd_wrapper.h"\n' + templates)
    sig = re.search(r'extern "C" __global__[^\n]*void %s\((.*)\)\n' % src.symbol, text).group(1)
    names = [p.split()[-1].lstrip("*") for p in sig.split(", ")]
    text += ('\nextern "C" void sim_run(int fd_nblocks, int fd_nthreads, %s)\n{\n  fd_sim::run(fd_nblocks, fd_nthreads, [&] { %s(%s); });\n}\n'
             % (sig, src.symbol, ", ".join(names)))
    lib = _compile_mt(text, src.symbol)
    lib.sim_run.restype = None
    maps = []
    for pa in pl.arguments:
        for m in getattr(pa, "maps", ()):
            if all(m._base() is not q for q in maps):
                maps.append(m._base())
    start, end = 0, pl.iterset.size
    nl = pl.iterset.layers - 1
    ncell = (end - start) * nl
    vd = int(gk.local_kernel.tp.get("vdim", 1))
    from firedrake_amd.codegen import tensor_matrix_groups
    nblocks = ncell * tensor_matrix_groups(src.tp, vd) if src.mode == "tp_matrix" else -(-ncell // src.tp["action_cells"])
    outs, keep = {}, []

    def ptr(a):
        a = np.ascontiguousarray(a)
        keep.append(a)
        return ctypes.c_void_p(a.ctypes.data)

    csr = None
    cargs = [ctypes.c_int(nblocks), ctypes.c_int(src.block_threads), ctypes.c_int(start), ctypes.c_int(end)]
    for desc in src.layout:
        kind = desc[0]
        if kind in ("virt_col", "virt_layer"):
            cargs.append(ptr(np.ascontiguousarray(virt.col if kind == "virt_col" else virt.layer, dtype=np.int32)))
        elif kind == "layers":
            cargs.append(ptr(np.asarray(pl.iterset.layers_array, dtype=np.int32)))
        elif kind == "arg":
            pa = pl.arguments[desc[1]]
            if isinstance(pa, MatParloopArg):
                csr = oracle_pattern(pa.data.sparsity)
                if initial is not None:
                    csr.values[:] = initial
                outs[desc[1]] = csr
                cargs.append(ctypes.c_void_p(csr.values.ctypes.data))
            else:
                host = pa.data._host if pa.data._host_valid else pa.data._to_host()
                a = np.array(host, copy=True)
                outs[desc[1]] = a
                cargs.append(ctypes.c_void_p(a.ctypes.data))
        elif kind == "map":
            cargs.append(ptr(np.asarray(maps[desc[1]].values_with_halo, dtype=np.int32)))
        elif kind == "mat_node_rowptr":
            # node-level row starts: scalar row (node, p) of a (D, D)-blocked Mat starts at node_rowptr[node]*D*D + p*rowlen*D
            node_rp = np.asarray(csr.rowptr[::vd] // (vd * vd), dtype=NNZ)
            node_ci = [np.asarray(csr.colidx[csr.rowptr[n * vd]:csr.rowptr[n * vd + 1]][::vd] // vd) for n in range(len(node_rp) - 1)]
            cargs.append(ptr(node_rp))
        elif kind == "tp_offtab":
            # Parloop._tp_offtab restated: position of entry (i, j) inside its CSR row for the bottom / an interior / the top cell
            m = pl.arguments[desc[1]].maps[0]._base()
            mv, off = np.asarray(m.values_with_halo, dtype=np.int64), np.asarray(m.offset, dtype=np.int64)
            nd = m.arity
            tab = np.zeros((mv.shape[0], 3, nd, nd), dtype=np.uint16)
            for c in range(mv.shape[0]):
                for v, lay in enumerate((0, min(1, nl - 1), nl - 1)):
                    nodes = mv[c] + off * lay
                    for i, rn in enumerate(nodes):
                        cols = node_ci[rn]
                        pos = np.searchsorted(cols, nodes)
                        assert (cols[pos] == nodes).all()
                        tab[c, v, i] = pos
            cargs.append(ptr(tab))
        elif kind in ("mat_row_lgmap", "mat_col_lgmap"):
            pa = pl.arguments[desc[1]]
            cargs.append(ptr(np.asarray(pa.lgmaps[0 if kind == "mat_row_lgmap" else 1], dtype=np.int32)))
        elif kind == "tp_tables":
            tp = gk.local_kernel.tp
            L, DL, qp, qw = gll_gauss_tables(tp["degree"], tp["nq"])
            cargs.append(ptr(np.concatenate([L.ravel(), DL.ravel(), qp, qw])))
        else:
            raise AssertionError(f"hostsim (tensor) cannot provide {kind}")
    lib.sim_run(*cargs)
    return [outs.get(k) for k in range(len(pl.arguments))]
