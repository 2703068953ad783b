"""Environment-backed configuration (mirror of pyop2/configuration.py:41-166).

Two kinds of entries: switches a user or a measurement script sets from the environment (``FDHIP_*``, the analogue of ``PYOP2_*``), and
tuning constants that are part of the design (block sizes, thresholds measured once and recorded in DESIGN.md) -- plain values here,
reachable by the tests through the dictionary, not by the environment."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _env(name, default, conv=str):
    v = os.environ.get(name)
    return conv(v) if v is not None else default


configuration = {
    # ---- environment switches -------------------------------------------------------------------------------------------------
    # hipcc analogue of PYOP2_CFLAGS / PYOP2_CACHE_DIR (pyop2/configuration.py:83-115)
    "hipcc": _env("FDHIP_HIPCC", "/opt/rocm/bin/hipcc"),
    "arch": _env("FDHIP_ARCH", "gfx950"),
    "cflags": _env("FDHIP_CFLAGS", ""),
    "cache_dir": _env("FDHIP_CACHE_DIR", os.path.join(_HERE, "_cache")),
    "debug": _env("FDHIP_DEBUG", 0, int),
    "trace": _env("FDHIP_TRACE", 0, int),     # roctx range + flop log per parloop (profiling.py; pyop2/parloop.py:219-232)
    "type_check": _env("FDHIP_TYPE_CHECK", 1, int),
    "phase_times": _env("FDHIP_PHASE_TIMES", 0, int),   # profiling aid: per-block wall-clock stamps of the owner-computes-rows phases
    "mode": _env("FDHIP_MODE", "auto"),                 # auto | staged | direct
    "lds_limit": _env("FDHIP_LDS_LIMIT", 64 * 1024, int),      # LDS budget of a staged / owner-computes-rows block
    "prefetch": _env("FDHIP_PREFETCH", 1, int),          # software-pipeline the packed index rows (first trip's ahead of the staging phase)
    "plan_copies": _env("FDHIP_PLAN_COPIES", 1, int),     # READ Dats unchanged between calls are kept in plan order and streamed
    "locality_order": _env("FDHIP_LOCALITY_ORDER", 1, int),   # maps without producer hints: blocks derived from the loop's position field
    "tensor_wrappers": _env("FDHIP_TENSOR_WRAPPERS", 1, int),   # MFMA matrix / sum-factorised action for TensorProductLocalKernels
    "mat_ocr": _env("FDHIP_MAT_OCR", 1, int),             # owner-computes-rows matrix assembly (no global atomics)
    "ocr_sliced": _env("FDHIP_OCR_SLICED", 1, int),       # row-sliced instances (entity, local row) for large element matrices
    # row-sliced loops on scalar matrices: TWO local rows per instance sharing one evaluation of the local kernel and one index
    # record (fd_ocrplan_create_paired; the pairs are picked from the rows' co-ownership counts), 0 = one row per instance
    "ocrs_pairs": _env("FDHIP_OCRS_PAIRS", 1, int),
    "ocr_records": _env("FDHIP_OCR_RECORDS", 1, int),     # one bit-packed record per instance (0 = uint16 / uint8 index rows)
    # 1 = whole-entity owner-computes-rows loops (scalar fp64 matrices) reduce their element matrices in LDS as CHECKED 64-bit
    # fixed-point sums through integer atomics (codegen "_fx": exact, order-independent sums of contributions ROUNDED to a quantum of
    # 2^-44..2^-47 of the row block's largest contribution; blocks whose largest contribution leaves the window of their scale redo
    # their rows in fp64 inside the launch).  The guarantee is NORMWISE PER ROW BLOCK only -- rows whose entries are far below the
    # block's largest lose relative accuracy in proportion -- so it is OPT-IN: the default (0) adds in fp64 (ds_add_f64), like the
    # reference's MatSetValuesLocal(..., ADD_VALUES) (pyop2/codegen/builder.py:573-625), and keeps every row accurate relative to ITS
    # OWN entries (tests/test_gpu_graded_mesh.py).  Measured gain of the opt-in: 0-3 % of the P1 Jacobian
    "ocr_fixed_point": _env("FDHIP_OCR_FIXED_POINT", 0, int),
    # whole-entity loops over a derived row order: the column lgmap (BC columns) is folded into the flush's place table ("ocrpm",
    # fd_row_entry_positions_masked) instead of a select per contribution in the main loop; 0 = select in the loop ("ocrp")
    "ocr_flush_colmask": _env("FDHIP_OCR_FLUSH_COLMASK", 1, int),
    "ocr_nnz_per_block": _env("FDHIP_OCR_NNZ", 2048, int),  # whole-entity row-block size (CSR entries) when the producer gives no hint
    # the same under a backend-derived row order: the largest leaves whose blocks still fit three to a CU (4416 entries of P1 rows on
    # tetrahedra = 288 rows = 52.9 KB of LDS; a plan that comes out above 53 KB is rebuilt on smaller leaves, parloop._ocr_geometry)
    "ocr_nnz_per_block_ordered": _env("FDHIP_OCR_NNZ_ORDERED", 4416, int),
    "ocrs_nnz_per_block": _env("FDHIP_OCRS_NNZ", 4096, int),     # row-sliced loops: accumulator entries per row block (x8 bytes of LDS)
    "ocrs_block_threads": _env("FDHIP_OCRS_BLOCK_THREADS", 256, int),
    # a wrapper that comes out of hipcc with scratch memory is recompiled with this LLVM -unroll-threshold (0 = never) and the
    # result kept if the scratch shrinks: element tensors must end up in registers (kernel.GlobalKernel._unrolled_variant)
    "unroll_retry_threshold": _env("FDHIP_UNROLL_RETRY", 30000, int),
    # occupancy-directed variants: a staged/OCR wrapper whose register count leaves room for one more resident workgroup
    # per CU is recompiled with the matching __launch_bounds__ and kept if that costs at most this many bytes of scratch
    # per lane (-1 = off).  DG-advection interior-facet loop: 172 -> 128 VGPRs, 12 B scratch, 0.50 -> 0.32 ms
    "auto_occupancy_scratch": _env("FDHIP_AUTO_OCCUPANCY_SCRATCH", 16, int),
    # ---- tuning constants (measured once, not switches) ------------------------------------------------------------
    "block_threads": 0,                 # staged loops: 0 = 256 lanes (512 for maps of arity >= 8)
    "ents_per_block": 1024,             # staged loops without hints: entities per block
    "ocr_block_threads": 0,             # whole-entity owner-computes-rows: 0 = 512 lanes
    "ocr_lds_limit": 0,                 # 0 = lds_limit (the whole CU for element matrices above 32 entries kept whole)
    "ocr_fx_headroom": 3,               # fixed-point scales: bits between a block's largest contribution and the limit of its scale
    "ocr_records_diag": 1,              # records without the diagonal offsets (they ride in the row node's LDS word)
    "staged_direct_noreuse": 1,         # staged loops: Dat arguments on maps without reuse inside a block bypass LDS ("_d" variants)
    # staged blocks: a lane stages up to this many nodes per batch (and at most ~32 doubles) -- all node ids, then all rows, then LDS --
    # and preloads the node ids of the flush ahead of the main loop's last barrier; index rows travel as raw words (0 = one node
    # per trip, decode at the load: the wrappers of rounds 1-5)
    "stage_batch": 8,
    "lane_strided": 1,                  # plans in lane order (fd_plan_set_lane_order)
    # staged loops: node stride of the LDS arrays compiled in (P1 residual 0.43 -> 0.41 ms), rounded up to a multiple of this: the
    # stride is part of the variant's name, so an exact stride (1) sends every mesh whose blocks hold a few nodes more or less
    # through hipcc again (0.35 s per loop); 16 shares a code object between meshes of alike blocks at < 1 KB of LDS and no measurable
    # time (profiles/r6s3_ab_lds_stride_quantum.txt); 0 = run-time stride
    "lds_const_stride": 16,
    "tp_action_waves": 3,
    # MFMA matrix template (csrc/fd_tensor.h): a 16-row panel of more than tp_max_panel_tiles column tiles is cut into chunks of
    # tp_chunk_tiles (Q5: 14 tiles -> 2 chunks of 7, Q6: 22 -> 3 of 8; 4 accumulator registers per tile); per-point weights beyond tp_weight_lds bytes
    # are computed one q1 plane at a time
    "tp_max_panel_tiles": 8,
    "tp_chunk_tiles": 8,
    "tp_weight_lds": 48 * 1024,               # wavefronts per SIMD the tensor-product action wrapper is compiled for
    "ocrs_lds_limit": 0,                # row-sliced blocks: their own LDS budget (0 = lds_limit)
    "ocr_sliced_min_arity": 8,          # scalar rows of the element matrix from which row-sliced instances pay (P1: 4, whole; P2: 10, sliced)
    "ocr_sliced_max_arity": 32,
    "ocr_sliced_max_entries": 1024,
    "ocrs_run_flush": 1,                # derived row orders, scalar matrices: run-coded places (1 B per entry) instead of row by row
    "ocrs_interleave": 7,               # stride permutation of the instances of every (block, row index) group
    # order of the instances inside an owner-computes-rows block: "stencil" (sorted by ownership pattern and owned-row signature)
    # or "natural" (entity order: what the numpy restatements of the tests are written in)
    "ocr_order": "stencil",
    "unroll_retry_max_scratch": 8192,   # bytes per lane; beyond: too large for registers anyway
    "min_waves": 0,                     # 2nd __launch_bounds__ argument (waves per SIMD), 0 = unset
    "locality_min_entities": 8192,      # loops below this size keep the caller's order
    "locality_tile_entities": 1536,     # entities per leaf of the derived order
    # loops too small to fill the device with leaves of that size get smaller ones: at least this many blocks (4 per CU), leaves of
    # no less than 256 entities / 32 rows (0 = off).  tools/size_sweep.py, profiles/r6s3_size_sweep*.txt: the C2 step on cubes of
    # 16..48 per axis -27..-8 % (n = 48: 0.0433 -> 0.0398 ms at 1024 blocks, 0.0425 at 2048, 0.0455 at 4096), C1 0.0231 -> 0.0187 ms;
    # from 1.5 M cells on the leaves have their full size
    "small_loop_blocks": 1024,
    "use_preferred_blocks": 1,          # plan blocks = the producer's traversal tiles when a Map carries them
    "mat_scatter": "table",             # direct wrapper: element->nonzero table | "search" (row search, what hostsim runs)
    "mat_staged": 1,                    # staged wrapper: reduce element matrices in LDS (matrix plans)
}
