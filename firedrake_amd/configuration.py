"""Environment-backed configuration (mirror of pyop2/configuration.py:41-166)."""
import os

_HERE = os.path.dirname(os.path.abspath(__file__))


def _env(name, default, conv=str):
    v = os.environ.get(name)
    return conv(v) if v is not None else default


configuration = {
    # hipcc analogue of PYOP2_CFLAGS / PYOP2_CACHE_DIR (pyop2/configuration.py:83-115)
    "hipcc": _env("FDHIP_HIPCC", "/opt/rocm/bin/hipcc"),
    "arch": _env("FDHIP_ARCH", "gfx950"),
    "cflags": _env("FDHIP_CFLAGS", ""),
    "cache_dir": _env("FDHIP_CACHE_DIR", os.path.join(_HERE, "_cache")),
    "debug": _env("FDHIP_DEBUG", 0, int),
    "trace": _env("FDHIP_TRACE", 0, int),     # roctx range + flop log per parloop (profiling.py; pyop2/parloop.py:219-232)
    "type_check": _env("FDHIP_TYPE_CHECK", 1, int),
    "phase_times": _env("FDHIP_PHASE_TIMES", 0, int),   # profiling aid: per-block wall-clock stamps of the staged / owner-computes-rows phases
    # wrapper generation
    "mode": _env("FDHIP_MODE", "auto"),                 # auto | staged | direct
    "block_threads": _env("FDHIP_BLOCK_THREADS", 0, int),
    "ents_per_block": _env("FDHIP_ENTS_PER_BLOCK", 1024, int),
    "flush_batch": _env("FDHIP_FLUSH_BATCH", 4, int),      # flushes through index tables (derived row orders): table loads requested per trip
    "flush_preload": _env("FDHIP_FLUSH_PRELOAD", 1, int),  # ... a block's whole table requested ahead of the barrier that ends its main loop
    "plan_copies": _env("FDHIP_PLAN_COPIES", 1, int),     # READ Dats unchanged between calls are kept in plan order and streamed
    # whole-entity owner-computes-rows loops (scalar fp64 matrices) reduce their element matrices in LDS as CHECKED 64-bit
    # fixed-point sums through integer atomics (codegen "_fx"; exact, order-independent; blocks that meet a contribution beyond
    # the scale's limit redo their rows in fp64 inside the launch); 0 = fp64 atomics (ds_add_f64)
    "ocr_fixed_point": _env("FDHIP_OCR_FIXED_POINT", 1, int),
    "ocr_fx_headroom": _env("FDHIP_OCR_FX_HEADROOM", 3, int),   # bits between the largest contribution seen and the limit
    # owner-computes-rows index tables: one bit-packed record per instance (fd_ocr_pack_records; 0 = uint16 / uint8 rows), the
    # diagonal offsets taken from the row node's LDS word, and the flush of a derived row order run-coded (fd_ocr_row_runs)
    "ocr_records": _env("FDHIP_OCR_RECORDS", 1, int),
    "ocr_records_diag": _env("FDHIP_OCR_RECORDS_DIAG", 1, int),
    "ocr_lds_limit": _env("FDHIP_OCR_LDS_LIMIT", 0, int),  # 0 = auto (whole CU for large element matrices)
    "lane_strided": _env("FDHIP_LANE_STRIDED", 1, int),   # plans in lane order (fd_plan_set_lane_order)
    # staged rows addressed with a COMPILE-TIME node stride (max nodes per block rounded up to a multiple of this value;
    # 0 = run-time stride): the LDS offsets of all staged arrays fold into ds_read/ds_add immediates instead of one
    # v_add_u32 per access
    "lds_const_stride": _env("FDHIP_LDS_CONST_STRIDE", 1, int),     # staged loops: P1 residual 0.43 -> 0.41 ms
    "prefetch": _env("FDHIP_PREFETCH", 1, int),          # software-pipeline the packed index rows
    "early_loads": _env("FDHIP_EARLY_LOADS", 1, int),    # first trip's index rows requested ahead of the staging phase; accumulators
                                                         # zeroed by capacity (no second level of dependent scalar loads before the loop)
    "tensor_wrappers": _env("FDHIP_TENSOR_WRAPPERS", 1, int),   # MFMA matrix / sum-factorised action for TensorProductLocalKernels
    "tp_action_waves": _env("FDHIP_TP_ACTION_WAVES", 3, int),   # wavefronts per SIMD the action wrapper is compiled for (register cap; 0 = none)
    "tp_store_single_rows": _env("FDHIP_TP_STORE_SINGLE_ROWS", 0, int),   # a zeroed tensor-product Mat: zero the shared rows only, store the rest
                                                                        # (measured 1 % slower than fill + atomics: profiles/r4n_c3_single_rows.txt)
    "mat_ocr": _env("FDHIP_MAT_OCR", 1, int),             # owner-computes-rows matrix assembly (no global atomics)
    "ocr_nnz_per_block_ordered": _env("FDHIP_OCR_NNZ_ORDERED", 3840, int),   # the same under a backend-derived row order
    "ocr_nnz_per_block": _env("FDHIP_OCR_NNZ", 2048, int),  # row-block size (CSR entries) when the producer gives no hint
    # row-sliced owner-computes-rows (codegen.generate_sliced_wrapper): instances are (entity, local row), the local kernel is
    # instantiated once per row; pays when the rows of the element matrix dominate its shared (geometry) part
    "ocr_sliced": _env("FDHIP_OCR_SLICED", 1, int),
    "ocr_sliced_min_arity": _env("FDHIP_OCR_SLICED_MIN_ARITY", 8, int),
    "ocr_sliced_max_arity": _env("FDHIP_OCR_SLICED_MAX_ARITY", 32, int),
    "ocr_sliced_max_entries": _env("FDHIP_OCR_SLICED_MAX_ENTRIES", 1024, int),
    "ocrs_prefetch": _env("FDHIP_OCRS_PREFETCH", 1, int),         # row-sliced loops: index rows requested 1 or 2 trips ahead
    "ocrs_run_flush": _env("FDHIP_OCRS_RUN_FLUSH", 1, int),       # derived row orders, scalar matrices: run-coded places (1 B per entry)
    "ocrs_nnz_per_block": _env("FDHIP_OCRS_NNZ", 4096, int),     # accumulator entries per row block (x8 bytes of LDS)
    "ocrs_block_threads": _env("FDHIP_OCRS_BLOCK_THREADS", 256, int),
    "ocrs_interleave": _env("FDHIP_OCRS_INTERLEAVE", 7, int),    # > 1: stride permutation of the instances of every (block, row index) group
    "ocr_block_threads": _env("FDHIP_OCR_BLOCK_THREADS", 0, int),  # 0 = auto: 512 for small element matrices, else block_threads
    # order of the instances inside an owner-computes-rows block: "stencil" (sorted by ownership pattern and owned-row
    # signature) or "natural" (entity order: what the numpy restatements of the tests are written in)
    "ocr_order": _env("FDHIP_OCR_ORDER", "stencil"),
    # a wrapper that comes out of hipcc with scratch memory is recompiled with this LLVM -unroll-threshold (0 = never) and the
    # result kept if the scratch shrinks: element tensors must end up in registers (kernel.GlobalKernel._unrolled_variant)
    "unroll_retry_threshold": _env("FDHIP_UNROLL_RETRY", 30000, int),
    "unroll_retry_max_scratch": _env("FDHIP_UNROLL_RETRY_MAX_SCRATCH", 8192, int),   # bytes per lane; beyond: too large for registers anyway
    "min_waves": _env("FDHIP_MIN_WAVES", 0, int),       # 2nd __launch_bounds__ argument (waves per SIMD), 0 = unset
    # occupancy-directed variants: a staged/OCR wrapper whose register count leaves room for one more resident workgroup
    # per CU is recompiled with the matching __launch_bounds__ and kept if that costs at most this many bytes of scratch
    # per lane (-1 = off).  DG-advection interior-facet loop: 172 -> 128 VGPRs, 12 B scratch, 0.50 -> 0.32 ms
    "auto_occupancy_scratch": _env("FDHIP_AUTO_OCCUPANCY_SCRATCH", 16, int),
    # maps without producer hints: derive the entity order of staged loops from the loop's position field (box tiles of a
    # uniform grid over the entity centroids, fd_locality_order) instead of cutting the caller's order into uniform blocks
    "locality_order": _env("FDHIP_LOCALITY_ORDER", 1, int),
    "locality_min_entities": _env("FDHIP_LOCALITY_MIN", 8192, int),
    "locality_tile_entities": _env("FDHIP_LOCALITY_TILE", 1536, int),    # entities per box tile of the derived order
    "use_preferred_blocks": _env("FDHIP_PREFERRED_BLOCKS", 1, int),   # plan blocks = the producer's traversal tiles
    "lds_limit": _env("FDHIP_LDS_LIMIT", 64 * 1024, int),
    "mat_scatter": _env("FDHIP_MAT_SCATTER", "table"),  # table | search (direct scatter flavours)
    "mat_staged": _env("FDHIP_MAT_STAGED", 1, int),     # reduce element matrices in LDS (matrix plans)
}
