/*
 * fdhip.h -- C ABI of libfdhip.so: the MI355X (gfx950) backend that sits behind
 * Firedrake's assemble() / pyop2.parloop.
 *
 * Every entry point replaces one piece of the reference's hot path.  Citations are
 * file:line relative to the reference tree (firedrakeproject/firedrake).
 *
 * Conventions
 *   - plain pointers and sizes only; device pointers are ordinary `void*`
 *   - every function returns 0 on success, non-zero on failure; the message is
 *     available from fd_last_error() (thread-local).  The reference ignores the
 *     wrapper's int return (pyop2/global_kernel.py:334-335,455) and surfaces
 *     failures as Python exceptions before launch; the Python host raises on any
 *     non-zero status.
 *   - IntType is int32 (pyop2/datatypes.py:6-8 default), ScalarType is float64
 *     (tsfc/parameters.py:19).
 *   - fd_stream_t is a hipStream_t (0 = default stream); nothing synchronises the
 *     host unless it says so.
 */
#ifndef FDHIP_H
#define FDHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Row starts of CSR value arrays (rowptr, node_rowptr, gstart) and of the owner-computes-rows accumulators (prowptr, nstart, the
 * displacements rdelta): PETSc's IntType for nnz (pyop2/datatypes.py:6-10; the 64-bit-index CI variant, .github/workflows/
 * core.yml:276).  64 bits, so that a pattern may hold more than 2^31 entries (BASELINE configs[4] on one GPU: 2.3e9; Q4 at
 * n = 64: 3.7e9).  Column indices, row / node numbers and every offset inside a row or a block stay int32_t. */
typedef int64_t fd_nnz_t;
typedef void *fd_stream_t;
typedef struct fd_kernel_s *fd_kernel_t;
typedef struct fd_plan_s *fd_plan_t;
typedef struct fd_event_s *fd_event_t;

/* ------------------------------------------------------------------ runtime */
int fd_version(void);
const char *fd_last_error(void);
int fd_device_count(int *n);
int fd_set_device(int device);
int fd_device_info(int device, char *name, size_t name_len, int *compute_units,
                   size_t *hbm_bytes, int *lds_bytes_per_block);      /* device < 0: the calling thread's current device */

/* Device buffers standing in for the numpy/PETSc buffers whose raw pointers the
 * reference hands to the wrapper: Dat  pyop2/types/dat.py:94-96, Map  map.py:57-59,
 * Global glob.py:32-33, Subset set.py:434-436, layers set.py:351-353. */
int fd_malloc(void **ptr, size_t bytes);
int fd_free(void *ptr);
int fd_memset(void *ptr, int byte, size_t bytes, fd_stream_t s);   /* Dat.zero dat.py:297-311, Mat.zero mat.py:851-855 */
int fd_memcpy_h2d(void *dst, const void *src, size_t bytes, fd_stream_t s);
int fd_memcpy_d2h(void *dst, const void *src, size_t bytes, fd_stream_t s);
int fd_memcpy_d2d(void *dst, const void *src, size_t bytes, fd_stream_t s);
int fd_stream_create(fd_stream_t *s);
int fd_stream_destroy(fd_stream_t s);
int fd_stream_sync(fd_stream_t s);
int fd_device_sync(void);
/* Stream every later call with a NULL stream argument runs on (NULL = the HIP null stream again), and the
 * event edge between two streams.  The reference runs its parloops one after another on the host
 * (pyop2/parloop.py:243-260); on the device two loops that share no written argument -- assemble(F) and
 * assemble(J) of one Newton step -- may be put on two streams and overlap (firedrake_amd/op2types.py: stream()). */
int fd_stream_set_default(fd_stream_t s);
/* the stream NULL-stream calls run on right now (NULL = the HIP null stream; the capturing stream between fd_graph_begin and
 * fd_graph_end): what a fork onto a side stream hands back to fd_stream_set_default when it ends, so that forks work inside a
 * captured step too (two branches of the graph: pyop2/parloop.py:243-260 has no counterpart, its loops are host-serial) */
int fd_stream_get_default(fd_stream_t *s);
int fd_stream_wait_event(fd_stream_t s, fd_event_t e);

/* HIP events recorded on the launch stream (PETSc Log.Event analogue of
 * pyop2/parloop.py:221-232; used by bench.py for per-kernel durations). */
int fd_event_create(fd_event_t *e);
int fd_event_destroy(fd_event_t e);
int fd_event_record(fd_event_t e, fd_stream_t s);
int fd_event_sync(fd_event_t e);
int fd_event_elapsed_ms(fd_event_t start, fd_event_t stop, float *ms);

/* Named ranges around the launches of one parloop: the PETSc log event `Parloop_<iterset>_<kernel>` of pyop2/parloop.py:219-232
 * (pyop2/profiling.py timed_region) as roctx markers (roctxRangePushA / roctxRangePop, libroctx64 bound at run time), shown by
 * `rocprofv3 --marker-trace` above the wrapper kernels.  fd_trace_available() = 1 when the library could be bound. */
int fd_trace_available(void);
int fd_trace_range_push(const char *name);
int fd_trace_range_pop(void);

/* hipGraph capture of one assembly step.  Between fd_graph_begin and fd_graph_end every entry point called with
 * a NULL stream records into the graph instead of executing; fd_graph_launch then replays the whole step
 * (zeroing, wrapper kernels, BC fix-up) with one host call.  For launch-bound problem sizes (config C1), where
 * the reference pays Python + ctypes overhead per parloop (pyop2/parloop.py:203-232). */
typedef struct fd_graph_s *fd_graph_t;
int fd_graph_begin(fd_graph_t *out);
int fd_graph_end(fd_graph_t g);
int fd_graph_launch(fd_graph_t g, fd_stream_t s);   /* s = NULL: the graph's own stream */
/* replay on the stream NULL-stream calls run on (the HIP null stream unless fd_stream_set_default changed it): ordered with the
 * eager launches and the synchronous copies before and after it, like the host-serial parloops of pyop2/parloop.py:243-260 */
int fd_graph_launch_default(fd_graph_t g);
int fd_graph_sync(fd_graph_t g);
int fd_graph_free(fd_graph_t g);

/* ------------------------------------------------- wrapper kernels (the hot loop)
 * Replaces  compilation.load(...) -> ctypes.CDLL -> wrap_<kernel>  of
 * pyop2/global_kernel.py:426-456 / pyop2/compilation.py:424-455.
 *
 * fd_kernel_load      load a JIT-built gfx950 code object (.hsaco) and look up the
 *                     wrapper symbol `wrap_<local kernel name>` (global_kernel.py:344-346).
 * fd_kernel_builtin   same handle type for wrappers compiled ahead of time into
 *                     libfdhip.so (the five benchmark configurations).
 * fd_kernel_launch    = func(start, end, *arglist)   pyop2/parloop.py:224-232.
 *                     `args` holds the SAME positional list the reference builds in
 *                     Parloop.arglist (parloop.py:203-212; order builder.py:962-981):
 *                     [layers*], [subset*], one pointer per Dat/Global/Mat-array,
 *                     one pointer per distinct Map, followed by backend-private plan
 *                     arrays / scalars.  Every element is 8 bytes (device pointer or
 *                     int64 scalar).  The launch is asynchronous on `s`.
 *     ents_per_block  iteration-set entities handled by one workgroup
 *     nblocks         <=0: derived as ceil((end-start)*layers/ents_per_block)
 *     lds_bytes       dynamic LDS for the staged (plan) wrappers, else 0
 */
int fd_kernel_load(const char *hsaco_path, const char *symbol, fd_kernel_t *out);
/* JIT inside the library: compile wrapper source (what codegen emits) with hipcc --genco into cache_dir, keyed by
 * (source, flags) like compilation.load()/make_so() (pyop2/compilation.py:424-455, 527-611), then load `symbol`.
 * FDHIP_HIPCC / FDHIP_ARCH select the compiler / target; extra_flags may be NULL. */
int fd_kernel_create(const char *wrapper_src, const char *symbol, const char *cache_dir, const char *extra_flags,
                     fd_kernel_t *out);
int fd_kernel_builtin(const char *symbol, fd_kernel_t *out);
int fd_kernel_free(fd_kernel_t k);
int fd_kernel_launch(fd_kernel_t k, int32_t start, int32_t end,
                     const void *const *args, int nargs,
                     int block_threads, int ents_per_block, int nblocks,
                     size_t lds_bytes, fd_stream_t s);

/* ------------------------------------------------------ block-localisation plans
 * Backend-private lookup tables built once per (Map, iteration range), the way the
 * reference builds and caches maps / sparsities once per function space
 * (firedrake/functionspacedata.py:497-520).  The iteration range [start,end) is cut
 * into blocks of `ents_per_block` consecutive entities; per block the plan holds the
 * sorted list of distinct target nodes and, per (entity, i), the uint16 position of
 * map[entity][i] in that list.  The staged wrappers gather Dat rows once per block
 * into LDS, run the local kernel out of LDS, reduce INC contributions in LDS and
 * issue one global atomic per distinct node (DatPack semantics of
 * pyop2/codegen/builder.py:338-429 at block granularity).
 * Negative map entries (VALUE_UNDEFINED, pyop2/types/map.py:33) are not supported by
 * plans; the direct wrappers handle them.
 */
int fd_plan_create(const int32_t *map_dev, int arity, int32_t start, int32_t end,
                   int ents_per_block, fd_stream_t s, fd_plan_t *out);
/* Same, with caller-chosen block boundaries (host array of nblocks+1 entity offsets, e.g. the mesh
 * generator's traversal tiles): block b covers [block_starts[b], block_starts[b+1]). */
int fd_plan_create_blocks(const int32_t *map_dev, int arity, const int32_t *block_starts_host,
                          int32_t nblocks, fd_stream_t s, fd_plan_t *out);
/* Re-store the plan's local-map rows in LANE ORDER for workgroups of `lane_threads` lanes: a block of n
 * entities is cut into lane_threads contiguous runs (the first n %% lane_threads one longer) and the k-th entity
 * of run t goes to slot k*lane_threads + t.  A wrapper that walks the slots with stride lane_threads then has
 * the lanes of one trip on entities ~n/lane_threads apart (entities adjacent in a locality-preserving numbering
 * share nodes, and lanes adding into one LDS accumulator in the same ds_add are serialised), with coalesced
 * index rows.  Execution order inside a block is free for INC semantics (pyop2/codegen/builder.py:338-429 only
 * fixes what is accumulated, not when).  Call once, before fd_matplan_create. */
int fd_plan_set_lane_order(fd_plan_t p, int lane_threads, fd_stream_t s);
int fd_plan_block_starts(fd_plan_t p, const int32_t **block_starts_dev, int32_t *max_ents_per_block);
int fd_plan_info(fd_plan_t p, int32_t *nblocks, int32_t *max_nodes_per_block, int64_t *list_len);
int fd_plan_arrays(fd_plan_t p, const int32_t **block_offsets, const int32_t **node_list,
                   const uint16_t **local_map);
int fd_plan_free(fd_plan_t p);

/* Matrix plans: for every plan block, the distinct (row node, col node) pairs its entities touch, in
 * CSR order, with their position in the global node CSR.  Lets the staged wrapper reduce the block's
 * element matrices in LDS and issue ONE global atomic per distinct nonzero of the block -- the
 * block-granular form of MatSetValuesLocal(..., ADD_VALUES) (pyop2/codegen/builder.py:573-625).
 *   mb_off[nblocks+1]  offsets of the blocks' nonzero lists;   gpos[total]  global CSR positions
 *   lrp                local row pointers, block b at lrp[blkoff_r[b] + b .. + ndr_b]
 *   kidx               per (entity,i,j): offset inside its local row (uint8 if kbytes==1 else uint16) */
typedef struct fd_matplan_s *fd_matplan_t;
int fd_matplan_create(fd_plan_t row_plan, fd_plan_t col_plan, const fd_nnz_t *node_rowptr_dev,
                      const int32_t *node_colidx_dev, int64_t nnz, fd_stream_t s, fd_matplan_t *out);
/* gpos entries that exactly one block touches are stored as ~pos (negative): the wrapper writes them
 * without an atomic.  zero_list = all other CSR positions (shared between blocks, or never touched by
 * this loop): with a pending Mat.zero() (mat.py:851-855) only those need clearing before the loop --
 * exclusive entries are overwritten -- which fuses the zeroing pass into assembly. */
int fd_matplan_zero_list(fd_matplan_t m, const int32_t **zero_list, int64_t *n_zero, int64_t *n_exclusive);
int fd_matplan_info(fd_matplan_t m, int32_t *max_nnz_per_block, int32_t *max_row_len, int32_t *kbytes,
                    int64_t *total);
int fd_matplan_arrays(fd_matplan_t m, const int32_t **mb_off, const int32_t **gpos, const int32_t **lrp,
                      const void **kidx);
int fd_matplan_free(fd_matplan_t m);

/* Owner-computes-rows plans: matrix assembly without global atomics.  Row nodes are cut into contiguous
 * blocks [row_block_starts[b], row_block_starts[b+1]); block b visits every entity of [start,end) that touches
 * one of its rows ("instances"; border entities are visited by several blocks) and so produces COMPLETE rows,
 * which are contiguous in the CSR value array and are written with plain coalesced stores -- no atomics, and a
 * pending Mat.zero() costs nothing.  Block-granular form of SURVEY.md 8e option 1; replaces
 * MatSetValuesLocal(ADD_VALUES) + MatAssemblyBegin/End (builder.py:573-625, mat.py:940-954).
 *   inst_off[nblocks+1], inst_entity[ninst]: the instances of every block (device; inst_off also on the host)
 * fd_gather_rows builds the per-instance copy of a Map (dst[k] = src[idx[k]]) that fd_plan_create_blocks
 * then localises; fd_csr_elem_row_offsets gives, per entity (i,j), the offset of column cmap[e][j] inside
 * CSR row rmap[e][i] (uint8/uint16; all-ones = entry absent). */
typedef struct fd_ocrplan_s *fd_ocrplan_t;
int fd_ocrplan_create(const int32_t *rmap_dev, int rarity, int32_t start, int32_t end,
                      const int32_t *row_block_starts_host, int32_t nblocks, int interleave,
                      fd_stream_t s, fd_ocrplan_t *out);   /* interleave == 1: stencil order (instances of a block sorted by
                                                             * ownership pattern -- a wavefront's 64 consecutive instances
                                                             * then skip the atomics of rows none of them owns --, by the
                                                             * signature of owned rows + node offsets, then by first owned
                                                             * row: conflict-free LDS atomics on structured pieces);
                                                             * interleave > 1: multiplicative permutation of every instance list;
                                                             * 0: entity order */
/* The per-(block, staged node) words of the whole-entity owner-computes-rows wrapper in PLAN order (blkoff / list of the row
 * map's block-localisation plan over the instances): bits 0..29 = 1 + offset of the node's row in the block accumulator (0 =
 * not owned by the block, or dropped by the row lgmap: MatSetValuesLocal semantics, builder.py:573-625), bit 31 = column dropped
 * by the column lgmap.  base_by_node = CSR row starts by node (by_offset 0: ownership = node in the block's row range) or the
 * accumulator starts by node of a row order (by_offset 1: ownership = offset inside the block's range; nodes >= npos are not
 * rows); start_by_pos[rblk[b]] = start of block b.  One table per pair of lgmaps (the reference swaps them per assemble). */
int fd_ocr_node_words(const int32_t *blkoff_dev, const int32_t *list_dev, int32_t nblocks, const int32_t *rblk_dev,
                      const fd_nnz_t *base_by_node_dev, const fd_nnz_t *start_by_pos_dev, int by_offset, int32_t npos,
                      const int32_t *row_lgmap_dev, const int32_t *col_lgmap_dev, uint32_t *out_dev, fd_stream_t s);
/* words[i] |= (position of the diagonal entry inside the CSR row of node list[i]) << 20 (bits 20..27): with one map on both
 * sides of the Mat the place of entry (i, i) of an element matrix belongs to the row NODE, so the per-instance records below
 * need not carry it (the diagonal is always allocated: pyop2/sparsity.pyx:198-203).  Fails when an owned row has no diagonal
 * entry among its first 256 columns. */
int fd_ocr_node_diag(const int32_t *list_dev, int64_t n, int32_t nrows, const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev,
                     uint32_t *words_dev, fd_stream_t s);
/* Bit-packed per-instance records of a whole-entity owner-computes-rows loop: `words` 32-bit words per instance hold, back to
 * back from bit 0, the local-map rows of the nmaps staged maps (uint16 tables of the node plans, arities[m] entries of
 * lbits[m] bits) and the nr x nc row-offset entries (kidx, kbytes 1|2 per entry, kbits bits each; the nr diagonal entries
 * left out when skipdiag).  The wrapper streams ONE record per instance (fdw::load_rec / rec_field) instead of a uint16 row per
 * map plus a uint8 row of offsets: P1 tetrahedra 24 -> 12 bytes (the reference has no such tables: MatSetValuesLocal searches
 * every row on every call, builder.py:573-625).  Row-sliced loops (nr = 1) append one uint16 per instance -- the accumulator slot
 * of the instance's row -- as a last field of ebits bits (extra_dev), and mark dropped rows / columns with the all-ones value of
 * the source type: `sentinel` maps it to the all-ones value of the field.  Fails when an index does not fit its field. */
int fd_ocr_pack_records(int64_t ninst, int nmaps, const uint16_t *const *lmaps_dev, const int32_t *arities, const int32_t *lbits,
                        const void *kidx_dev, int kbytes, int nr, int nc, int kbits, int skipdiag, const uint16_t *extra_dev, int ebits,
                        int sentinel, int words, uint32_t *out_dev, fd_stream_t s);
/* The records of a paired row-sliced plan (fd_ocrplan_create_paired): kidx holds nr rows of nc positions per instance and
 * extra_dev nr slots per instance (all-ones source values = dropped, mapped to the all-ones value of the field). */
int fd_ocr_pack_records_rows(int64_t ninst, int nmaps, const uint16_t *const *lmaps_dev, const int32_t *arities, const int32_t *lbits,
                             const void *kidx_dev, int kbytes, int nr, int nc, int kbits, const uint16_t *extra_dev, int ebits,
                             int words, uint32_t *out_dev, fd_stream_t s);
/* Row blocks as ranges of row POSITIONS under a backend-derived row order (fd_first_touch_order): pinv[node] = position
 * for node < npos, prowptr[p] = CSR row start of the p-th row in that order (npos + 1 entries).  The CSR itself keeps
 * the caller's numbering; a block's rows are then a set of CSR rows, flushed row by row.  The tables are borrowed. */
int fd_ocrplan_create_ordered(const int32_t *rmap_dev, int rarity, int32_t start, int32_t end,
                              const int32_t *pos_block_starts_host, int32_t nblocks, int interleave,
                              const int32_t *pinv_dev, int32_t npos, const fd_nnz_t *prowptr_dev,
                              fd_stream_t s, fd_ocrplan_t *out);
int fd_ocrplan_info(fd_ocrplan_t p, int64_t *ninst, int32_t *max_inst_per_block);
int fd_ocrplan_arrays(fd_ocrplan_t p, const int32_t **inst_off_dev, const int32_t **inst_off_host,
                      const int32_t **inst_entity_dev, const int32_t **row_block_starts_dev);
int fd_ocrplan_free(fd_ocrplan_t p);
/* Row-sliced owner-computes-rows plans (large element matrices): an instance is (entity, local row i) for every row-map
 * entry that falls into a row block -- the wrapper instantiates the local kernel once per i and keeps row i of its
 * output, so no instance is redundant whatever the block size.  The instance list of a block is grouped by i and each
 * group is padded to a multiple of 64 slots (one wavefront = one i): fd_ocrplan_info reports the PADDED count,
 * fd_ocrplan_arrays the padded offsets / entities (padding repeats a real entity), fd_ocrplan_sliced_arrays the local
 * row index of every 64-slot chunk, the per-slot validity bytes and the number of real instances.  pinv_dev (nullable)
 * = row order as in fd_ocrplan_create_ordered.  fd_ocrplan_sliced_tables fills, for ONE pair of lgmaps (nullable), the
 * per-instance accumulator slot of the row (0xffff = padding or row dropped: negative map entry / negative row lgmap,
 * MatSetValuesLocal semantics, builder.py:573-625) and the positions of the entity's columns inside that CSR row
 * (all-ones = dropped); acc_by_node[r] - acc_by_pos[block start] is the row's offset in the block accumulator
 * (node_rowptr twice for the caller's row order; nstart / prowptr of a RowOrder otherwise). */
int fd_ocrplan_create_sliced(const int32_t *rmap_dev, int rarity, int32_t start, int32_t end,
                             const int32_t *block_starts_host, int32_t nblocks, const int32_t *pinv_dev, int32_t npos,
                             int interleave, fd_stream_t s, fd_ocrplan_t *out);   /* interleave > 1: the instances of a group
                             * in the order j -> (j*P) mod count, P = smallest integer >= interleave coprime with count */
/* Two rows per instance: the local rows are partitioned into ngroups GROUPS of one or two rows (group_rows_host[2g], [2g+1];
 * 255 = no second row) and an instance is (entity, group) -- the wrapper evaluates the local kernel ONCE per instance and keeps
 * both rows of its output, so the geometry and the instance's index record are shared by the two rows.  The block that owns the
 * group's first row gets the instance; when another block owns the second row it gets an instance of its own, and each of the
 * two drops the row it does not own (its slot reads 0xffff).  The chunk role of fd_ocrplan_sliced_arrays is then the GROUP index,
 * and fd_ocrplan_sliced_tables writes two table rows per instance ((t, s) at t*2 + s: slot[2*ninst], kk[2*ninst*carity]; scalar
 * matrices with node lgmaps only).  fd_ocrplan_pair_counts gives the statistic the caller picks the groups from: counts_host[a*rarity
 * + b] (a < b) = entities whose local rows a and b fall into the same row block.  The reference has no counterpart: MatSetValuesLocal
 * is called with whole element matrices (builder.py:573-625). */
int fd_ocrplan_create_paired(const int32_t *rmap_dev, int rarity, int32_t start, int32_t end,
                             const int32_t *block_starts_host, int32_t nblocks, const int32_t *pinv_dev, int32_t npos,
                             int interleave, int ngroups, const uint8_t *group_rows_host, fd_stream_t s, fd_ocrplan_t *out);
int fd_ocrplan_pair_counts(const int32_t *rmap_dev, int rarity, int32_t start, int32_t end, const int32_t *block_starts_host,
                           int32_t nblocks, const int32_t *pinv_dev, int32_t npos, int64_t *counts_host, fd_stream_t s);
int fd_ocrplan_sliced_arrays(fd_ocrplan_t p, const uint8_t **chunk_role_dev, const uint8_t **valid_dev, int64_t *nreal);
int fd_ocrplan_sliced_tables(fd_ocrplan_t p, const int32_t *rmap_dev, const int32_t *cmap_dev, int carity,
                             const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev, const fd_nnz_t *acc_by_node_dev,
                             const fd_nnz_t *acc_by_pos_dev, const int32_t *row_lgmap_dev, const int32_t *col_lgmap_dev,
                             int kbytes, uint16_t *slot_out_dev, uint16_t *rowlen_out_dev /* nullable: entries of the
                             instance's CSR node row, needed to address vector-valued blocks */, void *kk_out_dev,
                             int rbs, int cbs, uint8_t *rowmask_out_dev, uint64_t *colmask_out_dev /* both nullable; given:
                             the lgmaps are per DOF (node*bs + component; MatSetValuesLocal on dof indices, mat.py:700-716):
                             bit p of rowmask / bit j*cbs+q of colmask = that scalar row / column survives */, fd_stream_t s);
int fd_gather_rows(const int32_t *src_dev, int arity, const int32_t *idx_dev, int64_t n, int32_t *dst_dev, fd_stream_t s);
int fd_csr_elem_row_offsets(const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev, const int32_t *rmap_dev,
                            const int32_t *cmap_dev, int32_t nent, int rarity, int carity, int kbytes,
                            void *out_dev, fd_stream_t s);

/* ------------------------------------------------------------ sparsity / CSR (a12)
 * Native replacement of pyop2/sparsity.pyx:105-159 (build_sparsity) + :162-389
 * (fill_with_zeros): union over (rowmap, colmap) pairs of the outer product of each
 * map row, plus the diagonal of square blocks (sparsity.pyx:198-203).  Runs on the
 * device.  Extruded pairs give nlayers[k] > 0 and host arrays of per-entry offsets
 * (node = map + offset*layer, builder.py:94-124).  Output arrays are device memory
 * owned by the caller (release with fd_free).
 */
int fd_csr_from_maps(int32_t nrow_nodes, int32_t ncol_nodes, int set_diag, int npairs,
                     const int32_t *const *rmaps_dev, const int32_t *const *cmaps_dev,
                     const int32_t *nent, const int32_t *rarity, const int32_t *carity,
                     const int32_t *nlayers, const int32_t *const *roffsets_host,
                     const int32_t *const *coffsets_host,
                     fd_nnz_t **rowptr_dev, int32_t **colidx_dev, int64_t *nnz, fd_stream_t s);
/* The same with the rest of fill_with_zeros' extruded cases: `region[k]` is the iteration region of pair k
 * (FD_ON_BOTTOM / FD_ON_TOP / FD_ON_INTERIOR_FACETS / FD_ALL, the values of pyop2's IterationRegion; sparsity.pyx:291-305,
 * 331-346 -- interior facets couple two stacked cells), `periodic[k]` marks periodic extrusion (sparsity.pyx:273, 343-346)
 * and `rquot_host[k]` / `cquot_host[k]` are the maps' offset quotients (sparsity.pyx:309-312, 357-368):
 *   node = map[e][i % arity] + offset[i % arity] * ((layer + i / arity + quot) % nlayers - quot % nlayers).
 * region, periodic and the quotient arrays (or single quotients) may be NULL = FD_ALL / not periodic / 0.
 * Variable layers (pyop2/types/set.py:326-337, sparsity.pyx:325-330): `layers_dev[k]` is the device (nent, 2) array of
 * per-entity [bottom, top) node levels and nlayers[k] the largest number of cell layers of any entity; NULL (array or
 * entry) = constant layers. */
#define FD_ON_BOTTOM 1
#define FD_ON_TOP 2
#define FD_ON_INTERIOR_FACETS 3
#define FD_ALL 4
int fd_csr_from_maps_ex(int32_t nrow_nodes, int32_t ncol_nodes, int set_diag, int npairs,
                        const int32_t *const *rmaps_dev, const int32_t *const *cmaps_dev,
                        const int32_t *nent, const int32_t *rarity, const int32_t *carity,
                        const int32_t *nlayers, const int32_t *const *roffsets_host,
                        const int32_t *const *coffsets_host, const int32_t *region, const int32_t *periodic,
                        const int32_t *const *rquot_host, const int32_t *const *cquot_host,
                        const int32_t *const *layers_dev,
                        fd_nnz_t **rowptr_dev, int32_t **colidx_dev, int64_t *nnz, fd_stream_t s);
/* node pattern -> scalar (aij) pattern for DataSet dims (rbs, cbs) (mat.py:254-278) */
int fd_csr_expand_blocks(int32_t nrow_nodes, const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev,
                         int rbs, int cbs, fd_nnz_t **rowptr_out, int32_t **colidx_out, fd_stream_t s);
/* element -> nonzero table: out[((e*nl+l)*ar+i)*ac+j] = position of (rmap[e][i]+roff[i]*l,
 * cmap[e][j]+coff[j]*l) in the node CSR, or -1 (nlayers = 0: non-extruded, nl = 1).  Replaces the per-call row search inside MatSetValuesLocal
 * (pyop2/codegen/builder.py:573-625). */
int fd_csr_elem_offsets(const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev,
                        const int32_t *rmap_dev, const int32_t *cmap_dev,
                        int32_t nent, int rarity, int carity,
                        int nlayers, const int32_t *roffsets_host, const int32_t *coffsets_host,
                        int32_t *out_dev, fd_stream_t s);
/* Mat.set_local_diagonal_entries (mat.py:896-937) and Mat.zero_rows (mat.py:857-891) */
int fd_csr_set_diagonal(const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev, double *vals_dev,
                        const int32_t *rows_dev, int32_t nrows_sel, double value, fd_stream_t s);
int fd_csr_zero_rows(const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev, double *vals_dev,
                     const int32_t *rows_dev, int32_t nrows_sel, double value, fd_stream_t s);
/* The same diagonal fix-up for a row list that is applied at every assembly (Dirichlet rows of a Newton loop: the reference
 * calls set_local_diagonal_entries after every assemble(J), firedrake/assemble.py -> mat.py:896-937, and PETSc searches each row
 * each time): the places of the diagonal entries are found once (-1: negative row id or no diagonal entry in the pattern) and
 * fd_csr_set_at stores through them. */
int fd_csr_diag_positions(const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev, const int32_t *rows_dev, int32_t nrows_sel,
                          fd_nnz_t *pos_out_dev, fd_stream_t s);
int fd_csr_set_at(double *vals_dev, const fd_nnz_t *pos_dev, int32_t n, double value, fd_stream_t s);
/* y = A x  (parity identity  A*x == action(a, x), tests/firedrake/regression/test_matrix_free.py:97-123) */
int fd_csr_spmv(int32_t nrows, const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev,
                const double *vals_dev, const double *x_dev, double *y_dev, fd_stream_t s);
/* diag[r] = A[r][r] (0 where the pattern has no diagonal entry): MatGetDiagonal, what a Jacobi-preconditioned Krylov solve
 * of the assembled operator needs (the reference's regression tests solve with PETSc: tests/firedrake/regression/
 * test_helmholtz.py:50; here only the parity tests do, tests/test_reference_thresholds.py) */
int fd_csr_get_diagonal(int32_t nrows, const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev,
                        const double *vals_dev, double *diag_dev, fd_stream_t s);

/* ------------------------------------------------- hand-over to a distributed PETSc matrix (f1)
 * The owned rows [0, nrows_owned) of the local CSR split into the two sequential blocks an MPIAIJ matrix is made of
 * (pyop2/types/mat.py:254-278: Sparsity.nnz = d_nnz, Sparsity.onnz = o_nnz; firedrake/preconditioners/offload.py:25-131 moves
 * such a matrix to the device): columns < ncols_owned form the DIAGONAL block (local column indices), the others the
 * OFF-DIAGONAL block with GLOBAL column indices (col_global[local column], NULL = keep local) -- exactly the six arrays of
 * MatCreateMPIAIJWithSplitArrays(comm, m, n, M, N, i, j, a, oi, oj, oa).  Columns are sorted inside a row and owned
 * columns come first, so the diagonal block is a prefix of every row.  The off-diagonal block is a sequential AIJ matrix of
 * its own and needs its rows sorted by (global) column, which the local ghost numbering does not guarantee: o_rank[entry] =
 * place of the suffix entry inside its sorted row.  fd_csr_split_mpiaij builds the two patterns and the ranks once (outputs
 * allocated here, release with fd_free); fd_csr_split_values refreshes the two value arrays after an assembly (o_rank_dev
 * NULL = suffix order). */
int fd_csr_split_mpiaij(int32_t nrows_owned, const fd_nnz_t *rowptr_dev, const int32_t *colidx_dev, int32_t ncols_owned,
                        const int32_t *col_global_dev, int32_t **d_rowptr_dev, int32_t **d_colidx_dev, int64_t *d_nnz,
                        int32_t **o_rowptr_dev, int32_t **o_colidx_dev, int32_t **o_rank_dev, int64_t *o_nnz, fd_stream_t s);
int fd_csr_split_values(int32_t nrows_owned, const fd_nnz_t *rowptr_dev, const double *vals_dev, const int32_t *d_rowptr_dev,
                        const int32_t *o_rowptr_dev, const int32_t *o_rank_dev, double *d_vals_dev, double *o_vals_dev, fd_stream_t s);

/* ------------------------------------------------- backend-derived locality orders
 * The reference's locality comes from DMPlex (RCM cell order + first-touch DoF numbering, firedrake/mesh.py:1214-1228,
 * firedrake/cython/dmcommon.pyx:2599-2729); execution order is free under the wrapper's semantics
 * (pyop2/codegen/builder.py:734-741), so when the producer gives no block hints the backend derives its own:
 *   fd_kd_order           k-d partition of n points (pdim doubles each) into ceil(n / leaf_size) leaves of equal
 *                         population (+-1): recursive splits at the population median, in leaf units, along the longest
 *                         axis of each segment's bounding box.  order[p] = base + index of the p-th point, leaves
 *                         contiguous, points in index order inside a leaf; leaf_starts_host[0 .. *nleaves] (host array of capacity max_leaves + 1) = their
 *                         boundaries.  Leaves are boxes with the same number of points whatever the mesh grading: one
 *                         leaf of entities = one staged block, one leaf of rows = one owner-computes-rows block
 *   fd_group_entities     entities of [start, end) grouped by the smallest label among their nodes (label_dev[node] in
 *                         [0, nlabels): the k-d leaf of a node of the position field), groups in label order, entity order
 *                         inside a group; counts_host[l] = entities of group l.  One group = one staged block: the
 *                         entities around a leaf of nodes, as a producer's mesh tiles hold them
 *   fd_first_touch_order  the reference's first-touch rule applied to an entity order: plist[p] = p-th node of
 *                         [0, nnodes), pinv[node] = p; nodes no entity touches come last.  rank_dev (optional):
 *                         rank_dev[p] = position in `order` of the entity that first touches plist[p] (-1: untouched)
 *   fd_invert_permutation pinv[plist[p]] = p
 * Private re-encodings: Dats, Maps and the CSR keep the caller's numbering. */
int fd_kd_order(const double *pts_dev, int pdim, int64_t n, int32_t base, int32_t leaf_size, int32_t *order_dev,
                int32_t *leaf_starts_host, int32_t max_leaves, int32_t *nleaves_out, fd_stream_t s);
/* label_dev[order_dev[i]] = l for leaf_starts_host[l] <= i < leaf_starts_host[l + 1]: the node labels of fd_kd_order's leaves */
int fd_leaf_labels(const int32_t *order_dev, int64_t n, const int32_t *leaf_starts_host, int32_t nleaves, int32_t *label_dev, fd_stream_t s);
int fd_group_entities(const int32_t *map_dev, int arity, int32_t start, int32_t end, const int32_t *label_dev, int32_t nnodes,
                      int32_t nlabels, int32_t *order_dev, int32_t *counts_host, fd_stream_t s);
int fd_first_touch_order(const int32_t *map_dev, int arity, const int32_t *order_dev, int64_t n, int32_t nnodes,
                         int32_t *pinv_dev, int32_t *plist_dev, int32_t *rank_dev, fd_stream_t s);
int fd_invert_permutation(const int32_t *plist_dev, int32_t n, int32_t *pinv_dev, fd_stream_t s);
/* The tables of a row order plist (row of every position): prowptr[p] = accumulator start of position p (npos + 1 entries: the
 * running sum of the row lengths in position order), gstart[p] = CSR start of that row, nstart[row] = prowptr[position of row]
 * -- the lookups the "ocrp" / "ocrsp" wrappers and the plan builders need, each a flat array */
int fd_row_order_tables(int32_t npos, const int32_t *plist_dev, const fd_nnz_t *rowptr_dev, fd_nnz_t *prowptr_dev, fd_nnz_t *nstart_dev,
                        fd_nnz_t *gstart_dev, fd_stream_t s);
/* gpos[prowptr[p] + k] = gstart[p] + k for the rows p of a row order (prowptr = accumulator starts by position, gstart = CSR
 * starts by position): the place of every accumulator entry in the CSR value array, streamed by the "ocrp" row flush */
int fd_row_entry_positions(int32_t npos, const fd_nnz_t *prowptr_dev, const fd_nnz_t *gstart_dev, int32_t *gpos_dev, fd_stream_t s);
/* The same places with a COLUMN lgmap folded in (what the reference's MatSetValuesLocal does with negative column indices,
 * pyop2/codegen/builder.py:573-625, pyop2/parloop.py:279-302): an entry in a masked column reads -2 - place.  The whole-entity
 * wrapper over a derived row order ("ocrpm") then adds every contribution unconditionally in LDS and lets the flush skip those
 * entries (accumulating) or store 0.0 there (overwriting its rows): no select per contribution in the main loop. */
int fd_row_entry_positions_masked(int32_t npos, const fd_nnz_t *prowptr_dev, const fd_nnz_t *gstart_dev, const int32_t *colidx_dev,
                                  const int32_t *col_lgmap_dev, int32_t *gpos_dev, fd_stream_t s);
/* The same places run-coded: rows that follow one another in a block of the row order (rblk: nblocks + 1 block starts in row
 * positions) AND in the CSR share one displacement (place - accumulator index).  grun[entry] = run of the entry's row counted
 * from its block's first run (one byte), brun[b] = first run of block b (nblocks + 1), rdelta[run] = displacement (room for
 * npos).  *max_runs_out = most runs in one block: the "ocrspr" flush keeps a block's displacements in LDS and needs <= 256. */
int fd_ocr_row_runs(int32_t npos, const fd_nnz_t *prowptr_dev, const fd_nnz_t *gstart_dev, const int32_t *rblk_dev, int32_t nblocks,
                    uint8_t *grun_dev, int32_t *brun_dev, fd_nnz_t *rdelta_dev, int32_t *nruns_out, int32_t *max_runs_out, fd_stream_t s);

/* ------------------------------------------------- halo exchange + Global reductions over RCCL
 * firedrake/halo.py:87-172 (PetscSF bcast owner->ghost with MPI.REPLACE, reduce ghost->owner with SUM/MIN/MAX behind
 * pyop2/types/halo.py:4-56) and pyop2/parloop.py:411-442 (MPI_Iallreduce of INC/MIN/MAX Globals), one process per
 * GPU.  A communicator wraps ncclCommInitRank (librccl.so is bound at run time); rank 0 obtains the 128-byte id with
 * fd_comm_unique_id and the host side broadcasts it.  A halo holds the per-neighbour node lists (local numbering,
 * matching order on both sides: send[k] = owned nodes neighbour k keeps as ghosts, recv[k] = our ghosts it owns),
 * persistent packed buffers and the side stream the grouped ncclSend/ncclRecv exchange runs on:
 *     g2l (forward)  owner -> ghost, REPLACE       pyop2/types/dat.py:622-658  (global_to_local_begin/end)
 *     l2g (reverse)  ghost -> owner, op            pyop2/types/dat.py:659-678  (local_to_global_begin/end)
 * *_begin queues pack + transfer and returns; *_end makes the compute stream wait for the transfer and queues the
 * unpack, so whatever is launched on `s` in between overlaps the exchange (parloop.py:250-253).  Several Dats may be
 * in flight at once (one begin/end pair per Dat and direction).
 *   dtype: FD_F64 .. FD_U64;  op: 0 = REPLACE (forward only), 1 = SUM, 2 = MIN, 3 = MAX.
 * A halo created with comm == NULL packs/unpacks only: the caller moves the packed rows itself between begin and end
 * (fd_halo_wire_buffers) -- the host-bounce wire of the gloo-launched tests. */
enum { FD_F64 = 0, FD_F32 = 1, FD_I32 = 2, FD_U32 = 3, FD_I64 = 4, FD_U64 = 5 };
typedef struct fd_comm_s *fd_comm_t;
typedef struct fd_halo_s *fd_halo_t;
int fd_comm_available(void);                          /* 1 if librccl.so could be bound, else 0 (reason: fd_last_error) */
int fd_comm_unique_id(unsigned char *id128);
int fd_comm_create(const unsigned char *id128, int rank, int nranks, fd_comm_t *out);   /* collective over all ranks */
int fd_comm_info(fd_comm_t c, int *rank, int *nranks);
int fd_comm_free(fd_comm_t c);
int fd_comm_allreduce(fd_comm_t c, void *buf_dev, int64_t count, int dtype, int op, fd_stream_t s);   /* in place */
int fd_halo_create(fd_comm_t comm, int nneigh, const int32_t *peer_ranks,
                   const int32_t *const *send_idx_host, const int32_t *nsend,
                   const int32_t *const *recv_idx_host, const int32_t *nrecv, fd_halo_t *out);
int fd_halo_free(fd_halo_t h);
int fd_halo_g2l_begin(fd_halo_t h, void *dat_dev, int cdim, int dtype, fd_stream_t s);
int fd_halo_g2l_end(fd_halo_t h, void *dat_dev, int cdim, int dtype, fd_stream_t s);
int fd_halo_l2g_begin(fd_halo_t h, void *dat_dev, int cdim, int dtype, int op, fd_stream_t s);
int fd_halo_l2g_end(fd_halo_t h, void *dat_dev, int cdim, int dtype, int op, fd_stream_t s);
int fd_halo_wire_buffers(fd_halo_t h, const void *dat_dev, int dir, void **send_buf, int64_t *send_rows,
                         void **recv_buf, int64_t *recv_rows);   /* dir: 0 = g2l, 1 = l2g */
/* identity of an access mode over a contiguous element range: kind 0 = zero (INC), 1 = largest (MIN), 2 = lowest (MAX)
 * value of the dtype -- the ghost-region fill of pyop2/types/dat.py:631-636 for any Dat dtype */
int fd_dat_fill_range(void *dat_dev, int64_t first_elem, int64_t count, int dtype, int kind, fd_stream_t s);

/* ------------------------------------------------- pointwise helpers (a13, a14)
 * bc.zero / bc.apply on a residual (firedrake/bcs.py:192-221, 404-457). */
int fd_dat_set_rows(double *dat_dev, int cdim, const int32_t *rows_dev, int32_t n,
                    double value, fd_stream_t s);
/* y = a*x + b*y over n doubles: Dat.assign / axpy kept on the device between assemblies
 * (pyop2/types/dat.py:354-620; firedrake/assign.py), e.g. the Runge-Kutta stages of the DG-advection demo. */
int fd_dat_axpby(double *y_dev, double a, const double *x_dev, double b, int64_t n, fd_stream_t s);
int fd_dat_copy_rows(double *dst_dev, const double *src_dev, int cdim, const int32_t *rows_dev,
                     int32_t n, fd_stream_t s);

#ifdef __cplusplus
}
#endif
#endif /* FDHIP_H */
