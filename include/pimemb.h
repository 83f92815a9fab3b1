/*
 * pimemb.h -- C ABI of the MI355X-native embedding-lookup engine (libpimemb.so).
 *
 * Drop-in boundary for the ONE hot path of UBC-ECE-Sasha/PIM-Embedding-Lookup: the multi-table
 * EmbeddingBag(sum) lookup that the reference offloads to UPMEM DPUs behind `populate_mram` /
 * `lookup` (upmem/include/emb_host.h:136, :234; built into emblib.so by upmem/Makefile:122-124).
 * Plain pointers and sizes only: no torch, no C++ types.  All functions are `extern "C"`.
 *
 * Two groups of entry points:
 *   (1) the runtime-shaped native API `emb_*` -- shapes are arguments, errors are return codes
 *       (the reference bakes NR_TABLES/NR_COLS/MAX_NR_BATCHES/MAX_INDICES_PER_BATCH in with -D,
 *       upmem/Makefile:69-81, and exit()s on error, emb_host.h:87-90,157);
 *   (2) the two reference signatures, byte-for-byte, so existing callers of emblib.so relink
 *       unchanged: `populate_mram`, `lookup` (+ `emb_configure` replacing the -D macros).
 *
 * Threading.  The reference assumes a single caller thread (unguarded globals, emb_host.h:32-33).
 * Here: table management (emb_alloc_table / emb_load_* / emb_set_hot_rows / emb_destroy) must not run
 * concurrently with lookups on the same engine.  Lookups may be issued from several threads:
 * host-pointer calls are serialised inside the engine (they share one staging buffer),
 * device-pointer calls and emb_plan_launch only enqueue work on the caller's stream (plan-less calls draw their launch
 * images from a pool of eight rings picked by the calling thread: threads do not queue up behind one engine mutex).  populate_mram /
 * lookup serialise themselves.  emb_last_error() is thread-local.
 *
 * Paths cited as upmem/... are relative to the reference checkout.
 */
#ifndef PIMEMB_H
#define PIMEMB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------------------------------ */
/* status codes (the reference has none: DPU_ASSERT aborts, emb_host.h:157; lookup returns 0)  */
/* ------------------------------------------------------------------------------------------ */
#define EMB_OK 0
#define EMB_ERR_INVALID (-1)     /* bad argument (null pointer, unknown table, bad dtype/dim ...) */
#define EMB_ERR_NOMEM (-2)       /* host or HBM allocation failed (replaces enomem(), emb_host.h:87) */
#define EMB_ERR_DEVICE (-3)      /* HIP runtime error, text in emb_last_error() */
#define EMB_ERR_UNSUPPORTED (-4) /* e.g. dim 0, more tiles than one grid holds */
#define EMB_ERR_RANGE (-5)       /* emb_validate_inputs found an out-of-range index / bad offsets */

/* element type of a table as it sits in HBM; pooled output is always fp32 [bag][dim]
 * (emb_host.h:210 writes float final_results[k*NR_COLS + j]) */
typedef enum emb_dtype {
    EMB_F32 = 0,     /* fp32 rows, fp32 accumulate: nn.EmbeddingBag(sum) parity target */
    EMB_F16 = 1,     /* fp16 rows, fp32 accumulate (BASELINE config C5) */
    EMB_FIXED32 = 2  /* int32 x1e9 fixed point, int32 wrap-around accumulate, out = (float)acc/1e9:
                        the reference's own arithmetic, emb_dpu_lookup.c:114 + emb_host.h:210 */
} emb_dtype;

/* width of indices AND offsets: uint32 at the reference ABI (emb_host.h:234), int64 at torch's */
typedef enum emb_index_type { EMB_IDX_U32 = 0, EMB_IDX_I64 = 1 } emb_index_type;

/* where caller buffers live */
typedef enum emb_memspace {
    EMB_MEM_HOST = 0,   /* host pointers: the engine copies in/out (reference behaviour) */
    EMB_MEM_DEVICE = 1  /* HBM pointers on the engine's GPU: zero-copy, nothing leaves the device */
} emb_memspace;

typedef struct emb_engine emb_engine; /* opaque; replaces the global `struct dpu_set_t *dpu_set`,
                                         emb_host.h:33 */
typedef struct emb_plan emb_plan;     /* opaque; a prepared multi-table lookup (descriptors in HBM) */

typedef struct emb_config {
    int32_t device;      /* HIP device ordinal; -1 = current device */
    uint32_t max_tables; /* table ids are 0 .. max_tables-1 (NR_TABLES, upmem/Makefile:69-81) */
    uint32_t flags;      /* EMB_FLAG_* */
} emb_config;
#define EMB_FLAG_STAGE_TIMING 1u /* host-pointer calls wait after every stage and clock it (see emb_stats) */
#define EMB_FLAG_CHECK_INPUTS 2u /* emb_lookup / emb_lookup_batched (and lookup()) run emb_validate_inputs on the
                                    caller's stream first and return EMB_ERR_RANGE instead of launching when an index
                                    is >= nr_rows or the offsets are broken.  Off by default: the reference never
                                    checks (emb_dpu_lookup.c:113) and the check costs a kernel and a host wait per
                                    call.  Prepared plans (emb_plan_launch) are never checked. */
#define EMB_FLAG_DEFER_CHECK 4u  /* with EMB_FLAG_CHECK_INPUTS: the verdict of a checked DEVICE-pointer call is not waited for
                                    inside the call (that wait puts the host one launch behind the GPU on every call: 21 -> 41 us
                                    per 26-table call at 39 292 bags per table).  The validation kernel still disarms the call's
                                    lookup on the device when it finds something (outputs untouched); the finding is returned --
                                    EMB_ERR_RANGE, emb_last_error naming the call it belongs to -- by the first later checked call
                                    that finds it arrived -- of any thread, on any stream: verdicts are the engine's -- (that call's own work
                                    has been launched), or by emb_check_report.  Like a
                                    device-side assert: late, never lost (emb_destroy prints a verdict nobody read).
                                    emb_lookup_batched_checked stays synchronous. */

/*
 * One table's share of a batched lookup -- what emb_host.h:234 passes as indices[t], offsets[t],
 * final_results[t], with the lengths made explicit (the reference always sends the compile-time
 * maxima, emb_host.h:282-283).
 *   offsets[b] = first position of bag b in `indices`; bag b ends at offsets[b+1]; the LAST bag
 *   ends at n_indices (emb_dpu_lookup.c:109-110; torch include_last_offset=False).
 *   Empty bag -> zeros (emb_dpu_lookup.c:108).  Duplicate indices are summed each time.
 *   offsets == NULL with fixed_pooling = L > 0 means offsets[b] = b*L (load_generator.c:88).
 *   pooled: float[n_bags][dim], row-major (emb_host.h:210).
 */
typedef struct emb_lookup_desc {
    uint32_t table_id;
    uint32_t fixed_pooling;
    const void *indices; /* uint32[n_indices] or int64[n_indices] */
    const void *offsets; /* uint32[n_bags] or int64[n_bags] (same width as indices), or NULL */
    uint64_t n_indices;
    uint64_t n_bags;
    float *pooled;
} emb_lookup_desc;

/* counters + stage times; the six stage fields mirror dpu_runtime_totals (emb_host.h:41-48) and the
 * six latencies lookup() prints when latency_print == 1 (emb_host.h:395-402) */
typedef struct emb_stats {
    uint64_t n_lookup_calls;     /* emb_lookup + emb_lookup_batched + emb_plan_launch + lookup() */
    uint64_t n_kernel_launches;
    uint64_t n_bags;             /* pooled (table, bag) outputs produced */
    uint64_t n_indices;          /* rows gathered */
    uint64_t table_bytes;        /* HBM held by tables */
    /* host-pointer calls only.  With stage timing on (EMB_FLAG_STAGE_TIMING / emb_set_stage_timing /
     * emb_trace_enable / lookup(latency_print=1)) the host waits after each stage and clocks it;
     * otherwise a call is one enqueue chain with one wait and its whole duration goes to us_sync. */
    double us_copy_in_indices;   /* "Indices and offsets copying latency": pack + host->HBM copy */
    double us_copy_in_lengths;   /* "Query copying latency": descriptor upload + launch enqueue */
    double us_launch;            /* "Dpu launch latency": the fused kernel */
    double us_copy_out;          /* "Results copy latency": HBM->host copy enqueue */
    double us_post_process;      /* "Callback prep latency" -- 0: conversion is fused in the kernel */
    double us_sync;              /* "DPU sync latency": the final wait */
    /* kernel launches by kind: [0] wave-batch (one-hot, 64 bags per wavefront), [1] lane-group (pooled /
     * small), [2] two-batch wave-batch (very large one-hot), [3] any-dim (rows that are not 16-byte
     * multiples), [4] lane-group with hot rows in LDS (emb_set_hot_rows) */
    uint64_t n_launches_by_kind[5];
} emb_stats;

/* ------------------------------------------------------------------------------------------ */
/* (1) native API                                                                              */
/* ------------------------------------------------------------------------------------------ */

/* Thread-local text of the last error returned on this thread ("" if none). */
const char *emb_last_error(void);

/* Library / build identification, e.g. "pimemb 0.1 gfx950". */
const char *emb_version(void);

/* Create an engine on one GPU.  Replaces the first-call dpu_alloc + dpu_load (emb_host.h:155-160). */
int emb_create(const emb_config *cfg, emb_engine **out);

/* Free every table and workspace.  The reference has no destroy function (emb_host.h:33).
 * Fails with EMB_ERR_INVALID while prepared plans of this engine are alive. */
int emb_destroy(emb_engine *e);

/* Upload one whole table, row-major [nr_rows][dim] of `dtype`, into HBM (synchronous copy, like
 * DPU_XFER_DEFAULT at emb_host.h:173).  Replaces NR_COLS populate_mram calls and the dead
 * row->column split alloc_buffers (emb_host.h:101-122).  Re-loading a table id replaces it. */
int emb_load_table(emb_engine *e, uint32_t table_id, uint64_t nr_rows, uint32_t dim, emb_dtype dtype,
                   const void *rows, emb_memspace space);

/* Allocate a table without filling it (for emb_load_table_column / device-side initialisation). */
int emb_alloc_table(emb_engine *e, uint32_t table_id, uint64_t nr_rows, uint32_t dim, emb_dtype dtype);

/* Upload ONE column of an EMB_FIXED32 table -- populate_mram's unit of transfer
 * (emb_host.h:136,167-173: one int32[nr_rows] slice per (table, col)). */
int emb_load_table_column(emb_engine *e, uint32_t table_id, uint32_t col, const int32_t *column,
                          uint64_t nr_rows);

/* Optional hint: the table's hottest rows, hottest first (n_rows == 0 clears the set).  Pooled launches
 * (more than two indices per bag on average) over tables with a hot set run a kernel variant whose
 * persistent workgroups stage a compact copy of those rows plus a row-id hash into LDS (as many as fit
 * 62 KiB: ~100 rows of 512 B) and serve hits from there; misses and all other launches are unchanged,
 * and results are bit-identical with or without the hint.  Measured +3 % on dim-128 pooling-32
 * Zipf(1.2) lookups (the L2 already holds the head of the distribution).  The copy is taken now: set the
 * hint after the table contents are final -- (re)loading the table or a column drops it.  Prepared
 * plans that involve the table must be re-created.  No reference counterpart (the DPU program has no
 * cache, emb_dpu_lookup.c:113). */
int emb_set_hot_rows(emb_engine *e, uint32_t table_id, const uint64_t *row_ids, uint32_t n_rows);
/* The ENGINE picks the hot rows (emb_set_hot_rows needs the caller to know them): it counts the row ids of a sample of ONE
 * batch's index array -- DEVICE or HOST memory, uint32 or int64; up to four evenly spaced runs of 65 536 ids are copied to the
 * host and counted there: a cold-path call of a fraction of a millisecond, synchronous on `stream` -- and stages the max_rows
 * most frequent ones exactly as emb_set_hot_rows does (most frequent first, smaller id first among equals), unless they cover
 * less than min_share of the sample (near-uniform accesses gain nothing from an LDS copy: the table's hot set is cleared).
 * *n_chosen (may be NULL) = rows staged (those that fit the LDS budget), *share (may be NULL) = the share of the sample the
 * max_rows most frequent ids cover.  Call it again when the access pattern drifts; prepared plans over the table must be
 * re-created, as after emb_set_hot_rows.  The DPU program stages its working set by itself as well -- it copies the batch's
 * indices and offsets into WRAM before its loop (emb_dpu_lookup.c:41-58). */
int emb_learn_hot_rows(emb_engine *e, uint32_t table_id, const void *indices, uint64_t n_indices, emb_index_type itype,
                       emb_memspace space, uint32_t max_rows, float min_share, void *stream, uint32_t *n_chosen, float *share);

/* HBM address / shape of a loaded table (for zero-copy initialisation or inspection). */
int emb_table_info(emb_engine *e, uint32_t table_id, void **device_rows, uint64_t *nr_rows,
                   uint32_t *dim, emb_dtype *dtype);

/* The north star's lookup(table_id, offsets, indices) -> pooled_rows, one table.
 * `stream` is a hipStream_t (NULL = the default stream).  With EMB_MEM_HOST the call copies in,
 * runs and copies out synchronously; with EMB_MEM_DEVICE it only enqueues work on `stream`. */
int emb_lookup(emb_engine *e, uint32_t table_id, const void *indices, uint64_t n_indices,
               const void *offsets, uint64_t n_bags, float *pooled, emb_index_type itype,
               emb_memspace space, void *stream);

/* All tables in ONE fused kernel launch -- the counterpart of lookup() at emb_host.h:234, which
 * serves every table per call.  descs: host array of n_descs entries (any subset/order of tables,
 * a table may appear more than once). */
int emb_lookup_batched(emb_engine *e, const emb_lookup_desc *descs, uint32_t n_descs,
                       emb_index_type itype, emb_memspace space, void *stream);

/* One index per bag, served by ROW RANGE (DEVICE buffers, uint32 indices): for every descriptor, pooled[b] = the row
 * indices[b] - row_lo[i] of the table for the bags whose index falls into [row_lo[i], row_lo[i] + the table's rows); every
 * other bag of `pooled` is left untouched.  The building block of the sharded lookup's direct path for one-hot row-split
 * tables: every shard scans the requester's raw index array and stores only the rows it holds, straight into the
 * requester's output -- each bag has exactly one shard, so there is nothing to route and nothing to add up
 * (emb_dpu_lookup.c:113-114 is the gather being served; the reference's column shards likewise all read the same index list,
 * emb_host.h:258-263).  offsets must be NULL and fixed_pooling 1; rows are 16-byte multiples up to 1 KiB.  A WHOLE table is
 * the range row_lo = 0: whole tables and shards of one row width share ONE launch of the tuned one-hot kernel (an index
 * outside the table then leaves its bag untouched instead of reading out of bounds). */
int emb_lookup_ranged(emb_engine *e, const emb_lookup_desc *descs, const uint64_t *row_lo, uint32_t n_descs, void *stream);
/* The same as a prepared launch (emb_plan_launch / emb_plan_destroy as for any plan): for a sharded step whose buffers recur. */
int emb_plan_create_ranged(emb_engine *e, const emb_lookup_desc *descs, const uint64_t *row_lo, uint32_t n_descs,
                           emb_plan **out);
/* COUNTED ranged lookups: served[i] (DEVICE, may be NULL; the array itself may be NULL = emb_lookup_ranged) is a COUNTER
 * descriptor i's launch ADDS the number of bags it served to: EMB_SERVED_LANES uint32 words, EMB_SERVED_STRIDE bytes apart
 * (EMB_SERVED_BYTES in all; a wavefront adds to the lane of its workgroup -- atomics on one line would retire one by one),
 * the count is the sum of the lanes.  The caller zeroes and reads them in stream order.  Each bag of a row-split table has exactly one shard holding its row, so the requester adds the
 * counts of all shards up: a sum short of its bag count means an index that NO shard holds -- a bag left untouched, which a
 * checked shard (EMB_SHARD_CHECK_SERVED) reports as EMB_ERR_RANGE on the requesting rank.  The reference never checks
 * (emb_dpu_lookup.c:113: an out-of-range index is a wild MRAM read). */
#define EMB_SERVED_LANES 64u
#define EMB_SERVED_STRIDE 256u
#define EMB_SERVED_BYTES (EMB_SERVED_LANES * EMB_SERVED_STRIDE)
int emb_lookup_ranged_counted(emb_engine *e, const emb_lookup_desc *descs, const uint64_t *row_lo, uint32_t *const *served,
                              uint32_t n_descs, void *stream);
int emb_plan_create_ranged_counted(emb_engine *e, const emb_lookup_desc *descs, const uint64_t *row_lo,
                                   uint32_t *const *served, uint32_t n_descs, emb_plan **out);
/* The same for either index width (the entry points above take uint32, the reference's width; DLRM's tensors are int64: SURVEY
 * App. B.4): itype is the width of every descriptor's index array.  An int64 id is compared as it is -- negative ids and ids of
 * 2^32 and more fall into no range.  row_lo[i] may carry EMB_RANGE_OPEN_END: descriptor i then also answers for every id at or
 * beyond the END of its range (the last shard of a row-split table, the one holder of a whole table -- ids no rank holds, negative
 * int64 ids included): such a bag is not served and not counted, but its pooled row is written as ZEROS instead of being left
 * untouched, so that a caller who logs the error a checked shard reports and still consumes the outputs reads zero rows, never
 * stale bytes. */
#define EMB_RANGE_OPEN_END (1ull << 63)
int emb_lookup_ranged_typed(emb_engine *e, const emb_lookup_desc *descs, const uint64_t *row_lo, uint32_t *const *served,
                            uint32_t n_descs, emb_index_type itype, void *stream);
int emb_plan_create_ranged_typed(emb_engine *e, const emb_lookup_desc *descs, const uint64_t *row_lo, uint32_t *const *served,
                                 uint32_t n_descs, emb_index_type itype, emb_plan **out);

/* Prepared lookup over DEVICE buffers: descriptors are resolved and uploaded once, every launch is
 * then a single kernel enqueue (graph-capturable: no allocation, copy or sync inside).  The plan
 * holds the pointers it was given: buffers must stay allocated; re-allocating one of its tables
 * (emb_alloc_table / emb_load_table with another size) makes emb_plan_launch fail with
 * EMB_ERR_INVALID instead of reading freed memory. */
int emb_plan_create(emb_engine *e, const emb_lookup_desc *descs, uint32_t n_descs,
                    emb_index_type itype, emb_plan **out);
int emb_plan_launch(emb_plan *p, void *stream);
int emb_plan_destroy(emb_plan *p);
/* Algorithmic bytes one launch of the plan moves:
 * sum_t n_idx*(dim*elem + idx) + n_bags*off + n_bags*dim*4. */
int emb_plan_bytes(const emb_plan *p, uint64_t *algorithmic_bytes, uint64_t *n_bags,
                   uint64_t *n_indices);
/* A hash of what the plan's launches ARE -- kernel kind and row geometry, grid, the XCD-aware workgroup map word for word, index
 * width, per descriptor its tile / bag / index / row counts -- and of nothing that names a buffer.  Two plans with the same
 * signature enqueue the same grid of the same kernel over the same shapes, whatever host code built them: bench.py ties a
 * committed counter profile (profiles/traffic.json) to the launch it measured with this and with the kernel's own code bytes,
 * so that host-only edits of the library do not orphan a profile and a changed launch always does. */
int emb_plan_signature(const emb_plan *p, uint64_t *signature);
/* The plan's launches as text, one "key=value ..." record per kernel launch, ';' between them: kind (0 wave-batch, 1 lane-group,
 * 2 two-batch wave-batch, 3 any-dim, 4 hot rows), dtype, itype, lanes_per_row, chunks, scalar_lanes, anydim_vec, ranged, descs, grid
 * (workgroups), bags_per_tile -- enough to name the kernel INSTANTIATION each launch runs (bench.py finds its code in the library's
 * gfx950 code object with it: pim-embedding-lookup_amd/codeobj.py). */
int emb_plan_describe(const emb_plan *p, char *buf, size_t capacity);
/* Time `iters` back-to-back launches with HIP events on `stream` after `warmup` untimed ones;
 * *avg_us = mean device time per launch. */
int emb_plan_time(emb_plan *p, void *stream, uint32_t warmup, uint32_t iters, float *avg_us);

/* Debug-only input check (the reference never checks: an out-of-range index is a wild MRAM read,
 * emb_dpu_lookup.c:113).  Counts indices >= nr_rows and non-monotone / out-of-range offsets over
 * DEVICE or HOST buffers; returns EMB_ERR_RANGE if *n_bad > 0.  (*n_bad, here and below, counts up to 16 777 215 and stays there.) */
int emb_validate_inputs(emb_engine *e, const emb_lookup_desc *descs, uint32_t n_descs,
                        emb_index_type itype, emb_memspace space, uint64_t *n_bad);
/* The same on a stream of the caller's (emb_validate_inputs uses the default stream): the check is ordered behind
 * whatever produced the indices on `stream`.  One small kernel whose last workgroup reports into a pinned word the host polls; no event, no allocation, no device-wide
 * synchronize. */
int emb_validate_inputs_on(emb_engine *e, const emb_lookup_desc *descs, uint32_t n_descs,
                           emb_index_type itype, emb_memspace space, void *stream, uint64_t *n_bad);
/* emb_lookup_batched that checks first, whatever the engine's flags say: the descriptors are resolved once, validated
 * on `stream`, and launched only if clean -- otherwise EMB_ERR_RANGE, *n_bad (may be NULL) = the number of offending
 * values, nothing launched.  What nn.EmbeddingBag-shaped layers call for tensors they do not trust (IndexError). */
int emb_lookup_batched_checked(emb_engine *e, const emb_lookup_desc *descs, uint32_t n_descs,
                               emb_index_type itype, emb_memspace space, void *stream, uint64_t *n_bad);
/* emb_lookup_batched_checked with the verdict DEFERRED (see EMB_FLAG_DEFER_CHECK; DEVICE pointers -- a host-pointer call is synchronous anyway):
 * returns EMB_ERR_RANGE for an EARLIER deferred call whose verdict has arrived (this call's work is launched all the same). */
int emb_lookup_batched_checked_deferred(emb_engine *e, const emb_lookup_desc *descs, uint32_t n_descs,
                                        emb_index_type itype, emb_memspace space, void *stream);
/* Wait for the verdicts of every deferred checked call made so far: EMB_OK, or EMB_ERR_RANGE with *n_bad (may be NULL) offending
 * values in all and the first offending call named in emb_last_error. */
int emb_check_report(emb_engine *e, uint64_t *n_bad);

int emb_get_stats(emb_engine *e, emb_stats *out);
int emb_reset_stats(emb_engine *e);
/* Switch per-stage waits + clocks of host-pointer calls on/off (see emb_stats). */
int emb_set_stage_timing(emb_engine *e, int on);

/* Per-call stage intervals of the host-pointer path, for the reference's profiling workflow: the
 * interval CSV "DPU, Start, Stop" written by upmem/dputypes.py:87-98 and plotted by
 * graph/runtime_intervals/main.py, and the Chrome trace of upmem/test.json.  Clock: CLOCK_MONOTONIC
 * in microseconds (the reference's TIME_NOW, emb_host.h:35). */
#define EMB_STAGE_COPY_IN 0      /* indices + offsets host -> HBM        (emb_host.h:258-270) */
#define EMB_STAGE_DESCRIPTORS 1  /* descriptor / lengths upload          (emb_host.h:280-287) */
#define EMB_STAGE_LAUNCH 2       /* kernel enqueue .. completion         (emb_host.h:297)     */
#define EMB_STAGE_COPY_OUT 3     /* pooled rows HBM -> host              (emb_host.h:312-321) */
#define EMB_STAGE_SYNC 4         /* final wait                           (emb_host.h:350)     */
typedef struct emb_trace_event {
    uint32_t stage;   /* EMB_STAGE_* */
    uint32_t call_id; /* running number of the host-path lookup call */
    double start_us;
    double stop_us;
} emb_trace_event;
/* Keep the last `capacity` events (0 switches tracing off and drops what was recorded). */
int emb_trace_enable(emb_engine *e, uint32_t capacity);
/* Copy out up to max_events recorded events, oldest first, and remove them from the buffer. */
int emb_trace_read(emb_engine *e, emb_trace_event *out, uint32_t max_events, uint32_t *n_events);

/* Device-memory helpers so callers without a HIP binding (ctypes, cgo, JNI) can keep inputs and
 * outputs resident in HBM. */
int emb_device_alloc(emb_engine *e, size_t bytes, void **out);
int emb_device_free(emb_engine *e, void *ptr);
int emb_copy_to_device(emb_engine *e, void *dst_device, const void *src_host, size_t bytes);
int emb_copy_to_host(emb_engine *e, void *dst_host, const void *src_device, size_t bytes);
int emb_memset_device(emb_engine *e, void *dst_device, int value, size_t bytes);
int emb_synchronize(emb_engine *e, void *stream);
/* A HIP stream (non-blocking) for callers without a HIP binding: one per serving thread keeps their launches independent. */
int emb_stream_create(emb_engine *e, void **stream);
int emb_stream_destroy(emb_engine *e, void *stream);
int emb_device_of(emb_engine *e, int32_t *device);

/* ------------------------------------------------------------------------------------------ */
/* request queue: R pending SMALL lookups -> ONE launch                                         */
/* ------------------------------------------------------------------------------------------ */
/* The reference serves DLRM at mini-batch 1 (README.md:6) and 32 with MAX_NR_BATCHES = 512 (upmem/run.sh:40-45,119): a
 * 26-table lookup of a handful of bags.  On the GPU such a call is one ~3.5-us kernel launch whatever it gathers, so a
 * server that issues them back to back is launch-bound.  A queue collects requests as they arrive -- from any number of
 * threads, under the queue's own short lock: emb_queue_add resolves the request's descriptors on the spot and never takes
 * the engine's mutex -- and emb_queue_flush runs everything pending as ONE fused launch (each request keeps its own
 * buffers and gets exactly the bits a lookup of its own would give).
 *   EMB_MEM_DEVICE: the descriptors' pointers are used as they are; results are complete in stream order behind the flush.
 *   EMB_MEM_HOST  : emb_queue_add copies the request's indices / offsets into pinned, device-visible memory of the queue;
 *                   the kernel gathers from there and stores the pooled rows there (no copy engine); emb_queue_wait(ticket)
 *                   blocks until the flush that carried the request has finished and copies its rows into the caller's
 *                   buffers (every waiter copies its own request: threads unpack in parallel).
 * All tables of a queue share one (dtype, dim) -- every reference preset does (NR_COLS); other shapes: emb_lookup_batched.
 * A host-space request's rows stay in the queue's staging until emb_queue_wait has copied them out: the fourth flush after
 * its own blocks (up to 30 s, then fails) while any request of that flush is uncollected, so a front end that flushes faster
 * than its clients collect is held back instead of overwriting rows.  No event is
 * recorded per flush (an event between two kernels costs GPU time): a host queue's kernel is followed by a one-thread kernel that
 * stores the flush number into a pinned word its waiters poll; a device queue's results are simply complete in stream order
 * behind the flush (emb_queue_wait on a device queue synchronises the flush's stream). */
typedef struct emb_queue emb_queue;
int emb_queue_create(emb_engine *e, emb_index_type itype, emb_memspace space, emb_queue **out);
int emb_queue_add(emb_queue *q, const emb_lookup_desc *descs, uint32_t n_descs, uint64_t *ticket);
/* Several requests in one call (a front end that drained its socket): descs holds them back to back, request r has
 * n_descs[r] descriptors; tickets are consecutive from *first_ticket. */
int emb_queue_add_many(emb_queue *q, const emb_lookup_desc *descs, const uint32_t *n_descs, uint32_t n_requests,
                       uint64_t *first_ticket);
/* ONE launch over every request added since the last flush (*n_requests: how many; 0 = nothing was pending). */
int emb_queue_flush(emb_queue *q, void *stream, uint32_t *n_requests);
int emb_queue_wait(emb_queue *q, uint64_t ticket);
int emb_queue_destroy(emb_queue *q);

/* ------------------------------------------------------------------------------------------ */
/* multi-GPU routing for ROW-RANGE sharded tables, any number of indices per bag:                  */
/* counts first, payload second                                                                   */
/* ------------------------------------------------------------------------------------------ */
/* The bag loop being sharded is upmem/src/dpu/emb_dpu_lookup.c:106-116; the reference tells its devices the
 * lengths before every launch (emb_host.h:280-287) -- here the lengths are computed on the GPU and are the
 * FIRST message of the exchange, the payload is sized from them (no capacity to overflow, whatever the skew).
 *
 * A bag of a row-split table is cut into one SUB-BAG per shard owning some of its rows (indices keep their
 * order inside a sub-bag).  Shard d receives, per (source rank, table), an ordinary lookup -- local row ids +
 * bag-start offsets, exactly emb_lookup_desc's format -- and returns one partial pooled row per sub-bag;
 * the bag's owner adds the partial rows in shard order d = 0 .. n_shards-1 starting from +0: deterministic,
 * bit-identical to an unsharded lookup when a bag lives in one shard (one index per bag), within fp32
 * re-association (<= 1e-6 on DLRM-scale tables) otherwise.
 *
 * emb_route_bags enqueues a few small kernels on `stream` and fills (all DEVICE memory, sizes from
 * emb_route_bags_sizes; K = n_tables, N = n_shards, pad4(x) = x rounded up to a multiple of 4):
 *   meta  uint32 words:  counts[N][K+1][2]: entry [d][k], k < K = {n_sub, n_idx} of the request to shard d for table k;
 *                                           entry [d][K] = {peak request words, peak partial rows}: the largest piece
 *                                           this rank sends to / expects back from any ONE peer (the same pair in every
 *                                           d).  counts[d] is the message for peer d and goes out FIRST; from the K+1
 *                                           entries it receives from every peer a rank sizes the payload AND learns the
 *                                           job-wide largest piece, so all ranks agree on how many rounds a transfer
 *                                           that is too large for one RCCL group takes (emb_comm_exchange cuts every pair's
 *                                           bytes into <= 512-MiB pieces on both ends from the same byte count)
 *                        base  [N][K][2]  = word offsets in `send` of that request's offsets / indices arrays
 *                        piece [N+1]      = word offset of destination d's piece (piece[N] = words in all)
 *                        ret_row0[N][K]   = first partial row of (d, k) in the returned rows (see below)
 *                        mode             = how `slots` is encoded: 0 bags, 1 one index per bag
 *   send  uint32 words:  for d: for k: offsets[pad4(n_sub)] then local row ids[pad4(n_idx)]  (every array
 *                        16-byte aligned; the host derives the split sizes from `counts` with the same rule)
 *   slots (opaque; kept by the source for emb_unroute_bags, valid until the next emb_route_bags into the same
 *                        buffers).  mode 0: uint32[K][N][n_bags], slot of bag b's sub-bag in (d, k)'s request,
 *                        0xffffffff = none.  mode 1: uint32[K][n_bags], (d << 24) | slot.
 * One index per bag (every table: offsets == NULL, fixed_pooling == 1; n_shards <= 64, n_bags < 2^24 -- the Criteo
 * shape, emb_host.h:258-263 pushes one index list per table) takes a fast path with the same request format and the
 * same slot order: each index is read once (ballot ranks inside 1024-bag blocks, block prefixes, no atomics) and
 * the un-router reads one slot word and one row per bag instead of n_shards slot words.
 * The serving rank runs emb_lookup_batched over the received pieces (n_bags = n_sub, n_indices = n_idx, device
 * pointers) and returns, per source, the partial rows of k = 0 .. K-1 back to back (float[n_sub][dim], no
 * padding).  emb_unroute_bags then reads `recv` = the returned rows of shard 0, shard 1, ... back to back
 * and writes pooled[k][b][:].  dim must be a multiple of 4; indices are uint32; at most 64 tables per call. */
typedef struct emb_route_table {
    const void *indices;      /* DEVICE uint32[n_indices] (int64 through emb_route_bags_typed): global row ids */
    const void *offsets;      /* DEVICE bag starts, same width (last bag runs to n_indices), or NULL */
    uint64_t n_indices;
    uint32_t fixed_pooling;   /* offsets == NULL: offsets[b] = b * fixed_pooling */
    uint32_t rows_per_shard;  /* shard d owns rows [d*rows_per_shard, (d+1)*rows_per_shard) */
} emb_route_table;
int emb_route_bags_sizes(uint32_t n_tables, uint64_t n_bags, uint64_t total_indices, uint32_t n_shards,
                         uint64_t *send_bytes, uint64_t *meta_bytes, uint64_t *slots_bytes, uint64_t *work_bytes);
int emb_route_bags(emb_engine *e, const emb_route_table *tables /* host array */, uint32_t n_tables,
                   uint64_t n_bags, uint32_t n_shards, void *send, uint32_t *meta, uint32_t *slots,
                   void *work, void *stream);
/* emb_route_bags over int64 (or uint32) index / offset arrays: the request pieces are uint32 LOCAL row ids whatever the width
 * handed in (a shard holds fewer than 2^32 rows); an id no shard can hold (negative, or beyond the last shard's 32-bit reach)
 * travels as local id 0xffffffff, which no shard's table contains -- a validating server refuses it, it never wraps into range. */
int emb_route_bags_typed(emb_engine *e, const emb_route_table *tables /* host array */, uint32_t n_tables,
                         emb_index_type itype, uint64_t n_bags, uint32_t n_shards, void *send, uint32_t *meta,
                         uint32_t *slots, void *work, void *stream);
int emb_unroute_bags(emb_engine *e, const float *recv, const uint32_t *meta, const uint32_t *slots,
                     uint32_t n_tables, uint64_t n_bags, uint32_t n_shards, uint32_t dim,
                     float *pooled /* [n_tables][n_bags][dim] */, void *stream);

/* Host-side arithmetic of one exchange step (no GPU work, no engine): what a rank derives from the counts message it
 * SENT (the first N*(K+1)*2 words of its `meta`, copied to the host) and the counts messages it RECEIVED (one
 * [K+1][2] block per source, in source order) -- so that callers in any language need not restate the layout rule.
 *   req_out_words[d]   uint32 words of the request piece this rank sends to d      (split sizes of the request exchange)
 *   req_in_words[s]    words of the piece it receives from s
 *   ret_rows_back[d]   partial rows shard d returns to this rank                    (split sizes of the return exchange,
 *   ret_rows_served[s] partial rows this rank returns to source s                    in rows of dim floats)
 *   peak_req_bytes / peak_ret_bytes: the JOB's largest request / return piece in bytes (the maximum over the peaks
 *   entries of all received messages, this rank's own included): every rank computes the same two numbers. */
int emb_route_exchange_sizes(const uint32_t *sent, const uint32_t *received, uint32_t n_tables, uint32_t n_shards,
                             uint32_t dim, uint64_t *req_out_words, uint64_t *req_in_words, uint64_t *ret_rows_back,
                             uint64_t *ret_rows_served, uint64_t *peak_req_bytes, uint64_t *peak_ret_bytes);
/* The descriptors of the ONE fused lookup that serves everything a rank received: `req_recv` (DEVICE) holds the pieces
 * of source 0, 1, ... back to back as they arrived; shard_table_ids[k] is this rank's engine table for row-split table
 * k; the partial rows of (source s, table k) go to ret_send (DEVICE) back to back in (s, k) order -- the layout the
 * return exchange sends.  Writes at most N*K descriptors (pieces with no sub-bag are skipped), their number and the
 * launch's algorithmic bytes; pass the array to emb_lookup_batched(..., EMB_IDX_U32, EMB_MEM_DEVICE, stream). */
int emb_route_serve_descs(const uint32_t *received, uint32_t n_tables, uint32_t n_shards, uint32_t dim,
                          const uint32_t *shard_table_ids, const void *req_recv, float *ret_send,
                          emb_lookup_desc *descs, uint32_t *n_descs, uint64_t *algorithmic_bytes);

/* ---- optional native exchange (multi-GPU sharded lookup) --------------------------------------- */
/* All-to-all of byte ranges issued directly to RCCL (grouped ncclSend/ncclRecv) on the caller's stream:
 * the "indices in / pooled rows out" step of the sharded lookup without torch.distributed's per-call
 * host cost.  One emb_comm per process (= per GPU).  Bootstrap like NCCL: rank 0 calls
 * emb_comm_unique_id, sends the 128 bytes to the other ranks by any means (the Python side uses
 * torch.distributed.broadcast), every rank calls emb_comm_create.  RCCL is resolved at run time
 * (EMB_ERR_UNSUPPORTED if librccl.so cannot be found).  The reference's counterpart is the broadcast
 * push / gather pull of dpu_push_xfer (emb_host.h:258-287, :321).  Exercised with one rank and with
 * two to four ranks over RCCL's socket transport (several ranks on one GPU), never over xGMI.  It is the ONLY transport of the
 * RCCL-mode sharded step: emb_shard_* issues every transfer through emb_comm_exchange, and bench.py's N > 1 legs call
 * emb_shard_* -- torch.distributed carries the bootstrap (this id), the barriers and the job clock, never payload. */
typedef struct emb_comm emb_comm;
int emb_comm_unique_id(void *id128);
int emb_comm_create(emb_engine *e, const void *id128, int32_t rank, int32_t world, emb_comm **out);
/* send_off / recv_off: world+1 byte offsets each; peer p gets send[send_off[p] .. send_off[p+1]) and
 * delivers into recv[recv_off[p] .. recv_off[p+1]).  Enqueues only. */
int emb_comm_all_to_all(emb_comm *c, const void *send, const uint64_t *send_off, void *recv,
                        const uint64_t *recv_off, void *stream);
int emb_comm_destroy(emb_comm *c);

/* A list of point-to-point transfers issued as ONE RCCL group on `stream` (the all-to-all above is the special case of one
 * send and one receive per peer).  Transfers between a pair of ranks match in the order they are listed: the k-th send
 * to peer p pairs with p's k-th receive from this rank, and both sides must give the same byte count.  Zero-byte entries
 * are skipped (on both sides, since both know the size).  peer == own rank is allowed (a device copy inside RCCL). */
typedef struct emb_comm_op {
    int32_t peer;
    int32_t is_recv;   /* 0: send `bytes` from ptr, 1: receive `bytes` into ptr */
    void *ptr;         /* DEVICE memory */
    uint64_t bytes;
} emb_comm_op;
int emb_comm_exchange(emb_comm *c, const emb_comm_op *ops, uint32_t n_ops, void *stream);
int emb_comm_rank(const emb_comm *c, int32_t *rank, int32_t *world);

/* ---- peer group: ranks that read and write each other's HBM directly (no collectives library) -------------------- */
/* The substrate of the sharded lookup's collective-free exchange (EMB_SHARD_PEER_STORES below).  One process per GPU; every
 * rank calls emb_peer_create with the same job_tag (any string unique to the job) and world.  The group is set up through
 * one POSIX shared-memory segment (`/pimemb-<tag>`, unlinked again once every rank has mapped it) and HIP IPC handles:
 * each rank allocates an ARENA of arena_bytes -- one device allocation (fine-grained where the runtime can export it) that
 * every peer maps -- and everything a peer may touch (index / offset arrays, output rows) must be carved from it with
 * emb_peer_alloc (a bump allocator: nothing is freed before emb_peer_destroy).  The segment also carries small mailboxes
 * [dst][src][slot] for the per-batch handshake: written by tiny kernels in stream order, read by the host with plain loads.
 * Exercised with 2-4 processes on ONE GPU; never over xGMI (link rates and cross-link visibility are unmeasured). */
typedef struct emb_peer emb_peer;
int emb_peer_create(emb_engine *e, const char *job_tag, int32_t rank, int32_t world, uint64_t arena_bytes, emb_peer **out);
int emb_peer_alloc(emb_peer *p, uint64_t bytes, void **ptr);
int emb_peer_info(emb_peer *p, int32_t *rank, int32_t *world, void **arena, uint64_t *arena_bytes, uint64_t *used,
                  int32_t *fine_grained);
int emb_peer_barrier(emb_peer *p);   /* host barrier of the group through the shared segment (start-up, teardown, tests) */
int emb_peer_destroy(emb_peer *p);   /* call emb_peer_barrier first: a peer may still be reading this rank's arena */
/* A last word for a process that may die inside code nobody has run across real links yet (bench.py's peer-store leg, run
 * AFTER the RCCL leg's numbers are in hand): while `line` is set, a fatal signal (SIGABRT -- what the runtime raises on a GPU
 * memory fault --, SIGSEGV, SIGBUS; and SIGTERM: what a launcher sends the surviving ranks when another rank died) writes it to file descriptor `fd` and ends the process with `status` instead of a core dump, so the
 * numbers measured before survive.  line = "" : just end with `status` (the ranks that print nothing).  line = NULL: take the
 * handlers out again.  The text is copied.  Async-signal-safe: write(2) and _exit(2), nothing else. */
int emb_peer_last_words(const char *line, int fd, int status);

/* ------------------------------------------------------------------------------------------ */
/* sharded lookup: ONE call per batch                                                            */
/* ------------------------------------------------------------------------------------------ */
/* The reference serves every device from one lookup() call: it pushes the indices to all DPUs (emb_host.h:258-270),
 * launches them (:297) and pulls every result into the caller's final_results (:312-321).  emb_shard is that call for
 * tables sharded over the GPUs of one node -- one process per GPU, every rank making the SAME sequence of calls:
 *
 *   emb_shard_submit(s, in, n_bags, stream, &seq)   hand over this rank's batch: per table its indices / offsets / where the
 *                                                    pooled rows go.  Enqueue only (plus one short host wait, see below).
 *   emb_shard_wait(s, seq, stream)                  make `stream` wait until batch seq's pooled rows are in place (a no-op for
 *                                                    the submit stream itself: they are complete there in stream order)
 *   emb_shard_flush(s)                              push every submitted batch through its remaining stages
 *   emb_shard_lookup(...)                           submit + flush + wait: the synchronous form
 *
 * Placement of a table:
 *   EMB_PLACE_REPLICATED  every rank holds it: looked up locally, nothing travels;
 *   EMB_PLACE_WHOLE       one owner rank holds it: the bags' indices travel to the owner exactly as the caller passed
 *                         them (straight out of the caller's buffers), the owner's fused lookup pools them and the pooled
 *                         rows arrive straight in the caller's output buffer -- no routing kernel, no copy on either side;
 *   EMB_PLACE_ROWS        split by ROW RANGE over all ranks (rank r holds rows [r*rows_per_shard, (r+1)*rows_per_shard)):
 *                         every bag is cut into per-shard sub-bags on the GPU (emb_route_bags), each shard returns one
 *                         partial row per sub-bag, the bag's owner adds them in shard order (emb_unroute_bags).
 * DIRECT path: when a batch has ONE index per bag on every row-split table and no peer sits behind RCCL (every peer is this
 * rank itself or a peer-store peer), those tables are not routed at all -- every shard scans the requester's raw index array
 * and serves the bags whose row it holds straight into the requester's output (emb_lookup_ranged): no router, no counts, no
 * un-router.  Decided per batch and per rank; same bits (a one-index bag's pooled row IS the table row).  Not taken by a
 * shard created with EMB_SHARD_NO_DIRECT, or whose rows are not 16-byte multiples up to 1 KiB.  A shard created with
 * EMB_SHARD_CHECK_SERVED keeps it: its ranged launches COUNT the bags they serve (emb_lookup_ranged_counted) and the
 * requesting rank compares the sums with its bag count (see the flag below).
 * Counts first, payload second: what a rank will send each peer (sub-bags and indices per table) leaves before the payload,
 * so nothing has a capacity that skewed indices could overflow.  The one host wait of a batch is for those counts.
 *
 * Stages of batch b.  Every kernel runs on the CALLER's stream (pass the same stream to every submit; a change is
 * honoured with one event), every transfer on one internal stream; events cross between the two only where a transfer
 * really happens -- none at all with one rank:
 *   R(b) route + counts out | L(b) local lookup of the replicated tables        at submit(b)
 *   Q(b) host reads the counts, request pieces travel                           at submit(b + d_req)
 *   S(b) ONE fused lookup over every piece received; T(b) pooled rows return    at submit(b + d_serve)
 *        (L of the batch submitted in that call rides in the same launch: L(b + d_serve) next to S(b) -- at depth 0 the
 *        batch's own L(b) -- so a call enqueues one lookup kernel; on the direct path the shards' ranged lookups join it too)
 *   U(b) partial rows added in shard order into the caller's buffers            at submit(b + d_un)
 * with (d_req, d_serve, d_un) = (0, 0, 0) for depth 0, (0, 1, 1) for depth 1, (1, 2, 2) for depth 2 and (1, 2, 3) for depth 3:
 * from depth 2 on the counts a rank waits for were sent a whole call earlier, so the host wait is short; at depth 3 the
 * returned rows of batch b travel while the lookup of batch b + 1 runs (the un-router is enqueued a call later, behind it).
 * Within a call the transfer stream gets the newest batch's counts, then the requests of the one before, then the returned
 * rows of the one before that -- a lookup never queues behind a return transfer.  After submit(b + d_un) the pooled rows of
 * batch b are complete in stream order on the caller's stream; emb_shard_wait orders another stream behind them.
 * A batch's inputs and outputs belong to the library from its submit until then.  RCCL transfers of all ranks are issued in
 * the same order because all ranks make the same calls: submit / flush are COLLECTIVE (every rank, same order; a rank
 * with nothing to look up submits n_bags = 0).  Lookups that recur byte for byte (same buffers, same lengths: static batch
 * slots) are served by prepared plans from their second sighting on.
 *
 * Threading: one caller thread per shard object (the reference assumes a single caller too, emb_host.h:32-33); several shard
 * objects may share an engine.
 * Indices and offsets are uint32 (the reference's width, emb_host.h:234) or int64 (torch's; emb_shard_input.index_type, one
 * width per batch, every rank of a job the same); every table has `dim` columns; a batch has the same number of bags for
 * every table (the reference's MAX_NR_BATCHES).  At most 64 row-split tables per shard object. */
typedef struct emb_shard emb_shard;
#define EMB_PLACE_REPLICATED 0u
#define EMB_PLACE_WHOLE 1u
#define EMB_PLACE_ROWS 2u
typedef struct emb_shard_table {
    uint32_t placement;      /* EMB_PLACE_* */
    int32_t owner;           /* EMB_PLACE_WHOLE: the rank that holds the table */
    uint32_t engine_table;   /* id, in THIS rank's engine, of what this rank holds of the table: the whole table
                                (replicated; whole, on its owner) or this rank's row range (rows).  Ignored on ranks that
                                hold nothing of it. */
    uint32_t rows_per_shard; /* EMB_PLACE_ROWS */
} emb_shard_table;
typedef struct emb_shard_input {      /* one per table, in table order */
    const void *indices;      /* DEVICE uint32[n_indices] or int64[n_indices] (index_type) */
    const void *offsets;      /* DEVICE bag starts of the same width (last bag runs to n_indices), or NULL */
    uint64_t n_indices;
    uint32_t fixed_pooling;   /* offsets == NULL: offsets[b] = b * fixed_pooling */
    uint32_t index_type;      /* EMB_IDX_U32 (0: the reference's width, emb_host.h:234) or EMB_IDX_I64 (torch's: DLRM hands
                                 int64 tensors over) -- the same for every table of one batch.  int64 arrays
                                 are used IN PLACE: the router reads them, replicated / whole tables and the direct path gather
                                 through them, whole tables' arrays travel as they are (8 bytes per id); nothing is narrowed,
                                 and an id outside its table stays what it is for the checks below to refuse */
    float *pooled;            /* DEVICE float[n_bags][dim] */
} emb_shard_input;
#define EMB_SHARD_SELF_VIA_COMM 1u /* pieces a rank addresses to ITSELF go through RCCL like any other (rehearsal / A-B);
                                      default: they are served in place -- no transfer, no copy */
#define EMB_SHARD_CHECK_SERVED 2u  /* indices are not trusted (the reference never checks: emb_dpu_lookup.c:113).  Two mechanisms, chosen
                                      per batch; every rank of the job must pass the flag alike (a mismatch is refused):
                                      ROUTED batches -- the fused lookup validates what it serves first (emb_lookup_batched_checked);
                                      a finding makes emb_shard_submit / _flush return EMB_ERR_RANGE on the SERVING rank after the batch
                                      has gone through all its stages (the offending pieces pool to zero rows), so no peer is left in a
                                      transfer.  ONE-INDEX batches on the direct path (and, with no peer behind RCCL, any call whose
                                      lookups are all one index per bag) -- nothing is validated up front: every launch counts the bags
                                      it serves, the counts travel back with the "served" handshake, and the REQUESTING rank compares:
                                      each replicated / whole table must have served all its bags, the shards of a row-split table
                                      must add up to the bag count.  A shortfall = an index no rank holds: EMB_ERR_RANGE on the requesting
                                      rank when the batch completes (emb_last_error names the table and the count), ONCE per batch;
                                      the other bags of the batch are correct, the next batch is unaffected.  Both mechanisms leave
                                      the same thing in the outputs: an offending bag's pooled row is ZEROS (routed: the offending
                                      pieces pool to zero rows; counted: the last shard / the holder of a whole table writes zero rows
                                      for ids beyond every range, EMB_RANGE_OPEN_END) -- never stale bytes of an earlier batch */
#define EMB_SHARD_PEER_STORES 4u   /* the collective-free exchange (needs emb_shard_config.peer): nothing travels through RCCL.  A
                                      rank posts what it asks each peer for -- counts and buffer addresses -- into the peer's
                                      mailbox; the owner's ONE fused lookup gathers the requester's indices IN PLACE (its mapped
                                      arena) and stores the pooled rows straight into the requester's HBM; a one-thread kernel
                                      behind it raises "served".  No request transfer, no return transfer, no staging copies; the
                                      host's waits are polls of mailbox words.  The caller's index / offset / output buffers of
                                      tables held by OTHER ranks, whole or split, must come from emb_peer_alloc. */
#define EMB_SHARD_NO_DIRECT 8u      /* never take the direct path for one-index-per-bag row-split tables (below): always route */
#define EMB_SHARD_DEFER_REPORT 16u  /* with EMB_SHARD_CHECK_SERVED: what the REQUESTING rank learns from the served counts of batch b is
                                      compared -- and a shortfall reported -- by a LATER call instead of inside the call that completes
                                      batch b: by the first emb_shard_submit / _lookup / emb_shard_wait(b) that finds b's counts arrived (they
                                      only look, never wait; at the latest the submit that recycles b's slot, six batches on), by an
                                      emb_shard_flush the caller makes, or by emb_shard_report (these two wait for the counts).  The call that completes a
                                      batch then never waits for the GPU: without the flag the synchronous emb_shard_lookup polls the
                                      counts its own launch is still producing (59 -> 73 us per call on the C4 share,
                                      profiles/r05/dist_world1.md).  The error names the batch it belongs to; unserved bags hold zero
                                      rows either way.  Like a device-side assert: a call or a few late, never lost --
                                      emb_shard_destroy prints a finding nobody collected to stderr. */
typedef struct emb_shard_config {
    uint32_t n_tables;
    uint32_t dim;
    uint32_t depth;           /* 0 .. 3 (see above) */
    uint32_t flags;           /* EMB_SHARD_* */
    const emb_shard_table *tables;
    emb_peer *peer;           /* EMB_SHARD_PEER_STORES: the group (rank / world are taken from it; comm may be NULL) */
} emb_shard_config;
/* what the last completed stages cost and moved (reporting; cumulative since create / reset) */
typedef struct emb_shard_stats {
    uint64_t n_batches;               /* batches that went through all stages */
    uint64_t bytes_to_peers;          /* counts + requests + returned rows handed to OTHER ranks */
    uint64_t bytes_to_self;           /* the same addressed to this rank itself (served in place unless SELF_VIA_COMM) */
    uint64_t served_algorithmic_bytes;/* algorithmic bytes of the fused lookups over received pieces */
    uint64_t local_algorithmic_bytes; /* ... of the lookups of the replicated tables */
    uint64_t served_sub_bags, served_indices;
    double us_host_submit;            /* host time inside emb_shard_submit / _flush, all stages */
    double us_host_wait_counts;       /* ... of it: waiting for the counts */
    double us_host_wait_served;       /* ... of it (EMB_SHARD_PEER_STORES): waiting for the peers' "served" words */
    /* device time of the four kernel families, HIP events on their own streams, summed over the batches that ran while
     * emb_shard_set_kernel_timing was on: R router, L local lookup, S fused lookup over received pieces, U un-router */
    double us_kernel_route, us_kernel_local, us_kernel_serve, us_kernel_unroute;
    uint64_t n_timed_batches;
    double us_kernel_direct;          /* ... D: the ranged one-hot lookups of the direct path */
} emb_shard_stats;
/* comm may be NULL for a world of one rank (everything is "self"). */
int emb_shard_create(emb_engine *e, emb_comm *comm, const emb_shard_config *cfg, emb_shard **out);
int emb_shard_submit(emb_shard *s, const emb_shard_input *in, uint64_t n_bags, void *stream, uint64_t *seq);
int emb_shard_flush(emb_shard *s);
int emb_shard_wait(emb_shard *s, uint64_t seq, void *stream);
int emb_shard_lookup(emb_shard *s, const emb_shard_input *in, uint64_t n_bags, void *stream);
/* EMB_SHARD_DEFER_REPORT: compare the served counts of every completed batch not looked at yet, now (waits for their publish
 * kernels): EMB_OK, or EMB_ERR_RANGE with the first finding in emb_last_error().  A no-op (EMB_OK) without the flag. */
int emb_shard_report(emb_shard *s);
int emb_shard_get_stats(emb_shard *s, emb_shard_stats *out, int reset);
/* Bracket the library's kernels with timing events (off by default: an event between two kernels costs GPU time itself). */
int emb_shard_set_kernel_timing(emb_shard *s, int on);
/* Per-(peer, table) request counts {sub-bags, indices} of the row-split tables this rank SENT for batch seq: uint32
 * [n_ranks][n_row_split][2] (reporting; valid until the batch's slot is reused, six submits later). */
int emb_shard_sent_counts(emb_shard *s, uint64_t seq, uint32_t *counts, uint32_t capacity_words);
int emb_shard_destroy(emb_shard *s);

/* ------------------------------------------------------------------------------------------ */
/* (2) reference-compatible entry points (same names, argument meaning and return values)      */
/* ------------------------------------------------------------------------------------------ */

struct dpu_set_t; /* opaque here; the reference returns its global DPU set, emb_host.h:182 */

/* emb_host.h:41-48 */
typedef struct dpu_runtime_totals {
    double execution_time_prepare;
    double execution_time_populate_copy_in;
    double execution_time_copy_in;
    double execution_time_copy_out;
    double execution_time_aggregate_result;
    double execution_time_launch;
} dpu_runtime_totals;

/* Runtime replacement for -DNR_TABLES -DNR_COLS -DMAX_NR_BATCHES -DMAX_INDICES_PER_BATCH
 * (upmem/Makefile:69-81).  If never called, the same-named environment variables are read on the
 * first populate_mram (run.sh:40-45 exports exactly these).  Calling it again after tables were
 * loaded drops them (a reference rebuild does the same). */
int emb_configure(uint32_t nr_tables, uint32_t nr_cols, uint32_t max_nr_batches,
                  uint32_t max_indices_per_batch);

/* emb_host.h:136.  Uploads int32 fixed-point column `col` of table `table_id` (nr_rows entries).
 * First call creates the process-global engine (emb_host.h:155-160).  `runtime` may be NULL
 * (load_generator.c:34); if not, execution_time_populate_copy_in accumulates milliseconds.
 * Returns the engine handle as the opaque pointer lookup() expects; NULL on error. */
struct dpu_set_t *populate_mram(uint32_t table_id, uint64_t nr_rows, uint32_t col,
                                int32_t *table_data, dpu_runtime_totals *runtime);

/* emb_host.h:234.  indices[t]: uint32[MAX_INDICES_PER_BATCH*MAX_NR_BATCHES]; offsets[t]:
 * uint32[MAX_NR_BATCHES] bag starts (last bag runs to INDICES_LEN, emb_dpu_lookup.c:109);
 * final_results[t]: float[MAX_NR_BATCHES*NR_COLS] row-major [bag][col] (emb_host.h:210), all HOST
 * pointers owned by the caller.  Synchronous.  latency_print == 1 prints the six stage latencies
 * (emb_host.h:395-402).  Always returns NULL like the reference (emb_host.h:403); on error the text
 * is in emb_last_error() and final_results is untouched. */
int32_t *lookup(uint32_t **indices, uint32_t **offsets, float **final_results,
                void *dpu_set_ptr_untyped, int64_t latency_print);

/* The engine behind the compat entry points (NULL before the first populate_mram). */
emb_engine *emb_compat_engine(void);
/* Tear the process-global compat engine down (tests; the reference never frees it). */
int emb_compat_reset(void);

#ifdef __cplusplus
}
#endif
#endif /* PIMEMB_H */
