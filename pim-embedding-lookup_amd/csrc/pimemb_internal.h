// Internal declarations shared by the HIP kernels and the host-side engine (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pimemb.h"
#include "pimemb_bag_kernels.h"

namespace pimemb {

// Row shape of one (dtype, dim) group.
struct LaunchGeom {
    uint32_t lanes_per_row;   // power of two, 1..64: lanes that cooperate on one bag
    uint32_t chunks;          // 16-byte pieces per row actually used (<= lanes_per_row)
    uint32_t scalar_lanes;    // != 0: rows that are not 16-byte multiples (or wider than 1 KiB) -- the
                              // any-dim kernels with this many threads per bag; chunks = dim
    bool anydim_vec;          // rows are 4-byte multiples: one thread per 16-byte piece (else per element)
};

// Which bag kernel a launch uses.
enum KernelKind : uint32_t {
    KERNEL_WAVEBATCH = 0,  // 64 bags per wavefront, coalesced bounds, one-hot fast path: big batches
    KERNEL_GROUP = 1,      // one lane group per bag, finest granularity: pooled launches, small batches
    KERNEL_WAVEBATCH2 = 2, // 2 x 64 bags per wavefront: very big one-hot launches
    KERNEL_ANYDIM = 3,     // one thread per output element: any dim (e.g. DLRM's default 2), not a tuned path
    KERNEL_HOT = 4         // lane-group kernel with the tables' hot rows staged in LDS (emb_set_hot_rows)
};

constexpr size_t kHotLdsBudget = 62u << 10;   // LDS bytes a hot set (rows + hash) may take per workgroup

// Returns EMB_OK / EMB_ERR_UNSUPPORTED and fills `g` for a table shape.
int geometry_for(emb_dtype dtype, uint32_t dim, LaunchGeom *g);

// Bags one workgroup finishes per tile for this kernel kind and row shape.
uint32_t bags_per_tile(KernelKind kind, const LaunchGeom &g);

// Picks the kernel for a launch group from its bag and index counts.
KernelKind choose_kernel(uint64_t total_bags, uint64_t total_indices, const LaunchGeom &g);

// Enqueue the fused gather + segment-sum over `n_descs` descriptors (all of one dtype/dim).
// d_xmap == nullptr: 2-D grid (max_tiles x n_descs).  Otherwise the XCD-aware 1-D map built by
// pimemb_xcd_map.h with `xgrid` workgroups (xdirect: expanded to one entry per workgroup).
// Pure enqueue: no allocation, copy or sync.
hipError_t launch_bag_sum(const DevDesc *d_descs, uint32_t n_descs, uint32_t max_tiles,
                          emb_dtype dtype, emb_index_type itype, const LaunchGeom &g,
                          KernelKind kind, const uint32_t *d_xmap, uint32_t xgrid, bool xdirect,
                          hipStream_t stream, bool ranged = false);

// Pooled launch with hot rows in LDS: `wgs` persistent workgroups per descriptor, `lds_bytes` of dynamic LDS.
hipError_t launch_bag_sum_hot(const DevDesc *d_descs, uint32_t n_descs, uint32_t wgs, uint32_t lds_bytes,
                              emb_dtype dtype, emb_index_type itype, const LaunchGeom &g, hipStream_t stream);

// ranged (wave-batch kinds, one index per bag): a descriptor serves only the bags whose row falls into
// [row_lo, row_lo + nr_rows) -- row_lo in DevDesc::pad_[0] -- as out[b] = W[idx[b] - row_lo]; other bags are left untouched
// (kRangeOpenEnd in pad_[0]: bags whose id lies at or beyond the end of the range are written as zero rows).

// Scatter an int32 column (device buffer, nr_rows entries) into column `col` of a row-major
// [nr_rows][dim] int32 table: the inverse of alloc_buffers' split (emb_host.h:116-118).
hipError_t launch_scatter_column(int32_t *table, const int32_t *column, uint64_t nr_rows,
                                 uint32_t dim, uint32_t col, hipStream_t stream);

// Input validation (see validate_kernel): counts out-of-range indices / broken offsets over the descriptors of a launch
// image and reports ONE pinned, device-visible word, written by the kernel's last workgroup: validate_word(seq, n) -- the
// call's sequence number over the number of offending values (saturating).  ctl: one ValidateCtl of HBM per call in flight
// (the engine keeps a ring of them), zeroed once at creation; the reporting workgroup leaves it zeroed for the next user.
// poison: a finding zeroes n_tiles of every descriptor, so lookup kernels enqueued behind it do nothing.
//
// Who is last: returning atomics on ONE address retire at ~20 per microsecond on this chip, a thousand workgroups drawing
// tickets from one counter would cost more than the checking.  So two levels: workgroup w signs off at sub[w % kValLanes]
// (each counter on a line of its own, kValStride bytes apart), and the workgroup that completes a sub-counter signs off at
// `tickets` with the sub-counter's findings; the one that completes THAT reports.  (Rounds 3-5 had a one-thread kernel behind big grids for this: 4.8 us of
// stream time per checked call, two PCIe writes with a system fence between them.)
#ifndef PIMEMB_VALIDATE_LANES
#define PIMEMB_VALIDATE_LANES 32
#endif
constexpr uint32_t kValLanes = PIMEMB_VALIDATE_LANES;
#ifndef PIMEMB_VALIDATE_STRIDE
#define PIMEMB_VALIDATE_STRIDE 1024
#endif
constexpr uint32_t kValStride = PIMEMB_VALIDATE_STRIDE;     // bytes between two counters of a ValidateCtl
// A counter: finished workgroups in the low kValTicketBits bits, what they found above.
constexpr uint32_t kValTicketBits = 24;
constexpr unsigned long long kValTicketMask = (1ull << kValTicketBits) - 1;
struct ValidateCtl {
    unsigned long long tickets;   // completed sub-counters (small grids: finished workgroups) | offending values << kValTicketBits
    char pad_[kValStride - 8];
    struct Lane {
        unsigned long long done;  // finished workgroups of this lane | what they found << kValTicketBits
        char pad_[kValStride - 8];
    } sub[kValLanes];
};
constexpr unsigned long long kValCountMask = 0xffffffull;     // a verdict word carries min(offending values, this)
constexpr unsigned long long validate_word(unsigned long long seq, unsigned long long bad) {
    return (seq << 24) | (bad < kValCountMask ? bad : kValCountMask);
}
uint32_t validate_workgroups(uint64_t max_items /* largest n_idx / n_bags of a descriptor */, emb_index_type itype);
hipError_t launch_validate(DevDesc *d_descs, uint32_t n_descs, emb_index_type itype, ValidateCtl *ctl, unsigned long long *result,
                           unsigned long long seq, uint32_t wgs_per_desc, bool poison, hipStream_t stream);

// Row-range routing of variable-length bags (pooled lookups over row-split tables; see pimemb.h,
// emb_route_bags).  All pointers are device pointers; the kernels (four; three on the one-index-per-bag fast path)
// are enqueued on `stream`.
constexpr uint32_t kRouteBagMaxTables = 64;
struct RouteBagDesc {
    const void *indices;      // uint32, or int64 with idx64 (launch_route_bags)
    const void *offsets;
    uint64_t n_indices;
    uint32_t fixed_pooling;
    uint32_t rows_per_shard;
};
hipError_t launch_route_bags(const RouteBagDesc *tables, uint32_t n_tables, uint64_t n_bags, uint32_t n_shards,
                             uint32_t *send, uint32_t *meta, uint32_t *slots, uint32_t *work, hipStream_t stream,
                             bool idx64 = false);
uint32_t route_bags_meta_words(uint32_t n_tables, uint32_t n_shards);   // uint32 words of `meta`
hipError_t launch_unroute_bags(const float *recv, const uint32_t *meta, const uint32_t *slots, uint32_t n_tables,
                               uint64_t n_bags, uint32_t n_shards, uint32_t dim, float *pooled, hipStream_t stream);
// The same with one output pointer per table (the sharded step writes straight into the caller's per-table buffers).
hipError_t launch_unroute_bags_to(const float *recv, const uint32_t *meta, const uint32_t *slots, uint32_t n_tables,
                                  uint64_t n_bags, uint32_t n_shards, uint32_t dim, float *const *pooled_of_table,
                                  hipStream_t stream);
// Word offsets inside a routing `meta` block (emb_route_bags): counts of (peer d, table k), first request word of peer d's
// piece, and the number of leading words that hold all counts.
uint32_t route_meta_counts_words(uint32_t n_tables, uint32_t n_shards);
uint32_t route_meta_piece_word(uint32_t n_tables, uint32_t n_shards);
// Sharded step helpers: up to three word arrays HBM -> pinned host + a flag behind them; zero fill of a few words.
hipError_t launch_publish_words(const uint32_t *const src[3], const uint32_t n[3], uint32_t *dst_host,
                                unsigned long long *flag_host, unsigned long long value, hipStream_t stream);
hipError_t launch_zero_words(uint32_t *p, uint32_t n, hipStream_t stream);
// n[i] uint32 words at src[i] -> int64 words at dst[i], for up to kWidenSegs segments in one launch (see widen_words_kernel).
constexpr uint32_t kWidenSegs = 16;
struct WidenArgs {
    const uint32_t *src[kWidenSegs];
    long long *dst[kWidenSegs];
    uint64_t n[kWidenSegs];
    uint32_t n_seg;
};
hipError_t launch_widen_words(const WidenArgs &a, hipStream_t stream);
// dst[i][:] = src[d_ids[i]][:], n rows of row_bytes (a multiple of 16) each, ids in device memory (emb_set_hot_rows).
hipError_t launch_gather_rows(void *dst, const void *src, const unsigned long long *d_ids, uint32_t n, uint32_t row_bytes, hipStream_t stream);

// Record the calling thread's error text (returned by emb_last_error()) and hand `code` back.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace pimemb
