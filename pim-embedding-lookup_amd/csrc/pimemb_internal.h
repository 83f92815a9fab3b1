// Internal declarations shared by the HIP kernels and the host-side engine (not installed).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pimemb.h"

namespace pimemb {

// One table's share of a fused launch as the kernel sees it (HBM-resident array, 64 B each so a
// workgroup fetches its descriptor with one scalar load burst).
struct alignas(64) DevDesc {
    const void *weights;    // row-major [nr_rows][dim] of the table dtype
    const void *indices;    // IdxT[n_idx]
    const void *offsets;    // IdxT[n_bags] bag starts, or nullptr when fixed_pooling > 0
    float *out;             // float[n_bags][dim]
    uint64_t n_idx;
    uint64_t n_bags;
    uint64_t nr_rows;       // only read by the validation kernel
    uint32_t fixed_pooling; // L > 0: offsets[b] = b*L (load_generator.c:88)
    uint32_t n_tiles;       // ceil(n_bags / bags_per_tile) for this launch geometry
};
static_assert(sizeof(DevDesc) == 64, "DevDesc must stay one 64-byte line");

// Launch geometry for one (dtype, dim) group.
struct LaunchGeom {
    uint32_t lanes_per_row;   // power of two, 1..64: lanes that cooperate on one bag
    uint32_t chunks;          // 16-byte pieces per row actually used (<= lanes_per_row)
    uint32_t bags_per_tile;   // bags one 256-thread workgroup finishes per tile
};

// Returns EMB_OK / EMB_ERR_UNSUPPORTED and fills `g` for a table shape.
int geometry_for(emb_dtype dtype, uint32_t dim, LaunchGeom *g);

// Enqueue the fused gather + segment-sum over `n_descs` descriptors (all of one dtype/dim).
// max_tiles = max over descs of n_tiles.  Pure enqueue: no allocation, copy or sync.
hipError_t launch_bag_sum(const DevDesc *d_descs, uint32_t n_descs, uint32_t max_tiles,
                          emb_dtype dtype, emb_index_type itype, const LaunchGeom &g,
                          hipStream_t stream);

// Scatter an int32 column (device buffer, nr_rows entries) into column `col` of a row-major
// [nr_rows][dim] int32 table: the inverse of alloc_buffers' split (emb_host.h:116-118).
hipError_t launch_scatter_column(int32_t *table, const int32_t *column, uint64_t nr_rows,
                                 uint32_t dim, uint32_t col, hipStream_t stream);

// Count out-of-range indices and broken offsets (debug check).  *d_bad must be zeroed by the caller.
hipError_t launch_validate(const DevDesc *d_descs, uint32_t n_descs, emb_index_type itype,
                           unsigned long long *d_bad, hipStream_t stream);

// Record the calling thread's error text (returned by emb_last_error()) and hand `code` back.
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

}  // namespace pimemb
