// pimemb_kernels.hip -- instantiates the bag kernels of pimemb_bag_kernels.h for the library and
// holds the two small helper kernels (column scatter, input validation).  gfx950 only.
#include <cstdlib>

#include "pimemb_internal.h"
#include "pimemb_peer.h"

namespace pimemb {
namespace {

// Shipped configurations, chosen with tools/tune_bag_kernels.hip on MI355X (DESIGN.md "Tuning"):
//   big batches : wave-batch kernel, 64-thread workgroups (finest interleave under the XCD map),
//                 <= 64 VGPRs (8 waves/SIMD), non-temporal pooled-row stores, speculative one-hot
//                 index prefetch;
//   pooled launches and small batches: lane-group kernel, 256-thread workgroups, a lane group
//                 loads a window of indices coalesced and broadcasts them by shuffle (-8 % on
//                 dim-128 pooling-32 Zipf, neutral elsewhere).
// Like the reference (an out-of-range index is a wild MRAM read, emb_dpu_lookup.c:113) the shipped
// kernels do not check indices or offsets: clamping row ids and bag ends costs 2.5 % on the headline
// shape (tools/tune_bag_kernels.hip).  Build with -DPIMEMB_CLAMP_INPUTS=1 for kernels that turn
// malformed input into garbage rows instead of out-of-bounds accesses; emb_validate_inputs checks
// inputs explicitly.
#ifndef PIMEMB_CLAMP_INPUTS
#define PIMEMB_CLAMP_INPUTS 0
#endif
constexpr bool kClampInputs = PIMEMB_CLAMP_INPUTS != 0;
//                        BLOCK U  ntS   ntM   inflight minW batches ntRow  spec
using WaveCfg = BagCfg<64, 8, true, false, 8, 8, 1, false, true, false, kClampInputs>;
// very big one-hot launches: two 64-bag batches per wavefront (8 gathers in flight per lane), 128-thread
// workgroups, unroll 4 on the general path so the kernel still fits 64 VGPRs without spilling: -4.5 %
using Wave2Cfg = BagCfg<128, 4, true, false, 8, 8, 2, false, true, false, kClampInputs>;
using GroupCfg = BagCfg<256, 8, true, false, 8, 1, 1, false, false, /*IDX_SHUFFLE*/ true, kClampInputs>;
// fp16 rows accumulate into 8 fp32 registers per gather instead of 4 and re-deal their output pieces by
// shuffle (store_row): under the 64-VGPR cap of the fp32 configurations the wave-batch kernels would spill
// 36-340 bytes per lane, so the fp16 instantiations target 4 waves per SIMD (128 VGPRs, no spills).
// One-hot fp16 lookups, 26 Kaggle-sized tables, B = 39292, us per launch before -> after (store_row +
// this occupancy; tools/f16_onehot_probe.py): dim 16 31.9 -> 21, dim 32 58.8 -> 31, dim 64 109 -> 52,
// dim 128 222 -> 97 (8.2 TB/s algorithmic).
#ifndef PIMEMB_F16_MINW
#define PIMEMB_F16_MINW 4
#endif
using WaveCfgF16 = BagCfg<64, 8, true, false, 8, PIMEMB_F16_MINW, 1, false, true, false, kClampInputs>;
using Wave2CfgF16 = BagCfg<128, 4, true, false, 8, PIMEMB_F16_MINW, 2, false, true, false, kClampInputs>;
template <int DT> struct WaveCfgOf { using One = WaveCfg; using Two = Wave2Cfg; };
template <> struct WaveCfgOf<EMB_F16> { using One = WaveCfgF16; using Two = Wave2CfgF16; };
// pooled launches over tables with a hot-row set: persistent 1024-thread workgroups stage the set into LDS
using HotCfg = BagCfg<1024, 8, true, false, 8, 1, 1, false, false, /*IDX_SHUFFLE*/ true, kClampInputs>;
constexpr int kBlock = 256;  // helper kernels below

template <typename IdxT, int DT, int L>
void launch_one(const DevDesc *d, uint32_t n, uint32_t max_tiles, const LaunchGeom &g_in, KernelKind kind,
                const uint32_t *xmap, uint32_t xgrid, bool xdirect, bool ranged, hipStream_t s) {
    const dim3 grid = xmap ? dim3(xgrid, 1, 1) : dim3(max_tiles, n, 1);
    struct { uint32_t chunks; } g{g_in.chunks | ((xmap && xdirect) ? kXmapDirect : 0u)};
    using One = typename WaveCfgOf<DT>::One;
    using Two = typename WaveCfgOf<DT>::Two;
    // developer A/B (tools/onehot_occupancy_sweep.sh): bytes of dynamic LDS per workgroup of the one-batch wave-batch launch -- LDS the
    // kernel never touches, reserved to CAP how many of its wavefronts a CU holds at once (160 KB per CU)
    static const uint32_t lds_pad = getenv("PIMEMB_WAVEBATCH_LDS_PAD") ? (uint32_t)atoi(getenv("PIMEMB_WAVEBATCH_LDS_PAD")) : 0u;
    if (ranged) {      // launch_bag_sum lets the two wave-batch kinds through only (uint32 and int64 indices alike)
        if (kind == KERNEL_WAVEBATCH)
            hipLaunchKernelGGL((bag_sum_wavebatch_kernel<IdxT, DT, L, One, true>), grid, dim3(One::kBlock), lds_pad, s, d, g.chunks, xmap);
        else if constexpr (L <= 4)
            hipLaunchKernelGGL((bag_sum_wavebatch_kernel<IdxT, DT, L, Two, true>), grid, dim3(Two::kBlock), 0, s, d, g.chunks, xmap);
        return;
    }
    if (kind == KERNEL_WAVEBATCH) {
        hipLaunchKernelGGL((bag_sum_wavebatch_kernel<IdxT, DT, L, One>), grid, dim3(One::kBlock), lds_pad, s, d, g.chunks, xmap);
    } else if (kind == KERNEL_WAVEBATCH2) {
        // choose_kernel hands out the two-batch geometry for <= 4 lanes per row only; wider rows are not
        // instantiated (they would spill under the 64-VGPR cap and are never launched)
        if constexpr (L <= 4)
            hipLaunchKernelGGL((bag_sum_wavebatch_kernel<IdxT, DT, L, Two>), grid, dim3(Two::kBlock), 0, s, d, g.chunks, xmap);
    } else
        hipLaunchKernelGGL((bag_sum_group_kernel<IdxT, DT, L, GroupCfg>), grid, dim3(GroupCfg::kBlock), 0,
                           s, d, g.chunks, xmap);
}

template <typename IdxT, int DT>
hipError_t launch_lpr(const DevDesc *d, uint32_t n, uint32_t max_tiles, const LaunchGeom &g,
                      KernelKind kind, const uint32_t *xmap, uint32_t xgrid, bool xdirect, bool ranged, hipStream_t s) {
    switch (g.lanes_per_row) {
#define PIMEMB_CASE(L)                                                          \
    case L:                                                                     \
        launch_one<IdxT, DT, L>(d, n, max_tiles, g, kind, xmap, xgrid, xdirect, ranged, s); \
        break;
        PIMEMB_CASE(1)
        PIMEMB_CASE(2)
        PIMEMB_CASE(4)
        PIMEMB_CASE(8)
        PIMEMB_CASE(16)
        PIMEMB_CASE(32)
        PIMEMB_CASE(64)
#undef PIMEMB_CASE
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename IdxT>
hipError_t launch_dtype(const DevDesc *d, uint32_t n, uint32_t max_tiles, emb_dtype dtype,
                        const LaunchGeom &g, KernelKind kind, const uint32_t *xmap, uint32_t xgrid,
                        bool xdirect, bool ranged, hipStream_t s) {
    switch (dtype) {
        case EMB_F32:
            return launch_lpr<IdxT, EMB_F32>(d, n, max_tiles, g, kind, xmap, xgrid, xdirect, ranged, s);
        case EMB_F16:
            return launch_lpr<IdxT, EMB_F16>(d, n, max_tiles, g, kind, xmap, xgrid, xdirect, ranged, s);
        case EMB_FIXED32:
            return launch_lpr<IdxT, EMB_FIXED32>(d, n, max_tiles, g, kind, xmap, xgrid, xdirect, ranged, s);
    }
    return hipErrorInvalidValue;
}

template <typename IdxT, int DT>
hipError_t launch_hot_lpr(const DevDesc *d, uint32_t n, uint32_t wgs, uint32_t lds, const LaunchGeom &g, hipStream_t s) {
    const dim3 grid(wgs, n, 1), block(HotCfg::kBlock);
    switch (g.lanes_per_row) {
#define PIMEMB_CASE(L)                                                                                       \
    case L:                                                                                                  \
        hipLaunchKernelGGL((bag_sum_hot_kernel<IdxT, DT, L, HotCfg>), grid, block, lds, s, d, g.chunks);     \
        break;
        PIMEMB_CASE(1)
        PIMEMB_CASE(2)
        PIMEMB_CASE(4)
        PIMEMB_CASE(8)
        PIMEMB_CASE(16)
        PIMEMB_CASE(32)
        PIMEMB_CASE(64)
#undef PIMEMB_CASE
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

template <typename IdxT>
hipError_t launch_hot_dtype(const DevDesc *d, uint32_t n, uint32_t wgs, uint32_t lds, emb_dtype dtype,
                            const LaunchGeom &g, hipStream_t s) {
    switch (dtype) {
        case EMB_F32: return launch_hot_lpr<IdxT, EMB_F32>(d, n, wgs, lds, g, s);
        case EMB_F16: return launch_hot_lpr<IdxT, EMB_F16>(d, n, wgs, lds, g, s);
        case EMB_FIXED32: return launch_hot_lpr<IdxT, EMB_FIXED32>(d, n, wgs, lds, g, s);
    }
    return hipErrorInvalidValue;
}

template <typename IdxT>
hipError_t launch_anydim(const DevDesc *d, uint32_t n, uint32_t max_tiles, emb_dtype dtype, const LaunchGeom &g,
                         hipStream_t s) {
    const dim3 grid(max_tiles, n, 1), block(256);
    if (g.anydim_vec) {
        switch (dtype) {
            case EMB_F32:
                hipLaunchKernelGGL((bag_sum_anydim_vec_kernel<IdxT, EMB_F32, kClampInputs>), grid, block, 0, s, d, g.chunks, g.scalar_lanes);
                break;
            case EMB_F16:
                hipLaunchKernelGGL((bag_sum_anydim_vec_kernel<IdxT, EMB_F16, kClampInputs>), grid, block, 0, s, d, g.chunks, g.scalar_lanes);
                break;
            case EMB_FIXED32:
                hipLaunchKernelGGL((bag_sum_anydim_vec_kernel<IdxT, EMB_FIXED32, kClampInputs>), grid, block, 0, s, d, g.chunks, g.scalar_lanes);
                break;
            default:
                return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    switch (dtype) {
        case EMB_F32:
            hipLaunchKernelGGL((bag_sum_anydim_kernel<IdxT, EMB_F32, kClampInputs>), grid, block, 0, s, d, g.chunks, g.scalar_lanes);
            break;
        case EMB_F16:
            hipLaunchKernelGGL((bag_sum_anydim_kernel<IdxT, EMB_F16, kClampInputs>), grid, block, 0, s, d, g.chunks, g.scalar_lanes);
            break;
        case EMB_FIXED32:
            hipLaunchKernelGGL((bag_sum_anydim_kernel<IdxT, EMB_FIXED32, kClampInputs>), grid, block, 0, s, d, g.chunks, g.scalar_lanes);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

// ---- column scatter for populate_mram-style uploads -----------------------------------------
__global__ void __launch_bounds__(kBlock)
scatter_column_kernel(int32_t *__restrict__ table, const int32_t *__restrict__ column,
                      uint64_t nr_rows, uint32_t dim, uint32_t col) {
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < nr_rows;
         r += (uint64_t)gridDim.x * kBlock)
        table[r * dim + col] = column[r];
}

// ---- input validation ------------------------------------------------------------------------
// Counts indices >= nr_rows and broken offsets of a batched call (grid: x = workgroups per descriptor, y = descriptor).
// HBM-bound integer work: 16-byte loads (two int64 / four uint32 ids per lane; element alignment is all the hardware asks of
// a global load), kValVecs of them in flight per lane before the first compare, ~1 000 workgroups for 26 x 39 292 ids.
// The kernel reports by itself, without a host-side event and without a second kernel: every workgroup signs off with what it
// found (pimemb_internal.h: two levels of tickets), and the LAST one holds the total,
// leaves the counters zeroed for the next call that gets this ValidateCtl, and writes validate_word(seq, total) into the
// pinned `result` word the host polls -- one 8-byte store, nothing to order it against.  With `poison` a workgroup that
// found something also zeroes n_tiles of EVERY descriptor of the launch image -- the guard every lookup kernel tests before it
// touches anything -- so the lookup kernels enqueued right behind this one on the same stream find no work (the host still
// learns the count and reports it; it just does not have to wait before enqueueing the lookup).
#ifndef PIMEMB_VALIDATE_EXPERIMENT
#define PIMEMB_VALIDATE_EXPERIMENT 0
#endif
#ifndef PIMEMB_VALIDATE_VECS
#define PIMEMB_VALIDATE_VECS 2
#endif
constexpr int kValVecs = PIMEMB_VALIDATE_VECS;              // 16-byte loads in flight per lane and array
template <typename IdxT> struct ValVec;
template <> struct ValVec<uint32_t> { typedef uint32_t T __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct ValVec<int64_t> { typedef int64_t T __attribute__((ext_vector_type(2), aligned(8))); };

template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
validate_kernel(DevDesc *__restrict__ descs, uint32_t n_descs, ValidateCtl *__restrict__ ctl,
                volatile unsigned long long *__restrict__ result, unsigned long long seq, int poison) {
    constexpr uint32_t kPer = 16 / sizeof(IdxT);
    using Vec = typename ValVec<IdxT>::T;
    __shared__ unsigned int s_found;
    if (threadIdx.x == 0) s_found = 0;
    __syncthreads();
    const DevDesc *dp = descs + blockIdx.y;
    const IdxT *indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *offsets = static_cast<const IdxT *>(dp->offsets);
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags, nr_rows = dp->nr_rows;
    uint32_t local = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock * kValVecs;            // in 16-byte vectors
    const uint64_t first = (uint64_t)blockIdx.x * kBlock * kValVecs + threadIdx.x;
    // ids: (uint64)v >= nr_rows also catches a negative int64
    const uint64_t id_vecs = n_idx / kPer;
    for (uint64_t v0 = first; v0 < id_vecs; v0 += stride) {
        Vec r[kValVecs];
#pragma unroll
        for (int j = 0; j < kValVecs; j++) {
            const uint64_t v = v0 + (uint64_t)j * kBlock;
            if (v < id_vecs) r[j] = *reinterpret_cast<const Vec *>(indices + v * kPer);
            else
                for (uint32_t k = 0; k < kPer; k++) r[j][k] = 0;
        }
#pragma unroll
        for (int j = 0; j < kValVecs; j++)
#pragma unroll
            for (uint32_t k = 0; k < kPer; k++) local += (uint64_t)r[j][k] >= nr_rows;
    }
    if (blockIdx.x == 0 && threadIdx.x < n_idx % kPer) local += (uint64_t)indices[id_vecs * kPer + threadIdx.x] >= nr_rows;
    if (offsets != nullptr) {
        // bag b is broken if offsets[b] > its end or its end > n_idx, its end = offsets[b + 1] (n_idx for the last bag);
        // as unsigned compares that covers negative int64 values on either side
        const uint64_t off_vecs = n_bags / kPer;
        for (uint64_t v0 = first; v0 < off_vecs; v0 += stride) {
            Vec r[kValVecs];
            uint64_t nxt[kValVecs];
#pragma unroll
            for (int j = 0; j < kValVecs; j++) {
                const uint64_t v = v0 + (uint64_t)j * kBlock;
                if (v < off_vecs) {
                    r[j] = *reinterpret_cast<const Vec *>(offsets + v * kPer);
                    nxt[j] = ((v + 1) * kPer < n_bags) ? (uint64_t)offsets[(v + 1) * kPer] : n_idx;
                } else {
                    for (uint32_t k = 0; k < kPer; k++) r[j][k] = 0;
                    nxt[j] = 0;
                }
            }
#pragma unroll
            for (int j = 0; j < kValVecs; j++)
#pragma unroll
                for (uint32_t k = 0; k < kPer; k++) {
                    const uint64_t end = (k + 1 < kPer) ? (uint64_t)r[j][k + 1 < kPer ? k + 1 : k] : nxt[j];
                    local += ((uint64_t)r[j][k] > end) | (end > n_idx);
                }
        }
        if (blockIdx.x == 0 && threadIdx.x < n_bags % kPer) {
            const uint64_t b = off_vecs * kPer + threadIdx.x;
            const uint64_t end = (b + 1 < n_bags) ? (uint64_t)offsets[b + 1] : n_idx;
            local += ((uint64_t)offsets[b] > end) | (end > n_idx);
        }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        if ((uint64_t)dp->fixed_pooling * n_bags != n_idx) local++;
    }
    if (local) atomicAdd(&s_found, local);            // (LDS; rare)
    __syncthreads();
    const uint32_t found = s_found;
    if (found && poison)
        for (uint32_t j = threadIdx.x; j < n_descs; j += kBlock) descs[j].n_tiles = 0;   // (a field this kernel never reads)
    if (threadIdx.x != 0) return;
#if PIMEMB_VALIDATE_EXPERIMENT == 1                   // timing floor: no tickets, workgroup (0, 0) reports at once (WRONG results)
    if (blockIdx.x == 0 && blockIdx.y == 0) *result = validate_word(seq, 0);
    return;
#endif
    // Sign off: ONE relaxed device-scope atomic per workgroup that carries both the ticket (low kValTicketBits bits) and what the
    // workgroup found (the bits above), so the count needs no ordering against the ticket -- it IS the ticket -- and whoever
    // completes a counter holds its total.  (An acquire / release pair on every ticket is a cache write-back and invalidate
    // apiece: the kernel at 24 us instead of 8.)  The poison stores need no ticket either: the lookup kernels are behind a
    // kernel boundary.
    const unsigned long long total_wgs = (unsigned long long)gridDim.x * gridDim.y;
    const unsigned long long w = (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x;
    unsigned long long mine = 1ull + ((unsigned long long)found << kValTicketBits), top_target = total_wgs;
    if (total_wgs > kValLanes) {
        const uint32_t lane = (uint32_t)(w % kValLanes);
        const unsigned long long lane_wgs = total_wgs / kValLanes + (lane < total_wgs % kValLanes ? 1ull : 0ull);
        const unsigned long long got = __hip_atomic_fetch_add(&ctl->sub[lane].done, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
        if ((got & kValTicketMask) != lane_wgs) return;
        mine = 1ull + (got & ~kValTicketMask);
        top_target = kValLanes;
    }
    const unsigned long long got = __hip_atomic_fetch_add(&ctl->tickets, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + mine;
    if ((got & kValTicketMask) != top_target) return;
    // the last one out: every other workgroup has made its last access to the counters (its sign-off IS that access), so THIS
    // thread zeroes them all -- its own stores, which its own fence below orders -- for whoever the host gives this ValidateCtl
    // next (it may be on another stream) ...
    if (total_wgs > kValLanes)
        for (uint32_t l = 0; l < kValLanes; l++) __hip_atomic_store(&ctl->sub[l].done, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&ctl->tickets, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __threadfence();
    *result = validate_word(seq, got >> kValTicketBits);   // ... before the host can know that this call is done with it
}

// ---- multi-GPU routing of variable-length BAGS to row-range shards -------------------------------
// A bag of a row-split table is cut into one SUB-BAG per shard that owns some of its rows; the shard
// returns one partial pooled row per sub-bag and the bag's owner adds the partials in shard order
// (emb_dpu_lookup.c:106-116 is the bag loop being split).  Counts first, payload second: pass 1 counts,
// per (table, shard, bag), the indices that fall into the shard; pass 2 turns the counts into slots and
// positions (exclusive scans over the bags) and the per-(shard, table) totals the peers need FIRST;
// pass 3 lays the request pieces out back to back; pass 4 writes them.  Everything is placed by prefix
// sums, never by atomics, so a sub-bag keeps its indices in bag order and the layout -- and with it the
// partial sums -- are the same on every run.
//
// One thread per (bag, shard), shard fastest: the 64 lanes of a wavefront read 64/n_shards bags'
// indices, each address broadcast to the n_shards lanes that test it against their own row range (no
// division, no cross-lane traffic, ragged bags welcome).  The index array is re-read n_shards times out
// of L1/L2; at B = 16384 x 32 indices x 8 tables x 8 shards that is 134 MB of cache reads, a few us.
constexpr uint32_t kNoSlot = 0xffffffffu;
constexpr int kScanBlock = 1024;

// Word offsets of the sections of `meta` (pimemb.h, emb_route_bags).  The counts message of destination d is
// counts[d][0 .. K] -- K table entries {n_sub, n_idx} and one trailing entry {peak request words, peak partial rows}
// that is the same for every d: the largest piece this rank sends / expects back from any ONE peer, so that every
// rank can derive the same number of transfer rounds from the counts it receives (sharding.py).
struct MetaLayout {
    uint32_t base, piece, row0, mode, words;
};
__host__ __device__ __forceinline__ MetaLayout meta_layout(uint32_t n_shards, uint32_t n_tables) {
    MetaLayout m;
    m.base = 2 * n_shards * (n_tables + 1);
    m.piece = m.base + 2 * n_shards * n_tables;
    m.row0 = m.piece + n_shards + 1;
    m.mode = m.row0 + n_shards * n_tables;
    m.words = m.mode + 1;
    return m;
}
__host__ __device__ __forceinline__ uint32_t counts_at(uint32_t d, uint32_t k, uint32_t n_tables) {
    return (d * (n_tables + 1) + k) * 2;
}
constexpr uint32_t kModeBags = 0, kModeOneHot = 1;   // meta[mode]: how `slots` is encoded (see unroute_bags_kernel)

struct RouteBagTable {
    const void *indices;      // uint32 or int64 (RouteBagParams::idx64), like the offsets
    const void *offsets;
    uint64_t n_indices;
    uint32_t fixed_pooling;
    uint32_t rows_per_shard;
};
struct RouteBagParams {
    RouteBagTable t[kRouteBagMaxTables];
    uint32_t idx64;           // the arrays of every table are int64 (torch's width) instead of uint32 (the reference's)
};

// one entry of an index / offset array as an unsigned 64-bit id (a negative int64 id: huge, beyond every range)
__device__ __forceinline__ uint64_t route_word(const void *a, uint32_t idx64, uint64_t i) {
    return idx64 ? (uint64_t)static_cast<const int64_t *>(a)[i] : (uint64_t)static_cast<const uint32_t *>(a)[i];
}

__device__ __forceinline__ void bag_bounds(const RouteBagTable &t, uint32_t idx64, uint64_t b, uint64_t n_bags, uint64_t *p, uint64_t *e) {
    if (t.offsets != nullptr) {
        *p = route_word(t.offsets, idx64, b);
        *e = (b + 1 < n_bags) ? route_word(t.offsets, idx64, b + 1) : t.n_indices;
    } else {
        *p = b * t.fixed_pooling;
        *e = *p + t.fixed_pooling;
    }
    if (*e > t.n_indices) *e = t.n_indices;      // malformed offsets never read past the index array
    if (*p > *e) *p = *e;
}

// shard d owns rows [lo, hi); the LAST shard also takes anything beyond (an out-of-range index stays in bounds of the
// routing structures; the lookup kernel sees it as a local row id like any other -- one no table of < 2^32 rows holds
// when it does not fit 32 bits: route_local)
struct ShardRange {
    uint64_t lo, hi;
    bool last;
    __device__ __forceinline__ bool holds(uint64_t r) const { return r >= lo && (last || r < hi); }
};
__device__ __forceinline__ ShardRange shard_range(uint32_t rps, uint32_t d, uint32_t n_shards) {
    ShardRange s;
    s.lo = (uint64_t)d * rps;
    s.hi = s.lo + rps;
    s.last = d + 1 == n_shards;
    return s;
}
__device__ __forceinline__ uint32_t route_local(uint64_t r, uint64_t lo) {
    const uint64_t v = r - lo;
    return v > 0xffffffffull ? 0xffffffffu : (uint32_t)v;      // (never wraps into range)
}

// f(id) for the indices [p, e) of one table, in order.  uint32 arrays: up to the next 16-byte boundary one by one, then four
// per load; int64 arrays: one by one (8 bytes each, consecutive: the same lines).
template <class F>
__device__ __forceinline__ void route_for_each(const RouteBagTable &t, uint32_t idx64, uint64_t p, uint64_t e, F &&f) {
    if (idx64) {           // two ids per 16-byte load (torch's arrays are 8-byte aligned: up to the next 16-byte boundary one by one)
        const int64_t *ix = static_cast<const int64_t *>(t.indices);
        typedef int64_t i64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
        for (; p < e && (p & 1u); p++) f((uint64_t)ix[p]);
        for (; p + 2 <= e; p += 2) {
            const i64x2_a8 r2 = *reinterpret_cast<const i64x2_a8 *>(ix + p);
            f((uint64_t)r2[0]);
            f((uint64_t)r2[1]);
        }
        for (; p < e; p++) f((uint64_t)ix[p]);
        return;
    }
    const uint32_t *ix = static_cast<const uint32_t *>(t.indices);
    for (; p < e && (p & 3u); p++) f((uint64_t)ix[p]);
    for (; p + 4 <= e; p += 4) {
        const u32x4_a4 r4 = *reinterpret_cast<const u32x4_a4 *>(ix + p);   // (4-byte aligned type: the caller's array may start anywhere)
#pragma unroll
        for (int q = 0; q < 4; q++) f((uint64_t)r4[q]);
    }
    for (; p < e; p++) f((uint64_t)ix[p]);
}

__global__ void __launch_bounds__(kBlock)
route_bags_count_kernel(RouteBagParams rp, uint64_t n_bags, uint32_t n_shards, uint32_t *__restrict__ work) {
    const uint32_t k = blockIdx.y;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t b = gid / n_shards;
    const uint32_t d = (uint32_t)(gid % n_shards);
    if (b >= n_bags) return;
    const RouteBagTable &t = rp.t[k];
    uint64_t p, e;
    bag_bounds(t, rp.idx64, b, n_bags, &p, &e);
    const ShardRange sr = shard_range(t.rows_per_shard, d, n_shards);
    uint32_t c = 0;
    route_for_each(t, rp.idx64, p, e, [&](uint64_t r) { c += sr.holds(r) ? 1u : 0u; });
    work[((uint64_t)k * n_shards + d) * n_bags + b] = c;
}

// One workgroup per (shard, table): exclusive scans over the bags of "has a sub-bag" (-> slot) and of the index
// count (-> position of the sub-bag's first index); the totals are the counts message.  A thread owns 16 consecutive
// bags and fetches them with four 16-byte loads issued together (4-byte aligned: the row of `work` starts wherever
// n_bags puts it), so a chunk of 16384 bags costs one memory round trip, one LDS scan of the 1024 per-thread sums
// and one round of stores -- the first version walked its bags one dependent load at a time: 34 us for this kernel.
constexpr uint32_t kScanPerThread = 16;
constexpr uint32_t kScanChunk = kScanBlock * kScanPerThread;

__global__ void __launch_bounds__(kScanBlock)
route_bags_scan_kernel(uint64_t n_bags, uint32_t n_tables, uint32_t *__restrict__ work, uint32_t *__restrict__ slots,
                       uint32_t *__restrict__ counts) {
    __shared__ uint32_t s_sub[kScanBlock], s_idx[kScanBlock];
    const uint32_t d = blockIdx.x, k = blockIdx.y, n_shards = gridDim.x;
    uint32_t *w = work + ((uint64_t)k * n_shards + d) * n_bags;
    uint32_t *sl = slots + ((uint64_t)k * n_shards + d) * n_bags;
    uint32_t carry_sub = 0, carry_idx = 0;                      // totals of the chunks before this one
    for (uint64_t c0 = 0; c0 < n_bags; c0 += kScanChunk) {
        const uint64_t b0 = c0 + (uint64_t)threadIdx.x * kScanPerThread;
        uint32_t v[kScanPerThread];
        if (b0 + kScanPerThread <= n_bags) {
#pragma unroll
            for (uint32_t q = 0; q < kScanPerThread; q += 4) {
                const u32x4_a4 x = *reinterpret_cast<const u32x4_a4 *>(w + b0 + q);
                v[q] = x[0]; v[q + 1] = x[1]; v[q + 2] = x[2]; v[q + 3] = x[3];
            }
        } else {
#pragma unroll
            for (uint32_t q = 0; q < kScanPerThread; q++) v[q] = (b0 + q < n_bags) ? w[b0 + q] : 0u;
        }
        uint32_t sub = 0, idx = 0;
#pragma unroll
        for (uint32_t q = 0; q < kScanPerThread; q++) {
            sub += v[q] ? 1u : 0u;
            idx += v[q];
        }
        __syncthreads();                                        // (previous chunk's readers of s_* are done)
        s_sub[threadIdx.x] = sub;
        s_idx[threadIdx.x] = idx;
        __syncthreads();
        for (uint32_t step = 1; step < kScanBlock; step <<= 1) {      // inclusive Hillis-Steele scan of both sums
            uint32_t a = 0, c = 0;
            if (threadIdx.x >= step) {
                a = s_sub[threadIdx.x - step];
                c = s_idx[threadIdx.x - step];
            }
            __syncthreads();
            s_sub[threadIdx.x] += a;
            s_idx[threadIdx.x] += c;
            __syncthreads();
        }
        uint32_t run_sub = carry_sub + s_sub[threadIdx.x] - sub, run_idx = carry_idx + s_idx[threadIdx.x] - idx;   // exclusive
        uint32_t o_slot[kScanPerThread], o_pos[kScanPerThread];
#pragma unroll
        for (uint32_t q = 0; q < kScanPerThread; q++) {
            o_slot[q] = v[q] ? run_sub : kNoSlot;
            o_pos[q] = run_idx;
            run_sub += v[q] ? 1u : 0u;
            run_idx += v[q];
        }
        if (b0 + kScanPerThread <= n_bags) {
#pragma unroll
            for (uint32_t q = 0; q < kScanPerThread; q += 4) {
                *reinterpret_cast<u32x4_a4 *>(sl + b0 + q) = u32x4{o_slot[q], o_slot[q + 1], o_slot[q + 2], o_slot[q + 3]};
                *reinterpret_cast<u32x4_a4 *>(w + b0 + q) = u32x4{o_pos[q], o_pos[q + 1], o_pos[q + 2], o_pos[q + 3]};
            }
        } else {
#pragma unroll
            for (uint32_t q = 0; q < kScanPerThread; q++)
                if (b0 + q < n_bags) {
                    sl[b0 + q] = o_slot[q];
                    w[b0 + q] = o_pos[q];
                }
        }
        carry_sub += s_sub[kScanBlock - 1];
        carry_idx += s_idx[kScanBlock - 1];
    }
    if (threadIdx.x == 0) {
        counts[counts_at(d, k, n_tables) + 0] = carry_sub;
        counts[counts_at(d, k, n_tables) + 1] = carry_idx;
    }
}

__host__ __device__ __forceinline__ uint32_t pad4(uint32_t v) { return (v + 3u) & ~3u; }

// meta words: counts[N][K+1][2] | base[N][K][2] | piece[N+1] | ret_row0[N][K] | mode   (see pimemb.h, emb_route_bags).
// Exclusive prefix over the N*K (shard, table) entries, kBlock entries per pass: every thread loads its entry's
// counts (in parallel -- one lane walking the list paid a dependent load per entry: 11 us for 64 entries), an LDS scan
// gives the word / row offsets, a carry links the passes.  Then the peaks: the largest request piece (words) and the
// largest number of partial rows exchanged with any one peer, copied into every destination's counts message.
__device__ __forceinline__ void layout_meta(uint32_t n_shards, uint32_t n_tables, uint32_t *__restrict__ meta, uint32_t mode) {
    __shared__ uint32_t s_w[kBlock], s_r[kBlock], s_peak[2];
    const uint32_t nk = n_shards * n_tables;
    const MetaLayout ml = meta_layout(n_shards, n_tables);
    uint32_t *counts = meta;
    uint32_t *base = meta + ml.base, *piece = meta + ml.piece, *ret_row0 = meta + ml.row0;
    uint32_t carry_w = 0, carry_r = 0;
    if (threadIdx.x < 2) s_peak[threadIdx.x] = 0;
    for (uint32_t e0 = 0; e0 < nk; e0 += kBlock) {
        const uint32_t e = e0 + threadIdx.x;
        uint32_t n_sub = 0, n_idx = 0;
        if (e < nk) {
            n_sub = counts[counts_at(e / n_tables, e % n_tables, n_tables)];
            n_idx = counts[counts_at(e / n_tables, e % n_tables, n_tables) + 1];
        }
        const uint32_t words = pad4(n_sub) + pad4(n_idx);
        __syncthreads();
        s_w[threadIdx.x] = words;
        s_r[threadIdx.x] = n_sub;
        __syncthreads();
        for (uint32_t step = 1; step < kBlock; step <<= 1) {
            uint32_t a = 0, c = 0;
            if (threadIdx.x >= step) {
                a = s_w[threadIdx.x - step];
                c = s_r[threadIdx.x - step];
            }
            __syncthreads();
            s_w[threadIdx.x] += a;
            s_r[threadIdx.x] += c;
            __syncthreads();
        }
        if (e < nk) {
            const uint32_t w0 = carry_w + s_w[threadIdx.x] - words;       // exclusive
            base[2 * e] = w0;
            base[2 * e + 1] = w0 + pad4(n_sub);
            ret_row0[e] = carry_r + s_r[threadIdx.x] - n_sub;
            if (e % n_tables == 0) piece[e / n_tables] = w0;               // first table of destination d
        }
        carry_w += s_w[kBlock - 1];
        carry_r += s_r[kBlock - 1];
    }
    if (threadIdx.x == 0) {
        piece[n_shards] = carry_w;
        meta[ml.mode] = mode;
    }
    __syncthreads();                                     // piece[] and ret_row0[] of this workgroup are visible
    for (uint32_t d = threadIdx.x; d < n_shards; d += kBlock) {
        const uint32_t words = piece[d + 1] - piece[d];
        const uint32_t rows = (d + 1 < n_shards ? ret_row0[(d + 1) * n_tables] : carry_r) - ret_row0[d * n_tables];
        atomicMax(&s_peak[0], words);
        atomicMax(&s_peak[1], rows);
    }
    __syncthreads();
    for (uint32_t d = threadIdx.x; d < n_shards; d += kBlock) {
        counts[counts_at(d, n_tables, n_tables)] = s_peak[0];
        counts[counts_at(d, n_tables, n_tables) + 1] = s_peak[1];
    }
}

__global__ void __launch_bounds__(kBlock)
route_bags_layout_kernel(uint32_t n_shards, uint32_t n_tables, uint32_t *__restrict__ meta) {
    layout_meta(n_shards, n_tables, meta, kModeBags);
}

__global__ void __launch_bounds__(kBlock)
route_bags_place_kernel(RouteBagParams rp, uint64_t n_bags, uint32_t n_shards, uint32_t n_tables,
                        const uint32_t *__restrict__ work, const uint32_t *__restrict__ slots,
                        const uint32_t *__restrict__ meta, uint32_t *__restrict__ send) {
    const uint32_t k = blockIdx.y;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t b = gid / n_shards;
    const uint32_t d = (uint32_t)(gid % n_shards);
    if (b >= n_bags) return;
    const uint64_t at = ((uint64_t)k * n_shards + d) * n_bags + b;
    const uint32_t slot = slots[at];
    if (slot == kNoSlot) return;
    const RouteBagTable &t = rp.t[k];
    const uint32_t *base = meta + meta_layout(n_shards, n_tables).base + 2 * (d * n_tables + k);
    uint32_t pos = work[at];
    send[base[0] + slot] = pos;                       // the sub-bag's start in the shard's index list (bag-start offsets)
    uint32_t *list = send + base[1];
    uint64_t p, e;
    bag_bounds(t, rp.idx64, b, n_bags, &p, &e);
    const ShardRange sr = shard_range(t.rows_per_shard, d, n_shards);
    route_for_each(t, rp.idx64, p, e, [&](uint64_t r) {          // kept in bag order
        if (sr.holds(r)) list[pos++] = route_local(r, sr.lo);
    });
}

// ---- one index per bag (the Criteo shape): every bag lives in exactly ONE shard --------------------------------
// Same request format, same slot order (a bag's slot = the number of earlier bags of its table that go to the same
// shard), same counts-first rule -- but each index is read once instead of n_shards times, and a bag costs one slot
// word instead of n_shards:
//   count : one thread per bag; dest = idx / rows_per_shard; the bag's rank among the bags of its 1024-bag block that
//           share its destination comes from wavefront ballots + an LDS prefix over the 16 wavefronts (no atomics);
//           per-(block, shard) totals go to `work`, (dest, rank in block) to `slots`;
//   layout: ONE workgroup turns the block totals into exclusive prefixes, the per-(shard, table) totals into the
//           counts message and the request layout (layout_meta);
//   place : one thread per bag writes its request (offset entry = slot: sub-bag s starts at index s; local row id)
//           and replaces its slot word by (dest << 24) | slot.
// Eligible when every table has fixed_pooling == 1 and no offsets array, n_shards <= 64 and n_bags < 2^24.
constexpr int kOneHotBlock = 1024;
constexpr uint32_t kOneHotMaxShards = 64;

// the shard of a one-index bag: id / rows_per_shard; an out-of-range id (beyond the table, a negative int64) goes to the last
// shard, as ShardRange::holds does (a 32-bit division unless the id needs more)
__device__ __forceinline__ uint32_t onehot_dest(uint64_t id, uint32_t rps, uint32_t n_shards) {
    const uint64_t d = (id >> 32) ? id / rps : (uint64_t)((uint32_t)id / rps);
    return d >= n_shards ? n_shards - 1 : (uint32_t)d;
}

__global__ void __launch_bounds__(kOneHotBlock)
route_onehot_count_kernel(RouteBagParams rp, uint64_t n_bags, uint32_t n_shards, uint32_t n_blocks,
                          uint32_t *__restrict__ packed, uint32_t *__restrict__ blockcnt) {
    __shared__ uint32_t wcnt[kOneHotBlock / 64][kOneHotMaxShards];   // per wavefront / shard, then exclusive prefix
    const uint32_t k = blockIdx.y, blk = blockIdx.x;
    const uint64_t b = (uint64_t)blk * kOneHotBlock + threadIdx.x;
    const bool live = b < n_bags;
    const RouteBagTable &t = rp.t[k];
    uint32_t dest = 0;
    if (live) dest = onehot_dest(route_word(t.indices, rp.idx64, b), t.rows_per_shard, n_shards);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t rank = 0;
    for (uint32_t d = 0; d < n_shards; d++) {            // wave-uniform trip count
        const unsigned long long m = __ballot(live && dest == d);
        if (lane == 0) wcnt[wave][d] = (uint32_t)__popcll(m);
        if (live && dest == d) rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    }
    __syncthreads();
    if (threadIdx.x < n_shards) {
        const uint32_t d = threadIdx.x;
        uint32_t total = 0;
        for (uint32_t w = 0; w < kOneHotBlock / 64; w++) {
            const uint32_t c = wcnt[w][d];
            wcnt[w][d] = total;
            total += c;
        }
        blockcnt[((uint64_t)k * n_blocks + blk) * n_shards + d] = total;
    }
    __syncthreads();
    if (live) packed[(uint64_t)k * n_bags + b] = (dest << 24) | (wcnt[wave][dest] + rank);
}

__global__ void __launch_bounds__(kBlock)
route_onehot_layout_kernel(uint32_t n_shards, uint32_t n_tables, uint32_t n_blocks, uint32_t *__restrict__ blockcnt,
                           uint32_t *__restrict__ meta) {
    const uint32_t nk = n_shards * n_tables;
    for (uint32_t e = threadIdx.x; e < nk; e += kBlock) {      // block totals -> exclusive prefix; the sum is the count
        const uint32_t d = e / n_tables, k = e % n_tables;
        uint32_t run = 0;
        for (uint32_t blk = 0; blk < n_blocks; blk++) {
            uint32_t *p = blockcnt + ((uint64_t)k * n_blocks + blk) * n_shards + d;
            const uint32_t c = *p;
            *p = run;
            run += c;
        }
        meta[counts_at(d, k, n_tables)] = run;                   // one index per sub-bag: n_sub == n_idx
        meta[counts_at(d, k, n_tables) + 1] = run;
    }
    __syncthreads();
    layout_meta(n_shards, n_tables, meta, kModeOneHot);
}

__global__ void __launch_bounds__(kOneHotBlock)
route_onehot_place_kernel(RouteBagParams rp, uint64_t n_bags, uint32_t n_shards, uint32_t n_tables, uint32_t n_blocks,
                          uint32_t *__restrict__ packed, const uint32_t *__restrict__ blockcnt,
                          const uint32_t *__restrict__ meta, uint32_t *__restrict__ send) {
    const uint32_t k = blockIdx.y, blk = blockIdx.x;
    const uint64_t b = (uint64_t)blk * kOneHotBlock + threadIdx.x;
    if (b >= n_bags) return;
    const RouteBagTable &t = rp.t[k];
    const uint32_t pk = packed[(uint64_t)k * n_bags + b];
    const uint32_t dest = pk >> 24;
    const uint32_t slot = blockcnt[((uint64_t)k * n_blocks + blk) * n_shards + dest] + (pk & 0xffffffu);
    const uint32_t *base = meta + meta_layout(n_shards, n_tables).base + 2 * (dest * n_tables + k);
    send[base[0] + slot] = slot;                                          // sub-bag `slot` starts at index `slot`
    send[base[1] + slot] = route_local(route_word(t.indices, rp.idx64, b), (uint64_t)dest * t.rows_per_shard);   // local row id
    packed[(uint64_t)k * n_bags + b] = (dest << 24) | slot;
}

// Layout and placement in ONE kernel (shapes with n_shards * n_tables <= 1024 entries and <= 16 block totals per
// thread -- every BASELINE shape): each workgroup re-derives, from the block totals of the count kernel, what it needs --
// the totals of every (shard, table) entry (LDS adds), the prefix over the entries (request layout), the totals of the
// blocks of its own table before it -- and places its 1024 bags; workgroup (0, 0) also writes `meta`.  The totals are a
// few KB out of L2, so the redundancy costs ~1 us per workgroup, all of them in parallel, and saves a launch and the
// one-workgroup kernel in the middle (11.8 -> 8 us for the router at the C4 shape).
constexpr uint32_t kFusedMaxEntries = 1024;

__global__ void __launch_bounds__(kOneHotBlock)
route_onehot_place_fused_kernel(RouteBagParams rp, uint64_t n_bags, uint32_t n_shards, uint32_t n_tables, uint32_t n_blocks,
                                uint32_t *__restrict__ packed, const uint32_t *__restrict__ blockcnt,
                                uint32_t *__restrict__ meta, uint32_t *__restrict__ send) {
    __shared__ uint32_t s_tot[kFusedMaxEntries], s_w[kFusedMaxEntries], s_r[kFusedMaxEntries];
    __shared__ uint32_t s_before[kOneHotMaxShards], s_peak[2];
    const uint32_t k = blockIdx.y, blk = blockIdx.x, tid = threadIdx.x;
    const uint32_t nk = n_shards * n_tables;
    if (tid < nk) s_tot[tid] = 0;
    if (tid < n_shards) s_before[tid] = 0;
    if (tid < 2) s_peak[tid] = 0;
    __syncthreads();
    const uint32_t per_table = n_blocks * n_shards, n_words = n_tables * per_table;
    for (uint32_t w = tid; w < n_words; w += kOneHotBlock) {          // blockcnt[(k' * n_blocks + b') * n_shards + d]
        const uint32_t kk = w / per_table, rem = w - kk * per_table, bb = rem / n_shards, d = rem - bb * n_shards;
        const uint32_t c = blockcnt[w];
        if (c) {
            atomicAdd(&s_tot[d * n_tables + kk], c);
            if (kk == k && bb < blk) atomicAdd(&s_before[d], c);
        }
    }
    __syncthreads();
    // inclusive scan over the entries e = d * K + k' (the order of the request layout) of words and partial rows
    uint32_t n = 0, words = 0;
    if (tid < nk) {
        n = s_tot[tid];
        words = 2u * pad4(n);                   // offsets[pad4(n)] + row ids[pad4(n)]: one index per sub-bag
    }
    s_w[tid] = words;
    s_r[tid] = n;
    __syncthreads();
    for (uint32_t step = 1; step < nk; step <<= 1) {
        uint32_t a = 0, c = 0;
        if (tid >= step) {
            a = s_w[tid - step];
            c = s_r[tid - step];
        }
        __syncthreads();
        s_w[tid] += a;
        s_r[tid] += c;
        __syncthreads();
    }
    const MetaLayout ml = meta_layout(n_shards, n_tables);
    if (blk == 0 && k == 0) {                   // one workgroup publishes the layout (the host, the peers, the un-router)
        const uint32_t total_w = s_w[nk - 1], total_r = s_r[nk - 1];
        if (tid < nk) {
            const uint32_t d = tid / n_tables, kk = tid - d * n_tables;
            const uint32_t w0 = s_w[tid] - words;
            meta[counts_at(d, kk, n_tables)] = n;
            meta[counts_at(d, kk, n_tables) + 1] = n;
            meta[ml.base + 2 * tid] = w0;
            meta[ml.base + 2 * tid + 1] = w0 + pad4(n);
            meta[ml.row0 + tid] = s_r[tid] - n;
            if (kk == 0) meta[ml.piece + d] = w0;
        }
        if (tid < n_shards) {
            const uint32_t first = tid * n_tables, next = first + n_tables;
            const uint32_t w_lo = s_w[first] - 2u * pad4(s_tot[first]), r_lo = s_r[first] - s_tot[first];
            const uint32_t w_hi = next < nk ? s_w[next] - 2u * pad4(s_tot[next]) : total_w;
            const uint32_t r_hi = next < nk ? s_r[next] - s_tot[next] : total_r;
            atomicMax(&s_peak[0], w_hi - w_lo);
            atomicMax(&s_peak[1], r_hi - r_lo);
        }
        if (tid == 0) {
            meta[ml.piece + n_shards] = total_w;
            meta[ml.mode] = kModeOneHot;
        }
        __syncthreads();
        if (tid < n_shards) {
            meta[counts_at(tid, n_tables, n_tables)] = s_peak[0];
            meta[counts_at(tid, n_tables, n_tables) + 1] = s_peak[1];
        }
    }
    const uint64_t b = (uint64_t)blk * kOneHotBlock + tid;
    if (b >= n_bags) return;
    const RouteBagTable &t = rp.t[k];
    const uint32_t pk = packed[(uint64_t)k * n_bags + b];
    const uint32_t dest = pk >> 24, e = dest * n_tables + k;
    const uint32_t slot = s_before[dest] + (pk & 0xffffffu);
    const uint32_t cnt = s_tot[e], w0 = s_w[e] - 2u * pad4(cnt);
    send[w0 + slot] = slot;                                              // sub-bag `slot` starts at index `slot`
    send[w0 + pad4(cnt) + slot] = route_local(route_word(t.indices, rp.idx64, b), (uint64_t)dest * t.rows_per_shard);   // local row id
    packed[(uint64_t)k * n_bags + b] = (dest << 24) | slot;
}

// pooled[k][b][:] = sum over the shards d = 0 .. N-1 that served a sub-bag of bag b, IN THAT ORDER, of the partial
// row that came back: deterministic, and exact for bags that live in one shard (one-hot lookups).
// `slots` as emb_route_bags left it: meta[mode] == kModeBags: uint32[K][N][n_bags], slot of bag b's sub-bag in
// (d, k)'s request or kNoSlot; kModeOneHot: uint32[K][n_bags], (dest << 24) | slot -- one word and one row per bag.
struct UnrouteOut {
    float *p[kRouteBagMaxTables];     // pooled rows of table k: float[n_bags][dim]
};

__global__ void __launch_bounds__(kBlock)
unroute_bags_kernel(const float *__restrict__ recv, const uint32_t *__restrict__ meta, const uint32_t *__restrict__ slots,
                    uint64_t n_bags, uint32_t n_shards, uint32_t dim, UnrouteOut outs) {
    const uint32_t k = blockIdx.y, n_tables = gridDim.y;
    float *__restrict__ pooled_k = outs.p[k];
    const uint32_t pieces = dim / 4;
    const uint64_t gid = (uint64_t)blockIdx.x * kBlock + threadIdx.x;
    const uint64_t b = gid / pieces;
    const uint32_t piece = (uint32_t)(gid % pieces);
    if (b >= n_bags) return;
    const MetaLayout ml = meta_layout(n_shards, n_tables);
    const uint32_t *ret_row0 = meta + ml.row0;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (meta[ml.mode] == kModeOneHot) {
        const uint32_t pk = slots[(uint64_t)k * n_bags + b];
        const uint64_t row = (uint64_t)ret_row0[(pk >> 24) * n_tables + k] + (pk & 0xffffffu);
        acc += *(reinterpret_cast<const f32x4 *>(recv) + row * pieces + piece);
        __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(pooled_k + b * dim) + piece);
        return;
    }
    // eight shards at a time: their slot words are fetched together, then the partial rows that exist, then the adds in
    // shard order (a slot -> row -> add chain per shard paid up to 2 x n_shards dependent round trips per bag: 37 -> 23 us
    // at one index per bag, 8 tables x 16384 bags x 8 shards)
    for (uint32_t d0 = 0; d0 < n_shards; d0 += 8) {
        uint32_t slot[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            slot[j] = (d0 + j < n_shards) ? slots[((uint64_t)k * n_shards + d0 + j) * n_bags + b] : kNoSlot;
        f32x4 v[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (slot[j] != kNoSlot) {
                const uint64_t row = (uint64_t)ret_row0[(d0 + j) * n_tables + k] + slot[j];
                v[j] = *(reinterpret_cast<const f32x4 *>(recv) + row * pieces + piece);
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < 8; j++)
            if (slot[j] != kNoSlot) acc += v[j];
    }
    __builtin_nontemporal_store(acc, reinterpret_cast<f32x4 *>(pooled_k + b * dim) + piece);
}

}  // namespace

bool route_bags_is_onehot(const RouteBagDesc *tables, uint32_t n_tables, uint64_t n_bags, uint32_t n_shards) {
    static const bool off = getenv("PIMEMB_ROUTE_ONEHOT") && getenv("PIMEMB_ROUTE_ONEHOT")[0] == '0';   // A/B switch (tools/route_probe.py, tests)
    if (off || n_shards > kOneHotMaxShards || n_bags >= (1ull << 24)) return false;
    for (uint32_t k = 0; k < n_tables; k++)
        if (tables[k].offsets != nullptr || tables[k].fixed_pooling != 1 || tables[k].n_indices != n_bags) return false;
    return true;
}

hipError_t launch_route_bags(const RouteBagDesc *tables, uint32_t n_tables, uint64_t n_bags, uint32_t n_shards,
                             uint32_t *send, uint32_t *meta, uint32_t *slots, uint32_t *work, hipStream_t stream, bool idx64) {
    if (n_tables == 0 || n_bags == 0 || n_tables > kRouteBagMaxTables) return hipErrorInvalidValue;
    RouteBagParams rp{};
    rp.idx64 = idx64 ? 1u : 0u;
    for (uint32_t k = 0; k < n_tables; k++)
        rp.t[k] = RouteBagTable{tables[k].indices, tables[k].offsets, tables[k].n_indices, tables[k].fixed_pooling,
                                tables[k].rows_per_shard};
    if (route_bags_is_onehot(tables, n_tables, n_bags, n_shards)) {
        const uint32_t n_blocks = (uint32_t)((n_bags + kOneHotBlock - 1) / kOneHotBlock);
        const dim3 grid(n_blocks, n_tables, 1);
        hipLaunchKernelGGL(route_onehot_count_kernel, grid, dim3(kOneHotBlock), 0, stream, rp, n_bags, n_shards, n_blocks,
                           slots, work);
        static const bool no_fuse = getenv("PIMEMB_ROUTE_FUSED") && getenv("PIMEMB_ROUTE_FUSED")[0] == '0';   // A/B switch
        if (!no_fuse && n_shards * n_tables <= kFusedMaxEntries &&
            (uint64_t)n_tables * n_blocks * n_shards <= 16ull * kOneHotBlock) {
            hipLaunchKernelGGL(route_onehot_place_fused_kernel, grid, dim3(kOneHotBlock), 0, stream, rp, n_bags, n_shards,
                               n_tables, n_blocks, slots, work, meta, send);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(route_onehot_layout_kernel, dim3(1), dim3(kBlock), 0, stream, n_shards, n_tables, n_blocks, work,
                           meta);
        hipLaunchKernelGGL(route_onehot_place_kernel, grid, dim3(kOneHotBlock), 0, stream, rp, n_bags, n_shards, n_tables,
                           n_blocks, slots, work, meta, send);
        return hipGetLastError();
    }
    const uint64_t threads = n_bags * n_shards;
    const dim3 grid((uint32_t)((threads + kBlock - 1) / kBlock), n_tables, 1);
    hipLaunchKernelGGL(route_bags_count_kernel, grid, dim3(kBlock), 0, stream, rp, n_bags, n_shards, work);
    hipLaunchKernelGGL(route_bags_scan_kernel, dim3(n_shards, n_tables, 1), dim3(kScanBlock), 0, stream, n_bags, n_tables,
                       work, slots, meta);
    hipLaunchKernelGGL(route_bags_layout_kernel, dim3(1), dim3(kBlock), 0, stream, n_shards, n_tables, meta);
    hipLaunchKernelGGL(route_bags_place_kernel, grid, dim3(kBlock), 0, stream, rp, n_bags, n_shards, n_tables, work, slots,
                       meta, send);
    return hipGetLastError();
}

uint32_t route_bags_meta_words(uint32_t n_tables, uint32_t n_shards) { return meta_layout(n_shards, n_tables).words; }
uint32_t route_meta_counts_words(uint32_t n_tables, uint32_t n_shards) { return meta_layout(n_shards, n_tables).base; }
uint32_t route_meta_piece_word(uint32_t n_tables, uint32_t n_shards) { return meta_layout(n_shards, n_tables).piece; }

hipError_t launch_unroute_bags_to(const float *recv, const uint32_t *meta, const uint32_t *slots, uint32_t n_tables,
                                  uint64_t n_bags, uint32_t n_shards, uint32_t dim, float *const *pooled_of_table,
                                  hipStream_t stream) {
    if (n_tables == 0 || n_bags == 0) return hipSuccess;
    if (n_tables > kRouteBagMaxTables) return hipErrorInvalidValue;
    UnrouteOut outs{};
    for (uint32_t k = 0; k < n_tables; k++) outs.p[k] = pooled_of_table[k];
    const uint64_t threads = n_bags * (dim / 4);
    const dim3 grid((uint32_t)((threads + kBlock - 1) / kBlock), n_tables, 1);
    hipLaunchKernelGGL(unroute_bags_kernel, grid, dim3(kBlock), 0, stream, recv, meta, slots, n_bags, n_shards, dim, outs);
    return hipGetLastError();
}

hipError_t launch_unroute_bags(const float *recv, const uint32_t *meta, const uint32_t *slots, uint32_t n_tables,
                               uint64_t n_bags, uint32_t n_shards, uint32_t dim, float *pooled, hipStream_t stream) {
    if (n_tables == 0 || n_bags == 0) return hipSuccess;
    if (n_tables > kRouteBagMaxTables) return hipErrorInvalidValue;
    float *ptrs[kRouteBagMaxTables];
    for (uint32_t k = 0; k < n_tables; k++) ptrs[k] = pooled + (uint64_t)k * n_bags * dim;
    return launch_unroute_bags_to(recv, meta, slots, n_tables, n_bags, n_shards, dim, ptrs, stream);
}

// Sharded step: carry up to three short word arrays (the counts a rank sent / received) from HBM into pinned host memory
// and raise a flag behind them, so the host learns them by polling one word -- no copy-engine hop, no event.
struct PublishSrc {
    const uint32_t *p[3];
    uint32_t n[3];
};
__global__ void __launch_bounds__(kBlock)
publish_words_kernel(PublishSrc src, uint32_t *__restrict__ dst_host, volatile unsigned long long *flag, unsigned long long value) {
    uint32_t at = 0;
    for (int j = 0; j < 3; j++) {         // a missing source leaves its section of dst_host alone (the host fills it)
        if (src.p[j] != nullptr)
            for (uint32_t i = threadIdx.x; i < src.n[j]; i += kBlock)
                dst_host[at + i] = __builtin_nontemporal_load(src.p[j] + i);
        at += src.n[j];
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        *flag = value;
    }
}

hipError_t launch_publish_words(const uint32_t *const src[3], const uint32_t n[3], uint32_t *dst_host,
                                unsigned long long *flag_host, unsigned long long value, hipStream_t stream) {
    PublishSrc ps{};
    for (int j = 0; j < 3; j++) {
        ps.p[j] = src[j];
        ps.n[j] = n[j];
    }
    hipLaunchKernelGGL(publish_words_kernel, dim3(1), dim3(kBlock), 0, stream, ps, dst_host, flag_host, value);
    return hipGetLastError();
}

// One thread: everything queued before it on the stream is done (kernel boundary) -> say so in a pinned word the host polls.
__global__ void store_word_kernel(volatile unsigned long long *dst, unsigned long long value) {
    __threadfence_system();
    *dst = value;
}
hipError_t launch_store_word(volatile unsigned long long *dst, unsigned long long value, hipStream_t stream) {
    hipLaunchKernelGGL(store_word_kernel, dim3(1), dim3(1), 0, stream, dst, value);
    return hipGetLastError();
}

// ---- collective-free exchange: the per-batch handshake through the job's shared segment (pimemb_peer.h) -----------------
// One workgroup.  For every remote destination p: the message = p's counts out of `meta` (+ where p's piece and p's partial
// rows sit), then this rank's host-written constants for p; all of it stored into mailbox [p][me][slot], a system-scope
// fence, then the `posted` word.  Runs on the caller's stream behind the router and behind whatever produced the caller's
// index buffers: a peer that sees `posted` may read both.
__global__ void __launch_bounds__(kBlock)
peer_post_kernel(const uint32_t *__restrict__ meta, uint32_t n_row_tables, uint32_t n_shards, PeerPostArgs a, unsigned long long value) {
    const uint32_t Kr = n_row_tables;
    const MetaLayout ml = meta_layout(n_shards, Kr ? Kr : 1);
    for (uint32_t p = 0; p < n_shards; p++) {
        if (a.box[p] == 0) continue;
        PeerMsg *box = reinterpret_cast<PeerMsg *>(a.box[p]);
        uint32_t at = 0;
        if (Kr) {
            const uint32_t n = 2 * (Kr + 1);
            // (meta == nullptr: nothing was routed for this batch -- the direct one-hot path: all counts zero)
            for (uint32_t i = threadIdx.x; i < n; i += kBlock) box->words[i] = meta ? meta[counts_at(p, 0, Kr) + i] : 0u;
            if (threadIdx.x == 0) {
                box->words[n] = meta ? meta[ml.piece + p] : 0u;
                box->words[n + 1] = meta ? meta[ml.row0 + p * Kr] : 0u;
            }
            at = n + 2;
        }
        for (uint32_t i = threadIdx.x; i < a.n_consts[p]; i += kBlock) box->words[at + i] = __builtin_nontemporal_load(a.consts[p] + i);
    }
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence_system();
        for (uint32_t p = 0; p < n_shards; p++)
            if (a.box[p] != 0) reinterpret_cast<PeerMsg *>(a.box[p])->posted = value;
    }
}

hipError_t launch_peer_post(const uint32_t *meta, uint32_t n_row_tables, uint32_t n_shards, const PeerPostArgs &args,
                            unsigned long long value, hipStream_t stream) {
    if (n_shards > 64) return hipErrorInvalidValue;
    hipLaunchKernelGGL(peer_post_kernel, dim3(1), dim3(kBlock), 0, stream, meta, n_row_tables, n_shards, args, value);
    return hipGetLastError();
}

// One thread, behind the fused lookup that stored pooled rows into the requesters' HBM: the kernel boundary has completed
// those stores; say so to every requester.
__global__ void peer_done_kernel(PeerDoneArgs a, unsigned long long value) {
    __threadfence_system();
    for (uint32_t i = 0; i < a.n; i++) reinterpret_cast<PeerMsg *>(a.box[i])->served = value;
}

hipError_t launch_peer_done(const PeerDoneArgs &args, unsigned long long value, hipStream_t stream) {
    if (args.n == 0) return hipSuccess;
    hipLaunchKernelGGL(peer_done_kernel, dim3(1), dim3(1), 0, stream, args, value);
    return hipGetLastError();
}

// Checked shards: the counters of a counted ranged launch -> their requesters (pimemb_peer.h: ServedArgs).  Wavefront w of
// the grid takes entries w, w + W, ... of the concatenated segments: lane l swaps lane-word l of the counter for zero, the
// wavefront adds the 64 values up, lane 0 stores (tag << 32) | sum as one 8-byte word.
__global__ void __launch_bounds__(kBlock)
served_counts_kernel(ServedArgs a) {
    static_assert(EMB_SERVED_LANES == 64, "one lane of a wavefront per lane of a counter");
    const uint32_t lane = threadIdx.x & 63u, wave = blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6), n_waves = gridDim.x * (kBlock / 64);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.n_flag) {        // "served": the lookup before this kernel has completed its stores
        __threadfence_system();
        for (uint32_t j = 0; j < a.n_flag; j++) *reinterpret_cast<volatile unsigned long long *>(a.flag[j]) = a.value[j];
    }
    uint32_t total = 0;
    for (uint32_t i = 0; i < a.n_seg; i++) total += a.seg[i].n_counted + a.seg[i].n_fill;
    for (uint32_t g = wave; g < total; g += n_waves) {
        uint32_t i = 0, c = g;
        while (c >= a.seg[i].n_counted + a.seg[i].n_fill) {
            c -= a.seg[i].n_counted + a.seg[i].n_fill;
            i++;
        }
        const ServedSeg sg = a.seg[i];
        uint32_t v = 0xffffffffu;
        if (c < sg.n_counted) {
            // (the launch before this one added with agent-scope atomics: read and zero the same way)
            v = __hip_atomic_exchange(sg.ctr + (size_t)c * (EMB_SERVED_BYTES / 4u) + (size_t)lane * (EMB_SERVED_STRIDE / 4u), 0u,
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += (uint32_t)__shfl_xor((int)v, off, 64);
        }
        if (lane == 0) *reinterpret_cast<volatile unsigned long long *>(sg.dst + c) = ((unsigned long long)sg.tag << 32) | v;
    }
}

hipError_t launch_served_counts(const ServedArgs &args, hipStream_t stream) {
    if (args.n_seg == 0 && args.n_flag == 0) return hipSuccess;
    if (args.n_seg > kServedSegs || args.n_flag > kServedSegs) return hipErrorInvalidValue;
    uint32_t n = 0;
    for (uint32_t i = 0; i < args.n_seg; i++) n += args.seg[i].n_counted + args.seg[i].n_fill;
    uint32_t grid = (n + (kBlock / 64) - 1) / (kBlock / 64);        // a wavefront per entry, up to 256 workgroups of 4
    grid = grid < 1 ? 1 : (grid > 256 ? 256 : grid);
    hipLaunchKernelGGL(served_counts_kernel, dim3(grid), dim3(kBlock), 0, stream, args);
    return hipGetLastError();
}

// Zero the {n_sub, n_idx} counts and peaks of a routing `meta` block (a rank with an empty batch still sends counts).
__global__ void __launch_bounds__(kBlock)
zero_words_kernel(uint32_t *__restrict__ p, uint32_t n) {
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) p[i] = 0u;
}
hipError_t launch_zero_words(uint32_t *p, uint32_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(zero_words_kernel, dim3(1), dim3(kBlock), 0, stream, p, n);
    return hipGetLastError();
}

// dst[i][:] = src[ids[i]][:] for n rows of row_bytes (a multiple of 16) each: the hot-row set of a table copied next to its hash
// in ONE launch (emb_set_hot_rows did one device-to-device copy per row: ~10 us each, a millisecond per table).
__global__ void __launch_bounds__(kBlock)
gather_rows_kernel(char *__restrict__ dst, const char *__restrict__ src, const unsigned long long *__restrict__ ids, uint32_t n, uint32_t row_bytes) {
    const uint32_t chunks = row_bytes / 16;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < (uint64_t)n * chunks; i += (uint64_t)gridDim.x * kBlock) {
        const uint32_t r = (uint32_t)(i / chunks), c = (uint32_t)(i % chunks);
        reinterpret_cast<f32x4 *>(dst + (uint64_t)r * row_bytes)[c] = reinterpret_cast<const f32x4 *>(src + ids[r] * row_bytes)[c];
    }
}
hipError_t launch_gather_rows(void *dst, const void *src, const unsigned long long *d_ids, uint32_t n, uint32_t row_bytes, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    if (row_bytes % 16) return hipErrorInvalidValue;
    uint64_t wgs = ((uint64_t)n * (row_bytes / 16) + kBlock - 1) / kBlock;
    wgs = wgs > 4096 ? 4096 : wgs;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((uint32_t)wgs), dim3(kBlock), 0, stream, static_cast<char *>(dst), static_cast<const char *>(src), d_ids, n, row_bytes);
    return hipGetLastError();
}

// uint32 words -> int64 words, segment by segment (blockIdx.y): the request pieces of an int64 job's routed step (uint32 local
// row ids and sub-bag starts, as they travel) widened once, so that they ride in the SAME launch as the job's int64 arrays
// instead of a small launch of their own (pimemb_shard.cpp, widen_pieces).
__global__ void __launch_bounds__(kBlock)
widen_words_kernel(WidenArgs a) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4), aligned(4)));
    typedef long long i64x2 __attribute__((ext_vector_type(2), aligned(8)));
    const uint32_t *__restrict__ src = a.src[blockIdx.y];
    long long *__restrict__ dst = a.dst[blockIdx.y];
    const uint64_t n = a.n[blockIdx.y], quads = n / 4;
    for (uint64_t q = (uint64_t)blockIdx.x * kBlock + threadIdx.x; q < quads; q += (uint64_t)gridDim.x * kBlock) {
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + 4 * q);
        i64x2 lo, hi;
        lo[0] = v[0]; lo[1] = v[1]; hi[0] = v[2]; hi[1] = v[3];
        *reinterpret_cast<i64x2 *>(dst + 4 * q) = lo;
        *reinterpret_cast<i64x2 *>(dst + 4 * q + 2) = hi;
    }
    if (blockIdx.x == 0 && threadIdx.x < n % 4) dst[quads * 4 + threadIdx.x] = src[quads * 4 + threadIdx.x];
}
hipError_t launch_widen_words(const WidenArgs &a, hipStream_t stream) {
    if (a.n_seg == 0) return hipSuccess;
    if (a.n_seg > kWidenSegs) return hipErrorInvalidValue;
    uint64_t most = 0;
    for (uint32_t i = 0; i < a.n_seg; i++) most = a.n[i] > most ? a.n[i] : most;
    if (most == 0) return hipSuccess;
    uint64_t wgs = (most / 4 + kBlock * 2 - 1) / (kBlock * 2);          // ~two 16-byte loads per lane
    wgs = wgs < 1 ? 1 : (wgs > 1024 ? 1024 : wgs);
    hipLaunchKernelGGL(widen_words_kernel, dim3((uint32_t)wgs, a.n_seg), dim3(kBlock), 0, stream, a);
    return hipGetLastError();
}

int geometry_for(emb_dtype dtype, uint32_t dim, LaunchGeom *g) {
    uint32_t elem = (dtype == EMB_F16) ? 2u : 4u;
    if (dtype != EMB_F32 && dtype != EMB_F16 && dtype != EMB_FIXED32) return EMB_ERR_INVALID;
    uint64_t row_bytes = (uint64_t)dim * elem;
    if (dim == 0) return EMB_ERR_UNSUPPORTED;
    g->anydim_vec = false;
    if (row_bytes % 16 != 0 || row_bytes > 1024) {   // no aligned 16-byte lane pieces: the any-dim kernels
        // a thread per 16-byte piece (rows of >= 32 B that are 4-byte multiples), else a thread per element
        // (fp32 one-hot, us per launch, piece vs element: dim 2 12.9 / 11.2, dim 6 19.9 / 25.5, dim 10 25.8 / 44.8,
        // dim 30 52.9 / 91.5; tools/anydim_probe.py)
        g->anydim_vec = row_bytes % 4 == 0 && row_bytes >= 32;
        const uint64_t units = g->anydim_vec ? (row_bytes + 15) / 16 : dim;
        uint32_t lanes = 1;
        while (lanes < units && lanes < 256u) lanes <<= 1;
        g->lanes_per_row = 0;
        g->chunks = dim;
        g->scalar_lanes = lanes;
        return EMB_OK;
    }
    g->scalar_lanes = 0;
    uint32_t chunks = (uint32_t)(row_bytes / 16);
    uint32_t lpr = 1;
    while (lpr < chunks) lpr <<= 1;
    g->lanes_per_row = lpr;
    g->chunks = chunks;
    return EMB_OK;
}

uint32_t bags_per_tile(KernelKind kind, const LaunchGeom &g) {
    if (kind == KERNEL_ANYDIM) return 256u / g.scalar_lanes;
    if (kind == KERNEL_HOT) return (64u / g.lanes_per_row) * (HotCfg::kBlock / 64);
    if (kind == KERNEL_WAVEBATCH) return 64u * WaveCfg::kBatches * (WaveCfg::kBlock / 64);
    if (kind == KERNEL_WAVEBATCH2) return 64u * Wave2Cfg::kBatches * (Wave2Cfg::kBlock / 64);
    return (64u / g.lanes_per_row) * (GroupCfg::kBlock / 64);
}

KernelKind choose_kernel(uint64_t total_bags, uint64_t total_indices, const LaunchGeom &g) {
    // Measured on MI355X (tools/tune_bag_kernels.hip, tools/tune_pooled.hip):
    //   * one-hot-ish launches (<= 2 indices per bag on average) that fill the chip with 64-bag
    //     wave batches (>= 2 wavefronts per SIMD) run 1.5x faster on the wave-batch kernel
    //     (coalesced bounds, all gathers of a batch in flight, speculative index prefetch);
    //   * pooled launches (dim 128, 32 indices per bag) run 1.3x faster on the lane-group kernel:
    //     a bag is already a long stream of gathers, and 64 bags per wavefront leave too few
    //     wavefronts to balance the machine.
    if (g.scalar_lanes) return KERNEL_ANYDIM;
    const bool one_hot_ish = total_indices <= 2 * total_bags;
    if (!one_hot_ish || total_bags / 64u < 2048u) return KERNEL_GROUP;
    // Two batches per wavefront once that still leaves >= 4 wavefronts per SIMD (>= 524288 bags) --
    // for NARROW rows only (<= 4 lanes per row, e.g. dim 16 fp32): a batch of wider rows already
    // keeps 8 gathers in flight per lane, and two of them would spill (dim 128: 5.0 vs 5.4 TB/s).
    return (g.lanes_per_row <= 4 && total_bags / 128u >= 4096u) ? KERNEL_WAVEBATCH2 : KERNEL_WAVEBATCH;
}

hipError_t launch_bag_sum(const DevDesc *d_descs, uint32_t n_descs, uint32_t max_tiles,
                          emb_dtype dtype, emb_index_type itype, const LaunchGeom &g,
                          KernelKind kind, const uint32_t *d_xmap, uint32_t xgrid, bool xdirect,
                          hipStream_t stream, bool ranged) {
    if (n_descs == 0 || max_tiles == 0) return hipSuccess;
    if (d_xmap == nullptr && n_descs > 65535u) return hipErrorInvalidValue;
    if (ranged && ((kind != KERNEL_WAVEBATCH && kind != KERNEL_WAVEBATCH2) || (kind == KERNEL_WAVEBATCH2 && g.lanes_per_row > 4)))
        return hipErrorInvalidValue;
    if (kind == KERNEL_ANYDIM) {
        if (g.scalar_lanes == 0 || d_xmap != nullptr) return hipErrorInvalidValue;
        return itype == EMB_IDX_U32 ? launch_anydim<uint32_t>(d_descs, n_descs, max_tiles, dtype, g, stream)
                                    : launch_anydim<int64_t>(d_descs, n_descs, max_tiles, dtype, g, stream);
    }
    if (itype == EMB_IDX_U32)
        return launch_dtype<uint32_t>(d_descs, n_descs, max_tiles, dtype, g, kind, d_xmap, xgrid, xdirect, ranged, stream);
    return launch_dtype<int64_t>(d_descs, n_descs, max_tiles, dtype, g, kind, d_xmap, xgrid, xdirect, ranged, stream);
}

hipError_t launch_bag_sum_hot(const DevDesc *d_descs, uint32_t n_descs, uint32_t wgs, uint32_t lds_bytes,
                              emb_dtype dtype, emb_index_type itype, const LaunchGeom &g, hipStream_t stream) {
    if (n_descs == 0 || wgs == 0) return hipSuccess;
    if (n_descs > 65535u || lds_bytes > kHotLdsBudget || g.scalar_lanes) return hipErrorInvalidValue;
    return itype == EMB_IDX_U32 ? launch_hot_dtype<uint32_t>(d_descs, n_descs, wgs, lds_bytes, dtype, g, stream)
                                : launch_hot_dtype<int64_t>(d_descs, n_descs, wgs, lds_bytes, dtype, g, stream);
}

hipError_t launch_scatter_column(int32_t *table, const int32_t *column, uint64_t nr_rows,
                                 uint32_t dim, uint32_t col, hipStream_t stream) {
    if (nr_rows == 0) return hipSuccess;
    uint64_t blocks = (nr_rows + kBlock - 1) / kBlock;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scatter_column_kernel, dim3((uint32_t)blocks), dim3(kBlock), 0, stream, table,
                       column, nr_rows, dim, col);
    return hipGetLastError();
}

uint32_t validate_workgroups(uint64_t max_items, emb_index_type itype) {
    // workgroups per descriptor: kValVecs 16-byte loads per lane and array, 1..1024 (a small call is a small kernel: in a
    // synchronous checked call the caller waits for it)
    const uint64_t per_wg = (uint64_t)kBlock * kValVecs * (itype == EMB_IDX_U32 ? 4 : 2);
    const uint64_t wgs = (max_items + per_wg - 1) / per_wg;
    return (uint32_t)(wgs < 1 ? 1 : (wgs > 1024 ? 1024 : wgs));
}

hipError_t launch_validate(DevDesc *d_descs, uint32_t n_descs, emb_index_type itype, ValidateCtl *ctl, unsigned long long *result,
                           unsigned long long seq, uint32_t wgs_per_desc, bool poison, hipStream_t stream) {
    if (n_descs == 0) return hipSuccess;
    dim3 grid(wgs_per_desc, n_descs, 1), block(kBlock, 1, 1);
    if (itype == EMB_IDX_U32)
        hipLaunchKernelGGL(validate_kernel<uint32_t>, grid, block, 0, stream, d_descs, n_descs, ctl, result, seq, poison ? 1 : 0);
    else
        hipLaunchKernelGGL(validate_kernel<int64_t>, grid, block, 0, stream, d_descs, n_descs, ctl, result, seq, poison ? 1 : 0);
    return hipGetLastError();
}

}  // namespace pimemb
