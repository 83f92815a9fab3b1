// pimemb_bag_kernels.h -- device code of the hot path (gfx950 / MI355X, CDNA4), as templates so the
// library (pimemb_kernels.hip) and the tuning tool (tools/tune_bag_kernels.hip) compile the very
// same source with different compile-time knobs.
//
// Replaces the UPMEM DPU program upmem/src/dpu/emb_dpu_lookup.c:36-138 (one DPU per (table,
// column), 14 tasklets striding over bags, one 8-byte MRAM read per (index, column)) and the host
// post-process upmem/include/emb_host.h:186-222 with ONE fused launch over all tables:
//
//   out[t][b][:] = sum_{p = off_t[b]}^{end-1} W_t[idx_t[p]][:]      end = off_t[b+1] | n_idx_t
//
// Mapping to the machine (bandwidth/latency-bound indexing; no MFMA -- there is no contraction):
//   * rows stay row-major [nr_rows][dim] in HBM; a row is read as 16-byte pieces, one per lane,
//     so the LPR = row_bytes/16 lanes of a "lane group" fetch a whole row with one coalesced
//     16-byte-per-lane load -- flat_load_dwordx4 in the shipped code object (PIMEMB_GLOBAL_AS 0, below) --
//     (64 B for dim 16 fp32, 512 B for dim 128 fp32);
//   * a 64-lane wavefront owns 64 consecutive bags per step ("wave batch"): lane l first loads the
//     bounds of bag l (two coalesced 256-B loads instead of 64/LPR scattered ones), then the
//     wavefront walks the batch in LPR rounds of 64/LPR bags, lane groups picking their bag's
//     bounds / index out of the owning lane with a wavefront shuffle (ds_bpermute);
//   * one-hot batches (every bag <= 1 index: the Criteo-Kaggle shape) take a fast path that
//     issues all rounds' row gathers before the first store, so a lane keeps up to 8 independent
//     16-byte gathers in flight; longer bags keep kUnroll gathers in flight inside the bag;
//   * every output element is accumulated by ONE lane in index order starting from +0, the order
//     of a sequential CPU EmbeddingBag: fp32 results are bit-identical to the oracle for any
//     pooling factor (no cross-lane reduction tree whose rounding would differ);
//   * the fixed-point mode keeps the reference arithmetic: int32 wrap-around accumulate
//     (emb_dpu_lookup.c:114) and out = (float)acc / 1e9 through double (emb_host.h:210).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "pimemb.h"

namespace pimemb {

// One table's share of a fused launch as the kernel sees it (HBM-resident array, 128 B each: a
// workgroup fetches its descriptor with scalar loads).
struct alignas(64) DevDesc {
    const void *weights;    // row-major [nr_rows][dim] of the table dtype
    const void *indices;    // IdxT[n_idx]
    const void *offsets;    // IdxT[n_bags] bag starts, or nullptr when fixed_pooling > 0
    float *out;             // float[n_bags][dim]
    uint64_t n_idx;
    uint64_t n_bags;
    uint64_t nr_rows;       // row ids are clamped to nr_rows-1 (kClamp); also read by the validation kernel
    uint32_t fixed_pooling; // L > 0: offsets[b] = b*L (load_generator.c:88)
    uint32_t n_tiles;       // ceil(n_bags / bags_per_tile) for this launch geometry
    // hot rows of this table (bag_sum_hot_kernel only; n_hot == 0 switches the LDS path off)
    const void *hot_rows;     // compact copy [n_hot][dim] of the hot rows, staged into LDS
    const uint64_t *hot_hash; // open-addressing table: entry = (slot << 32) | row id, empty = ~0
    uint32_t n_hot;
    uint32_t hot_log2;        // log2 of the hash table size
    uint64_t pad_[4];
};
static_assert(sizeof(DevDesc) == 128, "DevDesc is two 64-byte lines");

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x8 = __attribute__((ext_vector_type(8))) float;

// Compile-time knobs of the bag kernels.
template <int BLOCK = 256, int UNROLL = 8, bool NT_STORE = false, bool NT_META = false,
          int ONEHOT_INFLIGHT = 8, int MIN_WAVES = 1, int BATCHES = 1, bool NT_ROW = false,
          bool SPECULATE = false, bool IDX_SHUFFLE = false, bool CLAMP = false>
struct BagCfg {
    static constexpr bool kClamp = CLAMP;         // clamp row ids to the table and bag ends to n_idx: malformed
                                                  // input gives garbage rows, never an out-of-bounds access
    static constexpr bool kIdxShuffle = IDX_SHUFFLE; // lane group loads a window of indices coalesced, broadcasts by shuffle
    static constexpr bool kSpeculate = SPECULATE; // prefetch indices[bag] before the bounds arrive
    static constexpr int kMinWaves = MIN_WAVES;   // __launch_bounds__ 2nd arg: waves per SIMD wanted
    static constexpr int kBatches = BATCHES;      // 64-bag batches a wavefront takes per step
    static constexpr bool kNtRow = NT_ROW;        // non-temporal row gathers
    static constexpr int kBlock = BLOCK;          // threads per workgroup (multiple of 64)
    static constexpr int kUnroll = UNROLL;        // row gathers in flight per lane inside a bag
    static constexpr bool kNtStore = NT_STORE;    // non-temporal pooled-row stores
    static constexpr bool kNtMeta = NT_META;      // non-temporal index / offset loads
    static constexpr int kOneHot = ONEHOT_INFLIGHT;  // rounds batched on the one-hot fast path
};

// ---- per-dtype accumulate / store -----------------------------------------------------------
template <int DT>
struct RowOps;

// Address space of row / index / pooled-row accesses.  Pointers that come out of a descriptor in memory are generic
// ("flat"), and the SHIPPED build keeps them so (PIMEMB_GLOBAL_AS 0): its row gathers are flat_load_dwordx4, its pooled-row
// stores flat_store_dwordx4 ... nt.  Every buffer this library touches is GPU-visible memory (HBM, or pinned host memory on
// the zero-copy host path), never LDS or scratch, so they COULD be accessed as GLOBAL (address space 1, -DPIMEMB_GLOBAL_AS=1):
// global_load / global_store skip the aperture check and count on vmcnt alone.  Measured A/B on MI355X: no difference on any
// shape (profiles/r02/tune_address_space_flat_vs_global.log), so the default stays the plain one.
#ifndef PIMEMB_GLOBAL_AS
#define PIMEMB_GLOBAL_AS 0
#endif
#if PIMEMB_GLOBAL_AS
#define PIMEMB_AS __attribute__((address_space(1)))
#else
#define PIMEMB_AS
#endif

template <bool NT>
__device__ __forceinline__ void store_f32x4(float *dst, f32x4 v) {
    typedef f32x4 PIMEMB_AS *ptr_t;
    if constexpr (NT)
        __builtin_nontemporal_store(v, (ptr_t)(reinterpret_cast<f32x4 *>(dst)));
    else
        *(ptr_t)(reinterpret_cast<f32x4 *>(dst)) = v;
}

template <>
struct RowOps<EMB_F32> {
    using Acc = f32x4;
    static constexpr uint32_t kFloatsPerLane = 4;
    static constexpr bool kGroupStore = false;
    static __device__ __forceinline__ Acc zero() { return Acc{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void add(Acc &a, u32x4 raw) { a += __builtin_bit_cast(f32x4, raw); }
    template <bool NT>
    static __device__ __forceinline__ void store(const Acc &a, float *dst) {
        store_f32x4<NT>(dst, a);
    }
};

template <>
struct RowOps<EMB_F16> {
    using Acc = f32x8;
    static constexpr uint32_t kFloatsPerLane = 8;
    // A lane holds 8 consecutive output floats (32 B): storing them as two 16-byte pieces per lane makes
    // every store instruction hit alternate 16-byte pieces (32-B lane stride), which non-temporal stores
    // cannot combine -- measured 3.5 TB/s on one-hot fp16 lookups against ~7 for fp32.  store_row()
    // therefore re-deals the pieces inside the lane group so that each instruction writes a contiguous run
    // (rows of >= 8 lanes; narrower rows use plain stores and leave the merging to the L2).
    static constexpr bool kGroupStore = true;
    static __device__ __forceinline__ Acc zero() { return Acc{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void add(Acc &a, u32x4 raw) {
        a += __builtin_convertvector(__builtin_bit_cast(f16x8, raw), f32x8);
    }
    template <bool NT>
    static __device__ __forceinline__ void store(const Acc &a, float *dst) {
        store_f32x4<NT>(dst, f32x4{a[0], a[1], a[2], a[3]});
        store_f32x4<NT>(dst + 4, f32x4{a[4], a[5], a[6], a[7]});
    }
};

template <>
struct RowOps<EMB_FIXED32> {
    using Acc = u32x4;  // unsigned add == int32 two's-complement wrap (emb_dpu_lookup.c:114)
    static constexpr uint32_t kFloatsPerLane = 4;
    static constexpr bool kGroupStore = false;
    static __device__ __forceinline__ Acc zero() { return Acc{0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ void add(Acc &a, u32x4 raw) { a += raw; }
    static __device__ __forceinline__ float conv(uint32_t acc) {
        // emb_host.h:210: (float)tmp / pow(10,9): int32 -> float, divide in double, round to float
        return (float)((double)(float)(int32_t)acc / 1.0e9);
    }
    template <bool NT>
    static __device__ __forceinline__ void store(const Acc &a, float *dst) {
        store_f32x4<NT>(dst, f32x4{conv(a[0]), conv(a[1]), conv(a[2]), conv(a[3])});
    }
};

template <bool NT, typename T>
__device__ __forceinline__ T load_meta(const T *p) {
    typedef const T PIMEMB_AS *ptr_t;
    if constexpr (NT)
        return __builtin_nontemporal_load((ptr_t)p);
    else
        return *(ptr_t)p;
}

// Row id -> row id inside the table.  A 64-bit row id that came from a uint32 index is clamped with
// one 32-bit min (the high half is known to be zero); int64 indices (negative = huge unsigned) take
// the 64-bit compare.
template <bool CLAMP, typename IdxT = int64_t>
__device__ __forceinline__ uint64_t clamp_row(uint64_t r, uint64_t last_row) {
    if constexpr (!CLAMP) {
        return r;
    } else if constexpr (sizeof(IdxT) == 4) {
        const uint32_t last32 = last_row > 0xffffffffull ? 0xffffffffu : (uint32_t)last_row;
        const uint32_t r32 = (uint32_t)r;
        return (uint64_t)(r32 < last32 ? r32 : last32);
    } else {
        return r < last_row ? r : last_row;
    }
}

template <bool NT>
__device__ __forceinline__ u32x4 load_row(const char *p) {
    typedef const u32x4 PIMEMB_AS *ptr_t;
    if constexpr (NT)
        return __builtin_nontemporal_load((ptr_t)(reinterpret_cast<const u32x4 *>(p)));
    else
        return *(ptr_t)(reinterpret_cast<const u32x4 *>(p));
}

__device__ __forceinline__ uint32_t shfl_u32(uint32_t v, uint32_t src) {
    return (uint32_t)__shfl((int)v, (int)src, 64);
}
__device__ __forceinline__ uint64_t shfl_u64(uint64_t v, uint32_t src) {
    uint32_t lo = shfl_u32((uint32_t)v, src), hi = shfl_u32((uint32_t)(v >> 32), src);
    return ((uint64_t)hi << 32) | lo;
}
template <typename IdxT>
__device__ __forceinline__ uint64_t shfl_index(IdxT v, uint32_t src) {
    if constexpr (sizeof(IdxT) == 8)
        return shfl_u64((uint64_t)v, src);
    else
        return (uint64_t)shfl_u32((uint32_t)v, src);
}

__device__ __forceinline__ float shfl_f32(float v, uint32_t src) {
    return __builtin_bit_cast(float, shfl_u32(__builtin_bit_cast(uint32_t, v), src));
}

// Store one pooled row held by a lane group.  `row` = out + bag * out_stride (the row's first float);
// `valid`: this lane owns a piece of a real bag.  Ops without kGroupStore: every lane stores its own
// 16 bytes (already contiguous across the group).  kGroupStore (fp16 tables: 32 B of output per lane):
// the row is 2*chunks 16-byte pieces, lane s holds pieces 2s and 2s+1; instruction 1 lets lane j write
// piece j, instruction 2 piece LPR + j, both fetched from the owning lane with ds_bpermute -- EVERY lane
// of the group must call this (the shuffles read the neighbours' registers).
template <class Ops, class Cfg, int LPR>
__device__ __forceinline__ void store_row(const typename Ops::Acc &acc, float *__restrict__ row, uint32_t sub,
                                          uint32_t grp, uint32_t chunks, bool valid) {
    if constexpr (!Ops::kGroupStore) {
        if (valid) Ops::template store<Cfg::kNtStore>(acc, row + sub * Ops::kFloatsPerLane);
    } else if constexpr (LPR <= 4) {
        // narrow rows: the shuffles cost more than they save (fp16 dim 16 one-hot: 42 us with them, 21 us
        // without); two plain 16-byte stores per lane instead, which the L2 merges into full lines
        if (valid) Ops::template store<false>(acc, row + sub * Ops::kFloatsPerLane);
    } else {
        const uint32_t pieces = 2u * chunks;
#pragma unroll
        for (uint32_t inst = 0; inst < 2; inst++) {
            const uint32_t p = inst * LPR + sub;               // piece this lane writes
            const uint32_t src = grp * LPR + (p >> 1);         // lane that holds it
            const bool hi = (p & 1u) != 0u;
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const float lo_c = shfl_f32(acc[c], src), hi_c = shfl_f32(acc[4 + c], src);
                v[c] = hi ? hi_c : lo_c;
            }
            // (validity is uniform over a lane group except for lanes beyond `chunks`, whose own
            // `valid` is false but which still write pieces of the row when p < pieces)
            const bool bag_ok = __shfl((int)valid, (int)(grp * LPR), 64) != 0;
            if (bag_ok && p < pieces) store_f32x4<Cfg::kNtStore>(row + 4u * p, v);
        }
    }
}

// Workgroup -> (descriptor, tile).  xmap == nullptr: 2-D grid (x = tile, y = descriptor).
// Otherwise the 1-D XCD-aware map built by pimemb_xcd_map.h: residue class blockIdx.x % 8 owns a
// contiguous share of the (table, tile) list, so a table's rows stay in ONE XCD's L2.
struct XcdSegDev {
    uint32_t desc, tile0, slot_begin, slot_end;
};
constexpr uint32_t kXmapDirect = 0x80000000u;   // flag in the `chunks` kernel argument: xmap is a per-workgroup table
constexpr uint64_t kRangeOpenEnd = 1ull << 63;  // flag in DevDesc::pad_[0] of a ranged launch (EMB_RANGE_OPEN_END, pimemb.h)

__device__ __forceinline__ bool decode_block(const uint32_t *__restrict__ xmap, uint32_t direct,
                                             uint32_t *desc_i, uint32_t *tile) {
    if (xmap == nullptr) {
        *desc_i = blockIdx.y;
        *tile = blockIdx.x;
        return true;
    }
    if (direct) {   // one 8-byte scalar load: {descriptor, tile} of this workgroup (0xffffffff = idle slot)
        const uint2 ent = reinterpret_cast<const uint2 *>(xmap)[blockIdx.x];
        *desc_i = ent.x;
        *tile = ent.y;
        return ent.x != 0xffffffffu;
    }
    const uint32_t cls = blockIdx.x & 7u, slot = blockIdx.x >> 3;
    const uint32_t ns = xmap[cls];
    const XcdSegDev *segs = reinterpret_cast<const XcdSegDev *>(xmap + 16) + xmap[8 + cls];
    // first segment whose slot_end exceeds `slot` (segments are sorted by slot): binary search, so a
    // launch over hundreds of tables costs ~log2 scalar loads per workgroup, not a linear walk
    uint32_t lo = 0, hi = ns;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (slot < segs[mid].slot_end)
            hi = mid;
        else
            lo = mid + 1;
    }
    if (lo >= ns) return false;
    const XcdSegDev sg = segs[lo];
    *desc_i = sg.desc;
    *tile = sg.tile0 + (slot - sg.slot_begin);
    return true;
}

// Walk one bag [p, e) in index order, adding the rows `fetch(row)` returns.  Two ways to get at the
// indices: (a) every lane of the group loads indices[p+k] itself (same address in all LPR lanes),
// kUnroll at a time; (b) kIdxShuffle: the group loads a WINDOW of indices coalesced -- lane `sub`
// holds indices[p + sub + i*LPR] -- and broadcasts them with ds_bpermute, one vector-memory
// instruction per LPR*IPL indices instead of one per index.  Every lane of the wavefront must call
// this (the shuffles need the whole lane group active); `live` lanes own a 16-byte piece of the row.
template <typename IdxT, int LPR, class Cfg, class Ops, class Probe, class Fetch>
__device__ __forceinline__ void walk_bag(const IdxT *__restrict__ indices, uint64_t p, uint64_t e, uint32_t sub,
                                         uint32_t grp, bool live, typename Ops::Acc &acc, Probe &&probe,
                                         Fetch &&fetch) {
    // probe(row) -> 32-bit token computed ONCE per index (e.g. the LDS slot of a hot row);
    // fetch(row, token) -> the lane's 16-byte piece of that row.
    constexpr int U = Cfg::kUnroll;
    constexpr bool kProbe = std::remove_reference_t<Probe>::kUsed;
    if constexpr (!Cfg::kIdxShuffle) {
        if (!live) return;
        for (; p + U <= e; p += U) {
            uint64_t r[U];
#pragma unroll
            for (int k = 0; k < U; k++) r[k] = (uint64_t)load_meta<Cfg::kNtMeta>(indices + p + k);
            u32x4 v[U];
#pragma unroll
            for (int k = 0; k < U; k++) v[k] = fetch(r[k], probe(r[k]));
#pragma unroll
            for (int k = 0; k < U; k++) Ops::add(acc, v[k]);
        }
        for (; p < e; p++) {
            const uint64_t r = (uint64_t)load_meta<Cfg::kNtMeta>(indices + p);
            Ops::add(acc, fetch(r, probe(r)));
        }
    } else {
        constexpr uint32_t IPL = (U + LPR - 1) / LPR;      // indices held per lane
        constexpr uint32_t W = IPL * LPR;                  // window
        while (p < e) {
            const uint32_t cnt = (e - p < W) ? (uint32_t)(e - p) : W;
            IdxT mine[IPL];
            uint32_t tok[IPL];
#pragma unroll
            for (uint32_t i = 0; i < IPL; i++) {
                mine[i] = 0;
                if (sub + i * LPR < cnt) mine[i] = load_meta<Cfg::kNtMeta>(indices + p + sub + i * LPR);
                tok[i] = probe((uint64_t)mine[i]);         // LPR different probes per instruction
            }
            uint32_t k = 0;
            for (; k + U <= cnt; k += U) {
                uint64_t r[U];
                uint32_t t[U];
#pragma unroll
                for (int j = 0; j < U; j++) {
                    const uint32_t q = k + j, src = grp * LPR + q % LPR;   // position in the window
                    r[j] = shfl_index<IdxT>(mine[(IPL == 1) ? 0 : q / LPR], src);
                    t[j] = kProbe ? shfl_u32(tok[(IPL == 1) ? 0 : q / LPR], src) : 0u;
                }
                if (live) {
                    u32x4 v[U];
#pragma unroll
                    for (int j = 0; j < U; j++) v[j] = fetch(r[j], t[j]);
#pragma unroll
                    for (int j = 0; j < U; j++) Ops::add(acc, v[j]);
                }
            }
            for (; k < cnt; k++) {
                const uint32_t src = grp * LPR + k % LPR;
                const uint64_t r = shfl_index<IdxT>(mine[(IPL == 1) ? 0 : k / LPR], src);
                const uint32_t t = kProbe ? shfl_u32(tok[(IPL == 1) ? 0 : k / LPR], src) : 0u;
                if (live) Ops::add(acc, fetch(r, t));
            }
            p += cnt;
        }
    }
}

struct NoProbe {
    static constexpr bool kUsed = false;
    __device__ __forceinline__ uint32_t operator()(uint64_t) const { return 0u; }
};

// ---- v1: one lane group per bag, no cross-lane traffic (kept as the A/B baseline) -------------
template <typename IdxT, int DT, int LPR, class Cfg>
__global__ void __launch_bounds__(Cfg::kBlock)
bag_sum_group_kernel(const DevDesc *__restrict__ descs, uint32_t chunks_arg,
                     const uint32_t *__restrict__ xmap) {
    using Ops = RowOps<DT>;
    constexpr uint32_t kWaves = Cfg::kBlock / 64;
    constexpr uint32_t BPW = 64 / LPR;
    constexpr uint32_t BAGS_PER_TILE = BPW * kWaves;

    uint32_t desc_i, tile;
    if (!decode_block(xmap, chunks_arg & kXmapDirect, &desc_i, &tile)) return;
    const uint32_t chunks = chunks_arg & ~kXmapDirect;
    const DevDesc *dp = descs + desc_i;
    const char *__restrict__ weights = static_cast<const char *>(dp->weights);
    const IdxT *__restrict__ indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
    float *__restrict__ out = dp->out;
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags, last_row = dp->nr_rows - 1;
    const uint32_t fixed_pooling = dp->fixed_pooling, n_tiles = dp->n_tiles;

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (LPR - 1), grp = lane / LPR;
    const uint32_t row_bytes = chunks * 16u;
    const uint32_t out_stride = chunks * Ops::kFloatsPerLane;
    const char *__restrict__ wsub = weights + sub * 16u;

    if (tile < n_tiles) {
        const uint64_t bag = (uint64_t)tile * BAGS_PER_TILE + wave * BPW + grp;
        const bool live = sub < chunks;
        if (bag >= n_bags) return;                     // whole lane group leaves together
        uint64_t p, e;
        if (offsets != nullptr) {
            p = (uint64_t)load_meta<Cfg::kNtMeta>(offsets + bag);
            e = (bag + 1 < n_bags) ? (uint64_t)load_meta<Cfg::kNtMeta>(offsets + bag + 1) : n_idx;
        } else {
            p = bag * fixed_pooling;
            e = p + fixed_pooling;
        }
        if (Cfg::kClamp && e > n_idx) e = n_idx;
        typename Ops::Acc acc = Ops::zero();
        walk_bag<IdxT, LPR, Cfg, Ops>(indices, p, e, sub, grp, live, acc, NoProbe{},
                                      [&](uint64_t r, uint32_t) -> u32x4 {
                                          return load_row<Cfg::kNtRow>(wsub + clamp_row<Cfg::kClamp, IdxT>(r, last_row) * row_bytes);
                                      });
        store_row<Ops, Cfg, LPR>(acc, out + bag * out_stride, sub, grp, chunks, live);
    }
}

// ---- any row width: one thread per output element --------------------------------------------
// Rows whose byte size is not a multiple of 16 (dim 2, 3, 10, 100 ... -- nn.EmbeddingBag takes any
// dim, dlrm_s_pytorch.py's default sparse feature size is 2) or exceeds 1 KiB cannot use the 16-byte
// lane pieces above.  `lanes` (power of two, <= 256) consecutive threads own one bag and stride over
// its columns; each output element is still summed by one thread in index order from +0, so results
// are bit-identical to the vector kernels and the oracle.  Generality path, not a tuned one.
template <int DT> struct ElemOps;
template <> struct ElemOps<EMB_F32> {
    using Elem = float; using Acc = float;
    static __device__ __forceinline__ void add(Acc &a, Elem v) { a = a + v; }
    static __device__ __forceinline__ float out(Acc a) { return a; }
};
template <> struct ElemOps<EMB_F16> {
    using Elem = _Float16; using Acc = float;
    static __device__ __forceinline__ void add(Acc &a, Elem v) { a = a + (float)v; }
    static __device__ __forceinline__ float out(Acc a) { return a; }
};
template <> struct ElemOps<EMB_FIXED32> {
    using Elem = uint32_t; using Acc = uint32_t;
    static __device__ __forceinline__ void add(Acc &a, Elem v) { a += v; }
    static __device__ __forceinline__ float out(Acc a) { return RowOps<EMB_FIXED32>::conv(a); }
};

template <typename IdxT, int DT, bool CLAMP>
__global__ void __launch_bounds__(256)
bag_sum_anydim_kernel(const DevDesc *__restrict__ descs, uint32_t dim, uint32_t lanes) {
    using E = ElemOps<DT>;
    const DevDesc *dp = descs + blockIdx.y;
    const typename E::Elem *__restrict__ weights = static_cast<const typename E::Elem *>(dp->weights);
    const IdxT *__restrict__ indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
    float *__restrict__ out = dp->out;
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags, last_row = dp->nr_rows - 1;
    const uint64_t bag = (uint64_t)blockIdx.x * (256u / lanes) + threadIdx.x / lanes;
    if (blockIdx.x >= dp->n_tiles || bag >= n_bags) return;
    uint64_t p0, e;
    if (offsets != nullptr) {
        p0 = (uint64_t)offsets[bag];
        e = (bag + 1 < n_bags) ? (uint64_t)offsets[bag + 1] : n_idx;
    } else {
        p0 = bag * dp->fixed_pooling;
        e = p0 + dp->fixed_pooling;
    }
    if (CLAMP && e > n_idx) e = n_idx;
    for (uint32_t col = threadIdx.x & (lanes - 1); col < dim; col += lanes) {
        typename E::Acc acc = 0;
        for (uint64_t p = p0; p < e; p++) {
            const uint64_t r = clamp_row<CLAMP, IdxT>((uint64_t)indices[p], last_row);
            E::add(acc, weights[r * dim + col]);
        }
        out[bag * dim + col] = E::out(acc);
    }
}

// Rows whose byte size is a multiple of 4 (every fp32 / fixed-point dim, even fp16 dims) but not of 16:
// one thread per 16-byte PIECE instead of per element.  A piece of an unpadded row starts at a 4-byte
// aligned address, which global_load/store_dwordx4 accept on gfx950, so full pieces move as vectors and
// only the last, partial piece of a row goes element by element.  Same summation order, same bits.
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef float f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

template <typename IdxT, int DT, bool CLAMP>
__global__ void __launch_bounds__(256)
bag_sum_anydim_vec_kernel(const DevDesc *__restrict__ descs, uint32_t dim, uint32_t lanes) {
    using Ops = RowOps<DT>;
    constexpr uint32_t EP = Ops::kFloatsPerLane;            // elements per 16-byte piece (4, or 8 halves)
    constexpr uint32_t ESZ = 16u / EP;
    constexpr int U = 4;
    const DevDesc *dp = descs + blockIdx.y;
    const char *__restrict__ weights = static_cast<const char *>(dp->weights);
    const IdxT *__restrict__ indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
    float *__restrict__ out = dp->out;
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags, last_row = dp->nr_rows - 1;
    const uint64_t bag = (uint64_t)blockIdx.x * (256u / lanes) + threadIdx.x / lanes;
    if (blockIdx.x >= dp->n_tiles || bag >= n_bags) return;
    uint64_t p0, e;
    if (offsets != nullptr) {
        p0 = (uint64_t)offsets[bag];
        e = (bag + 1 < n_bags) ? (uint64_t)offsets[bag + 1] : n_idx;
    } else {
        p0 = bag * dp->fixed_pooling;
        e = p0 + dp->fixed_pooling;
    }
    if (CLAMP && e > n_idx) e = n_idx;
    const uint32_t row_bytes = dim * ESZ, pieces = (dim + EP - 1) / EP;
    for (uint32_t piece = threadIdx.x & (lanes - 1); piece < pieces; piece += lanes) {
        const uint32_t n_el = (dim - piece * EP < EP) ? dim - piece * EP : EP;
        const char *__restrict__ wp = weights + piece * 16u;
        auto fetch = [&](uint64_t r) -> u32x4 {
            const char *src = wp + clamp_row<CLAMP, IdxT>(r, last_row) * row_bytes;
            if (n_el == EP) return *reinterpret_cast<const u32x4_a4 *>(src);
            u32x4 v = {0u, 0u, 0u, 0u};                     // partial last piece: whole dwords only
            const uint32_t n_dw = n_el * ESZ / 4u;
            for (uint32_t d = 0; d < n_dw; d++) v[d] = reinterpret_cast<const uint32_t *>(src)[d];
            return v;
        };
        typename Ops::Acc acc = Ops::zero();
        uint64_t p = p0;
        for (; p + U <= e; p += U) {
            uint64_t r[U];
#pragma unroll
            for (int k = 0; k < U; k++) r[k] = (uint64_t)indices[p + k];
            u32x4 v[U];
#pragma unroll
            for (int k = 0; k < U; k++) v[k] = fetch(r[k]);
#pragma unroll
            for (int k = 0; k < U; k++) Ops::add(acc, v[k]);
        }
        for (; p < e; p++) Ops::add(acc, fetch((uint64_t)indices[p]));
        float *o = out + bag * dim + piece * EP;
        float res[EP];
        if constexpr (DT == EMB_FIXED32) {
#pragma unroll
            for (uint32_t c = 0; c < EP; c++) res[c] = RowOps<EMB_FIXED32>::conv(acc[c]);
        } else {
#pragma unroll
            for (uint32_t c = 0; c < EP; c++) res[c] = acc[c];
        }
        if (n_el == EP) {
#pragma unroll
            for (uint32_t c = 0; c < EP; c += 4)
                *reinterpret_cast<f32x4_a4 *>(o + c) = f32x4{res[c], res[c + 1], res[c + 2], res[c + 3]};
        } else {
            for (uint32_t c = 0; c < n_el; c++) o[c] = res[c];
        }
    }
}

// Rounds of the one-hot fast path a wavefront keeps in flight for 1-KiB rows (LPR = 64: one bag per round).  Eight -- what every
// narrower row width uses -- needs two registers more than the 64-VGPR cap holds there (12 bytes of scratch per lane); four leaves one.
// 26 Kaggle-sized tables, dim 256 fp32, B = 39 292, us per launch (tools/wide_row_onehot_probe.py, profiles/r06/wide_rows_*.log):
// 8 in flight 270.4, 4 in flight 257.0-259.6 (-4 %), 2 in flight 267.4; dim 128 (LPR = 32, untouched) 122.6-125.2 in the same runs.
#ifndef PIMEMB_LPR64_ONEHOT_INFLIGHT
#define PIMEMB_LPR64_ONEHOT_INFLIGHT 4
#endif
// ... and for 512-byte rows (LPR = 32: dim 128 fp32, the Terabyte shape): tools/onehot_inflight_sweep.sh
#ifndef PIMEMB_LPR32_ONEHOT_INFLIGHT
#define PIMEMB_LPR32_ONEHOT_INFLIGHT 8
#endif

// ---- v2: wave batches of 64 bags, coalesced bounds, shuffle-distributed, one-hot fast path -------
// RANGED (the sharded lookup's direct path, pimemb_shard.cpp): one index per bag, and a descriptor serves only the bags
// whose row falls into [row_lo, row_lo + nr_rows) (row_lo in DevDesc::pad_[0]): out[b] = W[idx[b] - row_lo]; the other bags
// are left untouched -- another shard of the table writes them, straight into the same output (pad_[1], if not null: a
// uint32 counter in HBM the launch adds the number of bags it served to).  A shard scans the
// requester's RAW index array (its own, or a peer's through its mapping): no router, no counts, no un-router.  A whole
// table is the range [0, nr_rows), so replicated tables and shards share ONE launch of the tuned kernel.
template <typename IdxT, int DT, int LPR, class Cfg, bool RANGED = false>
__global__ void __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves)
bag_sum_wavebatch_kernel(const DevDesc *__restrict__ descs, uint32_t chunks_arg,
                         const uint32_t *__restrict__ xmap) {
    using Ops = RowOps<DT>;
    constexpr uint32_t kWaves = Cfg::kBlock / 64;
    constexpr uint32_t BPR = 64 / LPR;      // bags per round
    constexpr uint32_t ROUNDS = LPR;        // rounds per 64-bag wave batch
    constexpr uint32_t NB = Cfg::kBatches;  // wave batches per step
    constexpr int U = Cfg::kUnroll;
    // (1-KiB rows, LPR = 64: eight rounds in flight = 8 KiB per wavefront and two registers more than the 64-VGPR cap of the
    // fp32 configurations holds -- 12 bytes of scratch per lane; PIMEMB_LPR64_ONEHOT_INFLIGHT picks the depth for that row width)
    constexpr uint32_t kInFlight = (LPR == 64) ? (uint32_t)PIMEMB_LPR64_ONEHOT_INFLIGHT : (LPR == 32) ? (uint32_t)PIMEMB_LPR32_ONEHOT_INFLIGHT : (uint32_t)Cfg::kOneHot;
    constexpr uint32_t RU = (ROUNDS < kInFlight) ? ROUNDS : kInFlight;

    uint32_t desc_i, tile;
    if (!decode_block(xmap, chunks_arg & kXmapDirect, &desc_i, &tile)) return;
    const uint32_t chunks = chunks_arg & ~kXmapDirect;
    const DevDesc *dp = descs + desc_i;
    const char *__restrict__ weights = static_cast<const char *>(dp->weights);
    const IdxT *__restrict__ indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
    float *__restrict__ out = dp->out;
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags, last_row = dp->nr_rows - 1;
    const uint32_t fixed_pooling = dp->fixed_pooling, n_tiles = dp->n_tiles;

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (LPR - 1), grp = lane / LPR;
    const uint32_t row_bytes = chunks * 16u;
    const uint32_t out_stride = chunks * Ops::kFloatsPerLane;
    const char *__restrict__ wsub = weights + sub * 16u;
    const bool lane_live = sub < chunks;
    uint64_t row_lo = 0;
    bool open_end = false;               // RANGED: ids at or beyond the end of this range are this descriptor's to ZERO (EMB_RANGE_OPEN_END)
    uint32_t *served_ctr = nullptr;      // RANGED: where this descriptor's launch adds the number of bags it served (or null)
    if constexpr (RANGED) {
        row_lo = dp->pad_[0] & ~kRangeOpenEnd;
        open_end = (dp->pad_[0] & kRangeOpenEnd) != 0;
        served_ctr = reinterpret_cast<uint32_t *>(dp->pad_[1]);
    }

    if (tile < n_tiles) {
        const uint64_t step_base = ((uint64_t)tile * kWaves + wave) * (64u * NB);
        if (step_base >= n_bags) return;  // wave-uniform

        // lane l of batch q holds the bounds of bag step_base + 64q + l (past-the-end = empty)
        uint64_t st[NB];
        uint32_t len[NB];
        bool small = true;
        // Speculation: in a one-hot batch with contiguous bags offsets[b] == b, so bag b's only index
        // sits at indices[b].  Fetch it NOW, in the shadow of the bounds loads, and keep it only if
        // the bounds agree -- one memory round trip less on the critical path of the Kaggle shape.
        IdxT spec[NB];
        if constexpr (Cfg::kSpeculate) {
#pragma unroll
            for (uint32_t q = 0; q < NB; q++) {
                const uint64_t mb = step_base + 64u * q + lane;
                spec[q] = 0;
                if (mb < n_idx) spec[q] = load_meta<Cfg::kNtMeta>(indices + mb);
            }
        }
#pragma unroll
        for (uint32_t q = 0; q < NB; q++) {
            const uint64_t mb = step_base + 64u * q + lane;
            uint64_t en;
            if (offsets != nullptr) {
                st[q] = (mb < n_bags) ? (uint64_t)load_meta<Cfg::kNtMeta>(offsets + mb) : n_idx;
                en = (mb + 1 < n_bags) ? (uint64_t)load_meta<Cfg::kNtMeta>(offsets + mb + 1) : n_idx;
            } else {
                st[q] = (mb < n_bags ? mb : n_bags) * fixed_pooling;
                en = (mb + 1 < n_bags ? mb + 1 : n_bags) * fixed_pooling;
            }
            if (Cfg::kClamp) {                       // malformed offsets: keep [st, en) inside [0, n_idx]
                if (en > n_idx) en = n_idx;
                if (st[q] > en) st[q] = en;
            }
            len[q] = (uint32_t)(en - st[q]);
            small = small && (len[q] <= 1u);
        }

        if (__all(small)) {
            // one-hot step: lane l fetches its bag's only index (coalesced), groups pull theirs and
            // every gather of the step is issued before the first store.
            IdxT my[NB];
#pragma unroll
            for (uint32_t q = 0; q < NB; q++) {
                my[q] = 0;
                if constexpr (Cfg::kSpeculate) {
                    const uint64_t mb = step_base + 64u * q + lane;
                    if (len[q]) my[q] = (st[q] == mb) ? spec[q] : load_meta<Cfg::kNtMeta>(indices + st[q]);
                } else {
                    if (len[q]) my[q] = load_meta<Cfg::kNtMeta>(indices + st[q]);
                }
                if constexpr (RANGED) {     // row ids relative to this shard; a bag of another shard counts as empty
                    const uint64_t id = (uint64_t)my[q];               // (a negative int64 id: huge, beyond every range)
                    const uint64_t r = id - row_lo;                    // (wraps far out of range below row_lo)
                    // len: 1 = served here; 2 = an id no range holds, and this descriptor answers for the open end: the bag's
                    // pooled row is written as zeros (not served, not counted); 0 = some other shard's bag, left alone
                    if (r > last_row) len[q] = (len[q] && open_end && id >= row_lo) ? 2u : 0u;
                    my[q] = (IdxT)r;
                }
            }
            if constexpr (RANGED) {
                // Counted launch (a checked shard's direct path): the bags this wavefront serves, one atomic per wavefront,
                // no return value.  The requester adds the counts of all shards up: a sum short of its bag count means an
                // index no shard holds -- a bag left untouched (pimemb_shard.cpp: stage_unroute).
                // A counter is EMB_SERVED_LANES words EMB_SERVED_STRIDE bytes apart, a workgroup adds to lane blockIdx % LANES:
                // agent-scope atomics on ONE line retire one by one (~5.5 ns each: 8 000 wavefronts of the Kaggle launch on 26
                // neighbouring words took 44 us against the lookup's 20), spread over lines they cost nothing measurable.
                if (served_ctr != nullptr) {     // wave-uniform (a scalar load of the descriptor)
                    uint32_t n = 0;
#pragma unroll
                    for (uint32_t q = 0; q < NB; q++) n += (uint32_t)__popcll(__ballot(len[q] == 1u));
                    uint32_t *slot = served_ctr + (size_t)(blockIdx.x % EMB_SERVED_LANES) * (EMB_SERVED_STRIDE / 4u);
                    if (lane == 0 && n) (void)__hip_atomic_fetch_add(slot, n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
#pragma unroll
            for (uint32_t j0 = 0; j0 < ROUNDS; j0 += RU) {
                u32x4 v[NB][RU];
                bool has[NB][RU];
                bool zero_it[NB][RU];      // (RANGED only: the open end's bags)
#pragma unroll
                for (uint32_t q = 0; q < NB; q++)
#pragma unroll
                    for (uint32_t jj = 0; jj < RU; jj++) {
                        const uint32_t src = (j0 + jj) * BPR + grp;
                        const uint64_t r = clamp_row<Cfg::kClamp, IdxT>(shfl_index<IdxT>(my[q], src), last_row);
                        const uint32_t l = shfl_u32(len[q], src);
                        has[q][jj] = RANGED ? l == 1u : l != 0u;
                        zero_it[q][jj] = RANGED && l == 2u;
                        v[q][jj] = u32x4{0u, 0u, 0u, 0u};
                        if (has[q][jj] && lane_live) v[q][jj] = load_row<Cfg::kNtRow>(wsub + r * row_bytes);
                    }
#pragma unroll
                for (uint32_t q = 0; q < NB; q++)
#pragma unroll
                    for (uint32_t jj = 0; jj < RU; jj++) {
                        const uint64_t bag = step_base + 64u * q + (j0 + jj) * BPR + grp;
                        // (RANGED: only the bags this shard holds the row of are written)
                        const bool wr = RANGED ? (has[q][jj] || zero_it[q][jj]) : bag < n_bags;
                        if constexpr (!Ops::kGroupStore) {
                            if (wr && lane_live) {
                                typename Ops::Acc acc = Ops::zero();
                                if (has[q][jj]) Ops::add(acc, v[q][jj]);
                                Ops::template store<Cfg::kNtStore>(
                                    acc, out + bag * out_stride + sub * Ops::kFloatsPerLane);
                            }
                        } else {   // all lanes take part in the group store's shuffles
                            typename Ops::Acc acc = Ops::zero();
                            if (has[q][jj] && lane_live) Ops::add(acc, v[q][jj]);
                            store_row<Ops, Cfg, LPR>(acc, out + bag * out_stride, sub, grp, chunks,
                                                     wr && lane_live);
                        }
                    }
            }
            return;
        }
        if constexpr (RANGED) return;      // (the host hands a ranged launch fixed_pooling 1 only: every step is one-hot)

        // general step: each round, a lane group walks its bag in index order
#pragma unroll 1
        for (uint32_t q = 0; q < NB; q++) {
#pragma unroll 1
            for (uint32_t j = 0; j < ROUNDS; j++) {
                const uint32_t src = j * BPR + grp;
                uint64_t p = shfl_u64(st[q], src);
                const uint64_t e = p + shfl_u32(len[q], src);
                const uint64_t bag = step_base + 64u * q + src;
                // (no early `continue`: every lane must reach the next round's shuffles)
                typename Ops::Acc acc = Ops::zero();
                if (bag < n_bags && lane_live) {
                    for (; p + U <= e; p += U) {
                        uint64_t r[U];
#pragma unroll
                        for (int k = 0; k < U; k++) r[k] = (uint64_t)load_meta<Cfg::kNtMeta>(indices + p + k);
                        u32x4 v[U];
#pragma unroll
                        for (int k = 0; k < U; k++)
                            v[k] = load_row<Cfg::kNtRow>(wsub + clamp_row<Cfg::kClamp, IdxT>(r[k], last_row) * row_bytes);
#pragma unroll
                        for (int k = 0; k < U; k++) Ops::add(acc, v[k]);
                    }
                    for (; p < e; p++) {
                        const uint64_t r = clamp_row<Cfg::kClamp, IdxT>((uint64_t)load_meta<Cfg::kNtMeta>(indices + p), last_row);
                        Ops::add(acc, load_row<Cfg::kNtRow>(wsub + r * row_bytes));
                    }
                    if constexpr (!Ops::kGroupStore)
                        Ops::template store<Cfg::kNtStore>(acc, out + bag * out_stride + sub * Ops::kFloatsPerLane);
                }
                if constexpr (Ops::kGroupStore)
                    store_row<Ops, Cfg, LPR>(acc, out + bag * out_stride, sub, grp, chunks, bag < n_bags && lane_live);
            }
        }
    }
}

// ---- v3: lane-group kernel with the table's HOT ROWS staged in LDS -------------------------------
// For skewed (Zipf-like) pooled workloads.  The engine keeps, per table, a compact copy of up to a
// few hundred hot rows plus a small open-addressing hash (row id -> slot).  A persistent workgroup
// stages both into LDS once, then walks its share of the table's bags: every index is probed in the
// LDS hash (two ds_read_b64); a hit reads the row piece from LDS (ds_read_b128), a miss goes to
// L2/HBM as before.  The hot copy holds the same bits as the table, and the adds stay in index
// order, so results are bit-identical to the other kernels.
constexpr uint32_t kHotEmpty = 0xffffffffu;
constexpr uint32_t kHotProbes = 2;   // the host builder places a row within this many probes or drops it

__device__ __forceinline__ uint32_t hot_hash_pos(uint32_t key, uint32_t log2size) {
    return (key * 0x9E3779B1u) >> (32u - log2size);
}

struct HotProbe {
    static constexpr bool kUsed = true;
    const uint64_t *lhash;
    uint32_t n_hot, log2, mask;
    __device__ __forceinline__ uint32_t operator()(uint64_t r) const {
        if (n_hot == 0 || r >= 0xffffffffull) return 0u;
        const uint32_t key = (uint32_t)r, pos = hot_hash_pos(key, log2);
        const uint64_t e0 = lhash[pos], e1 = lhash[(pos + 1) & mask];
        if ((uint32_t)e0 == key) return 0x80000000u | (uint32_t)(e0 >> 32);
        if ((uint32_t)e1 == key) return 0x80000000u | (uint32_t)(e1 >> 32);
        return 0u;
    }
};

template <typename IdxT, int DT, int LPR, class Cfg>
__global__ void __launch_bounds__(Cfg::kBlock)
bag_sum_hot_kernel(const DevDesc *__restrict__ descs, uint32_t chunks) {
    using Ops = RowOps<DT>;
    constexpr uint32_t kWaves = Cfg::kBlock / 64;
    constexpr uint32_t BPW = 64 / LPR;
    constexpr uint32_t BAGS_PER_TILE = BPW * kWaves;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

    const DevDesc *dp = descs + blockIdx.y;
    const char *__restrict__ weights = static_cast<const char *>(dp->weights);
    const IdxT *__restrict__ indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
    float *__restrict__ out = dp->out;
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags;
    const uint32_t fixed_pooling = dp->fixed_pooling, n_tiles = dp->n_tiles;
    const uint32_t n_hot = dp->n_hot, hot_log2 = dp->hot_log2;
    const uint64_t last_row = dp->nr_rows - 1;

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (LPR - 1), grp = lane / LPR;
    const uint32_t row_bytes = chunks * 16u;
    const uint32_t out_stride = chunks * Ops::kFloatsPerLane;
    const char *__restrict__ wsub = weights + sub * 16u;

    // stage hot rows + hash into LDS (rows first: 16-byte aligned pieces)
    u32x4 *lrows = reinterpret_cast<u32x4 *>(lds_raw);
    uint64_t *lhash = reinterpret_cast<uint64_t *>(lds_raw + (size_t)n_hot * row_bytes);
    const uint32_t hash_mask = (1u << hot_log2) - 1u;
    if (n_hot) {
        const u32x4 *src = static_cast<const u32x4 *>(dp->hot_rows);
        for (uint32_t i = threadIdx.x; i < n_hot * chunks; i += Cfg::kBlock) lrows[i] = src[i];
        for (uint32_t i = threadIdx.x; i <= hash_mask; i += Cfg::kBlock) lhash[i] = dp->hot_hash[i];
    }
    __syncthreads();
    const u32x4 *lsub = lrows + sub;

    // probe: token = 0x80000000 | LDS slot if the row is hot, 0 otherwise (one probe per INDEX: with
    // kIdxShuffle each lane probes its own index of the window, 32 probes per instruction pair)
    HotProbe probe{lhash, n_hot, hot_log2, hash_mask};
    // one gathered row piece: LDS if the row is hot, L2/HBM otherwise
    auto fetch = [&](uint64_t r, uint32_t tok) -> u32x4 {
        if (tok & 0x80000000u) return lsub[(tok & 0x7fffffffu) * chunks];
        return load_row<Cfg::kNtRow>(wsub + clamp_row<Cfg::kClamp, IdxT>(r, last_row) * row_bytes);
    };

    const bool live = sub < chunks;
    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t bag = (uint64_t)tile * BAGS_PER_TILE + wave * BPW + grp;
        if (bag < n_bags) {                            // (no `continue`: lane groups stay whole for the shuffles)
            uint64_t p, e;
            if (offsets != nullptr) {
                p = (uint64_t)load_meta<Cfg::kNtMeta>(offsets + bag);
                e = (bag + 1 < n_bags) ? (uint64_t)load_meta<Cfg::kNtMeta>(offsets + bag + 1) : n_idx;
            } else {
                p = bag * fixed_pooling;
                e = p + fixed_pooling;
            }
            if (Cfg::kClamp && e > n_idx) e = n_idx;
            typename Ops::Acc acc = Ops::zero();
            walk_bag<IdxT, LPR, Cfg, Ops>(indices, p, e, sub, grp, live, acc, probe, fetch);
            store_row<Ops, Cfg, LPR>(acc, out + bag * out_stride, sub, grp, chunks, live);
        }
    }
}

}  // namespace pimemb
