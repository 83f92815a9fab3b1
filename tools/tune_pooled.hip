// tune_pooled.hip -- developer tool: interleaved A/B timing of bag-kernel variants on POOLED
// workloads (BASELINE config C3 shape: T tables x N rows x dim 128 fp32, B bags x L indices per bag,
// Zipf(alpha) ranks scattered by a multiplicative hash).  Same rules as tune_bag_kernels.hip: one
// process, rounds interleaved, every variant cross-checked bit for bit against the first.
//   tune_pooled [T rows B L alpha NB rounds iters]      defaults 16 4000000 16384 32 1.2 2 5 6
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "pimemb_bag_kernels.h"
#include "pimemb_xcd_map.h"
#include "pimemb_hot_rows.h"

using namespace pimemb;

// =====================================================================================================
// REJECTED VARIANT, kept here (tuner only) so the measurement can be repeated: "v4 stream".
// Hypothesis (round 2): the lane-group kernel is bound by outstanding misses per CU -- every bag starts with three
// dependent round trips (descriptor, bounds, index window) with no row in flight, and its eight gathers drain to
// zero before the next eight are issued.  v4 makes workgroups persistent over (descriptor, tile) items, prefetches
// the bounds of item k+2 and the index window and descriptor scalars of item k+1 BEHIND the first eight gathers of
// item k, and consumes the gathers as a rolling ring (global_load so that vmcnt counts in order: add slot j at
// vmcnt(7), re-issue it).  Bit-identical to every other kernel.  Measured on MI355X, 16 tables x 4M rows x dim
// 128, B = 16384, L = 32 (profiles/r02/tunep_stream_kernel_rejected.log): Zipf(1.2) 253-294 us against 222 us for
// the shipped kernel, uniform 791-825 against 762-768 -- slower in every geometry (G = 1024..8192 workgroups, 4 to 8
// waves per SIMD, 8 or 16 gathers in flight), and INSENSITIVE to occupancy and ring depth.  So the launch is not
// short of memory-level parallelism: at 20 TB/s algorithmic the CUs move 32 B/clk each, half of the L1's 64 B/clk,
// with 61 % of those bytes also crossing the L2->L1 fill path.  The simple kernel already sits on the vector-memory
// pipe's throughput; what the extra bookkeeping (scalar item arithmetic, carried descriptor state, 72 registers)
// buys back in latency it loses in issue slots.  Flat vs global address space for the shipped kernels: no
// difference either (-DPIMEMB_GLOBAL_AS=1: C2 19.5 vs 19.5-19.9 us, pooled 223.4 vs 224.0 us).
// =====================================================================================================
namespace pimemb {
// The same 16-byte row piece through a GLOBAL-address-space pointer (global_load_dwordx4).  Pointers that come out
// of a descriptor in memory are generic, and the compiler must assume a flat load may return out of order with
// respect to others: it waits vmcnt(0) before the first use of ANY of them.  Global loads return in order, so a
// ring of gathers can be consumed oldest-first with vmcnt(N-1) while the younger ones stay in flight.
typedef const u32x4 __attribute__((address_space(1))) *global_row_ptr;
template <bool NT, typename T>
__device__ __forceinline__ T load_meta_global(const T *p) {
    typedef const T __attribute__((address_space(1))) *gp;
    if constexpr (NT)
        return __builtin_nontemporal_load((gp)p);
    else
        return *(gp)p;
}
template <bool NT>
__device__ __forceinline__ u32x4 load_row_global(const char *p) {
    global_row_ptr g = (global_row_ptr)(reinterpret_cast<const u32x4 *>(p));
    if constexpr (NT)
        return __builtin_nontemporal_load(g);
    else
        return *g;
}

// ---- v4: persistent lane-group kernel: metadata prefetched two bags ahead, rolling gathers -----------
// For pooled launches.  The lane-group kernel above starts every bag with three DEPENDENT memory round
// trips during which the wavefront has no row in flight -- descriptor (scalar), bag bounds, index window --
// and then drains its eight gathers to zero before it issues the next eight.  Its outstanding misses per CU
// are therefore waves x duty x rows-in-flight x miss rate, and on cache-served mixes (Zipf: 21 % of the row
// bytes leave the XCD) that product, not a bandwidth, sets the speed.  Here a workgroup is persistent: it walks
// work items (descriptor, tile) = blockIdx.x + k * gridDim.x and, while it gathers the rows of item k, the
// bounds of item k+2 and the index window of item k+1 are already on their way (vector memory returns in
// order, so by the time the first gather of item k lands, both have).  Inside a bag the gathers ROLL: slot j
// of the 8-deep ring is added (oldest first, i.e. still in index order from +0 -- same bits as every other
// kernel) and immediately re-issued for index j+8, so eight rows stay in flight until the bag ends.
// Register diet (the first version kept three items' descriptor fields and eight lane masks alive and the
// allocator spilled scalars into AGPRs: 240 registers): a work item is two 32-bit scalars (descriptor, tile)
// advanced incrementally, descriptor fields are re-read with scalar loads where they are used (they sit in the
// scalar cache), a lane group's share of an item is three 32-bit values (bag, p, e; bag = ~0: none), and "slot j
// holds a gather" is recomputed from the window count instead of being carried.  The launch must satisfy
// n_descs * max_tiles < 2^32, n_bags < 2^32 - 1 and (uint32 indices) n_idx < 2^32 -- the host checks.
template <typename IdxT, int DT, int LPR, class Cfg>
__global__ void __launch_bounds__(Cfg::kBlock, Cfg::kMinWaves)
bag_sum_stream_kernel(const DevDesc *__restrict__ descs, uint32_t chunks, uint32_t n_descs, uint32_t max_tiles) {
    using Ops = RowOps<DT>;
    using Pos = typename std::conditional<sizeof(IdxT) == 4, uint32_t, uint64_t>::type;
    constexpr uint32_t kWaves = Cfg::kBlock / 64;
    constexpr uint32_t BPW = 64 / LPR;
    constexpr uint32_t BAGS_PER_TILE = BPW * kWaves;
    constexpr uint32_t U = (uint32_t)Cfg::kUnroll;
    constexpr uint32_t IPL = (U + LPR - 1) / LPR;   // indices held per lane
    constexpr uint32_t W = IPL * LPR;               // index window of a lane group (a multiple of U)
    constexpr uint32_t kNone = 0xffffffffu;

    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (LPR - 1), grp = lane / LPR;
    const uint32_t row_bytes = chunks * 16u;
    const uint32_t out_stride = chunks * Ops::kFloatsPerLane;
    const bool live = sub < chunks;
    const uint32_t step = gridDim.x;
    const uint32_t bag_in_tile = wave * BPW + grp;

    auto advance = [&](uint32_t &di, uint32_t &tile) {          // scalar: next work item of this workgroup
        tile += step;
        while (tile >= max_tiles && di < n_descs) {
            tile -= max_tiles;
            di++;
        }
    };
    // bounds of this lane group's bag in item (di, tile): loads issued here, consumed an iteration later
    auto open = [&](uint32_t di, uint32_t tile, uint32_t &bag, Pos &p, Pos &e) {
        bag = kNone;
        p = e = 0;
        if (di >= n_descs) return;
        const DevDesc *dp = descs + di;
        const uint64_t n_bags = dp->n_bags;
        const uint32_t b = tile * BAGS_PER_TILE + bag_in_tile;
        if (tile >= dp->n_tiles || b >= n_bags) return;
        const uint64_t n_idx = dp->n_idx;
        const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
        uint64_t p64, e64;
        if (offsets != nullptr) {
            p64 = (uint64_t)load_meta_global<Cfg::kNtMeta>(offsets + b);
            e64 = (b + 1 < n_bags) ? (uint64_t)load_meta_global<Cfg::kNtMeta>(offsets + b + 1) : n_idx;
        } else {
            p64 = (uint64_t)b * dp->fixed_pooling;
            e64 = p64 + dp->fixed_pooling;
        }
        if (Cfg::kClamp) {
            if (e64 > n_idx) e64 = n_idx;
            if (p64 > e64) p64 = e64;
        }
        bag = b;
        p = (Pos)p64;
        e = (Pos)e64;
    };
    // the lane's share of the index window [p, p + W) of descriptor di
    auto window = [&](uint32_t di, Pos p, Pos e, IdxT (&mine)[IPL]) {
#pragma unroll
        for (uint32_t i = 0; i < IPL; i++) mine[i] = 0;
        if (p >= e) return;                                     // (also: no such item)
        const IdxT *__restrict__ indices = static_cast<const IdxT *>(descs[di].indices);
        const uint32_t cnt = (e - p < W) ? (uint32_t)(e - p) : W;
#pragma unroll
        for (uint32_t i = 0; i < IPL; i++)
            if (sub + i * LPR < cnt) mine[i] = load_meta_global<Cfg::kNtMeta>(indices + p + sub + i * LPR);
    };

    uint32_t di0 = 0, tile0 = blockIdx.x;
    while (tile0 >= max_tiles && di0 < n_descs) {
        tile0 -= max_tiles;
        di0++;
    }
    uint32_t di1 = di0, tile1 = tile0;
    advance(di1, tile1);
    uint32_t di2 = di1, tile2 = tile1;
    uint32_t bag0, bag1, bag2;
    Pos p0, e0, p1, e1, p2, e2;
    open(di0, tile0, bag0, p0, e0);
    open(di1, tile1, bag1, p1, e1);
    IdxT mine[IPL], mine_nxt[IPL];
    window(di0, p0, e0, mine);
    // descriptor fields of the CURRENT item live in scalars loaded one item ahead (a scalar load that is waited for
    // while no gather is in flight is dead time for the whole wavefront)
    const char *weights0 = nullptr;
    float *out0 = nullptr;
    uint64_t last_row0 = 0;
    if (di0 < n_descs) {
        weights0 = static_cast<const char *>(descs[di0].weights);
        out0 = descs[di0].out;
        last_row0 = descs[di0].nr_rows - 1;
    }

    while (di0 < n_descs) {                                      // uniform over the workgroup
        // Every slot of the ring is loaded UNCONDITIONALLY (a conditional store into the loop-carried ring makes the
        // compiler copy the whole ring at every branch: 254 registers): window positions past the bag's end name
        // row 0 (their index register is 0), lanes past the row's last piece read piece 0; neither is ever added
        // or stored.
        const char *__restrict__ wsub = weights0 + (live ? sub : 0u) * 16u;
        const uint64_t last_row = last_row0;
        auto fetch = [&](uint64_t r) -> u32x4 {
            return load_row_global<Cfg::kNtRow>(wsub + clamp_row<Cfg::kClamp, IdxT>(r, last_row) * row_bytes);
        };
        // ---- A. ring prologue of this item's first window: U gathers in flight before anything else is waited for
        typename Ops::Acc acc = Ops::zero();
        u32x4 v[U];
        const uint32_t cnt0 = (e0 - p0 < W) ? (uint32_t)(e0 - p0) : W;     // 0: empty or no bag -> nothing is added
#pragma unroll
        for (uint32_t j = 0; j < U; j++)
            v[j] = fetch(shfl_index<IdxT>(mine[(IPL == 1) ? 0 : j / LPR], grp * LPR + j % LPR));
        __builtin_amdgcn_sched_barrier(0);

        // ---- B. prefetch behind the gathers: next item's descriptor scalars and index window (its bounds arrived an
        //         iteration ago), bounds of the item after it
        const char *weights1 = weights0;
        float *out1 = out0;
        uint64_t last_row1 = last_row0;
        if (di1 < n_descs) {
            weights1 = static_cast<const char *>(descs[di1].weights);
            out1 = descs[di1].out;
            last_row1 = descs[di1].nr_rows - 1;
        }
        window(di1 < n_descs ? di1 : di0, p1, e1, mine_nxt);
        advance(di2, tile2);
        open(di2, tile2, bag2, p2, e2);
        __builtin_amdgcn_sched_barrier(0);

        // ---- C. the rest of the first window, rolling: add the oldest slot, re-issue it
        uint32_t k0 = U;
#pragma unroll 1
        for (; k0 < cnt0; k0 += U) {                            // (kept a loop: unrolled, the scheduler hoists every gather)
#pragma unroll
            for (uint32_t j = 0; j < U; j++) {
                const uint32_t q = k0 + j;
                const uint64_t r = shfl_index<IdxT>(mine[(IPL == 1) ? 0 : (q / LPR) % IPL], grp * LPR + q % LPR);
                Ops::add(acc, v[j]);                            // position q - U < cnt: oldest first, index order
                v[j] = fetch(r);
                __builtin_amdgcn_sched_barrier(0);              // keep add-then-reissue per slot: the scheduler would
            }                                                   // otherwise drain all U gathers and re-issue them together
        }
#pragma unroll
        for (uint32_t j = 0; j < U; j++) {
            typename Ops::Acc with = acc;
            Ops::add(with, v[j]);
            acc = (k0 - U + j < cnt0) ? with : acc;
        }
        // ---- bags longer than one window (rare): the remaining windows on demand, same ring
        Pos p = p0 + cnt0;
        while (p < e0) {                                        // uniform over a lane group
            const uint32_t cnt = (e0 - p < W) ? (uint32_t)(e0 - p) : W;
            const IdxT *__restrict__ indices = static_cast<const IdxT *>(descs[di0].indices);
#pragma unroll
            for (uint32_t i = 0; i < IPL; i++) {
                mine[i] = 0;
                if (sub + i * LPR < cnt) mine[i] = load_meta_global<Cfg::kNtMeta>(indices + p + sub + i * LPR);
            }
#pragma unroll
            for (uint32_t j = 0; j < U; j++)
                v[j] = fetch(shfl_index<IdxT>(mine[(IPL == 1) ? 0 : j / LPR], grp * LPR + j % LPR));
            uint32_t k1 = U;
#pragma unroll 1
            for (; k1 < cnt; k1 += U) {
#pragma unroll
                for (uint32_t j = 0; j < U; j++) {
                    const uint32_t q = k1 + j;
                    const uint64_t r = shfl_index<IdxT>(mine[(IPL == 1) ? 0 : (q / LPR) % IPL], grp * LPR + q % LPR);
                    Ops::add(acc, v[j]);
                    v[j] = fetch(r);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (uint32_t j = 0; j < U; j++) {
                typename Ops::Acc with = acc;
                Ops::add(with, v[j]);
                acc = (k1 - U + j < cnt) ? with : acc;
            }
            p += cnt;
        }
        if (bag0 != kNone || Ops::kGroupStore)
            store_row<Ops, Cfg, LPR>(acc, out0 + (uint64_t)bag0 * out_stride, sub, grp, chunks, bag0 != kNone && live);

        // ---- rotate
        di0 = di1; tile0 = tile1; bag0 = bag1; p0 = p1; e0 = e1;
        di1 = di2; tile1 = tile2; bag1 = bag2; p1 = p2; e1 = e2;
        weights0 = weights1; out0 = out1; last_row0 = last_row1;
#pragma unroll
        for (uint32_t i = 0; i < IPL; i++) mine[i] = mine_nxt[i];
    }
}

// ---- v5 (round-5 experiment): the table's K HOTTEST rows in REGISTERS ------------------------------------------------
// Zipf(1.2) over 10 M rows names its hottest row in 18.5 % of all gathers, the top two in 26.5 %, the top four in 35 %.  Those
// gathers hit the L1, but an L1 hit still goes through the texture-address unit and the L1's 64 B/clk -- the two things the
// pooled Zipf launch is bound by.  Here every lane keeps its 16-byte piece of the K hottest rows of
// the descriptor's table in 4 K registers (row ids in DevDesc::pad_[2..3], rows = the first K of the compact hot copy the
// engine already keeps for the LDS path); a gather whose row id equals one of them is a register select, its lanes are
// masked out of the load.  Same bits: the register copy holds the table's own bits, the adds stay in index order.
template <typename IdxT, int DT, int LPR, class Cfg, int K>
__global__ void __launch_bounds__(Cfg::kBlock)
bag_sum_reghot_kernel(const DevDesc *__restrict__ descs, uint32_t chunks_arg, const uint32_t *__restrict__ xmap) {
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
    if (tile >= n_tiles) return;
    const uint64_t bag = (uint64_t)tile * BAGS_PER_TILE + wave * BPW + grp;
    const bool live = sub < chunks;
    if (bag >= n_bags) return;
    // the K hottest rows of this table: ids (0xffffffff = none) and this lane's piece of each
    uint32_t hid[K];
    u32x4 hrow[K];
    const uint32_t *ids = reinterpret_cast<const uint32_t *>(&dp->pad_[2]);
    const u32x4 *hot = static_cast<const u32x4 *>(dp->hot_rows);
#pragma unroll
    for (int k = 0; k < K; k++) {
        hid[k] = ((uint32_t)k < dp->n_hot) ? ids[k] : 0xffffffffu;
        hrow[k] = (hid[k] != 0xffffffffu && live) ? hot[(uint32_t)k * chunks + sub] : u32x4{0u, 0u, 0u, 0u};
    }
    uint64_t p, e;
    if (offsets != nullptr) {
        p = (uint64_t)load_meta<Cfg::kNtMeta>(offsets + bag);
        e = (bag + 1 < n_bags) ? (uint64_t)load_meta<Cfg::kNtMeta>(offsets + bag + 1) : n_idx;
    } else {
        p = bag * fixed_pooling;
        e = p + fixed_pooling;
    }
    typename Ops::Acc acc = Ops::zero();
    walk_bag<IdxT, LPR, Cfg, Ops>(indices, p, e, sub, grp, live, acc, NoProbe{},
                                  [&](uint64_t r, uint32_t) -> u32x4 {
                                      u32x4 v = {0u, 0u, 0u, 0u};
                                      bool hit = false;
#pragma unroll
                                      for (int k = 0; k < K; k++)
                                          if ((uint32_t)r == hid[k]) {
                                              v = hrow[k];
                                              hit = true;
                                          }
                                      if (!hit) v = load_row<Cfg::kNtRow>(wsub + clamp_row<Cfg::kClamp, IdxT>(r, last_row) * row_bytes);
                                      return v;
                                  });
    store_row<Ops, Cfg, LPR>(acc, out + bag * out_stride, sub, grp, chunks, live);
}

}  // namespace pimemb

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

#ifndef TUNE_DIM
#define TUNE_DIM 128
#endif
constexpr int D = TUNE_DIM, LPR = TUNE_DIM / 4;   // -DTUNE_DIM=64: the reference's NR_COLS=64 presets

__global__ void fill_table(float *w, uint64_t n, uint32_t seed) {
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        w[i] = (float)(int32_t)h * (1.0f / 2147483648.0f);
    }
}

static uint64_t rng_state = 88172645463325252ull;
static inline uint64_t xorshift() {
    rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
    return rng_state;
}
static inline double urand() { return (double)(xorshift() >> 11) * (1.0 / 9007199254740992.0); }

__global__ void copy_rows(u32x4 *dst, const u32x4 *table, const uint64_t *ids, uint32_t n, uint32_t chunks) {
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n * chunks; i += gridDim.x * blockDim.x)
        dst[i] = table[ids[i / chunks] * chunks + i % chunks];
}

using LaunchFn = void (*)(const DevDesc *, uint32_t, uint32_t, const uint32_t *, uint32_t, hipStream_t);
struct Variant {
    std::string name;
    uint32_t bags_per_tile = 256;
    LaunchFn fn = nullptr;
    bool xcd = false;
    uint64_t cacheable = kXcdCacheableBytes;
    uint32_t hot = 0;        // hot rows per table staged in LDS (0 = not the hot kernel)
    bool reghot = false;     // ... kept in registers instead (bag_sum_reghot_kernel): ordinary grid, no LDS
    uint32_t wgs = 0;        // persistent workgroups per table (hot kernel)
    size_t lds = 0;
    uint32_t *d_xmap = nullptr;
    uint32_t xgrid = 0;
    std::vector<float> us;
};

template <class Cfg, bool WAVEBATCH>
void do_launch(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *xmap, uint32_t xgrid, hipStream_t s) {
    dim3 grid = xmap ? dim3(xgrid, 1, 1) : dim3(tiles, n, 1), block(Cfg::kBlock, 1, 1);
    if (WAVEBATCH)
        hipLaunchKernelGGL((bag_sum_wavebatch_kernel<uint32_t, EMB_F32, LPR, Cfg>), grid, block, 0, s, d, (uint32_t)LPR, xmap);
    else
        hipLaunchKernelGGL((bag_sum_group_kernel<uint32_t, EMB_F32, LPR, Cfg>), grid, block, 0, s, d, (uint32_t)LPR, xmap);
}
static size_t g_lds = 0;
static uint32_t g_wgs = 0;
template <class Cfg>
void do_launch_hot(const DevDesc *d, uint32_t n, uint32_t, const uint32_t *, uint32_t, hipStream_t s) {
    hipLaunchKernelGGL((bag_sum_hot_kernel<uint32_t, EMB_F32, LPR, Cfg>), dim3(g_wgs, n, 1), dim3(Cfg::kBlock), g_lds, s,
                       d, (uint32_t)LPR);
}
template <class Cfg>
Variant make_hot(const char *name, uint32_t hot, uint32_t wgs) {
    Variant v;
    v.name = name;
    v.bags_per_tile = (64u / LPR) * (Cfg::kBlock / 64);
    v.fn = &do_launch_hot<Cfg>;
    v.hot = hot;
    v.wgs = wgs;
    return v;
}

template <class Cfg, int K>
void do_launch_reghot(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *xmap, uint32_t xgrid, hipStream_t s) {
    dim3 grid = xmap ? dim3(xgrid, 1, 1) : dim3(tiles, n, 1), block(Cfg::kBlock, 1, 1);
    hipLaunchKernelGGL((bag_sum_reghot_kernel<uint32_t, EMB_F32, LPR, Cfg, K>), grid, block, 0, s, d, (uint32_t)LPR, xmap);
}
template <class Cfg, int K>
Variant make_reghot(const char *name) {
    Variant v;
    v.name = name;
    v.bags_per_tile = (64u / LPR) * (Cfg::kBlock / 64);
    v.fn = &do_launch_reghot<Cfg, K>;
    v.hot = K;
    v.reghot = true;
    return v;
}

template <class Cfg, int G>
void do_launch_stream(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *, uint32_t, hipStream_t s) {
    hipLaunchKernelGGL((bag_sum_stream_kernel<uint32_t, EMB_F32, LPR, Cfg>), dim3(G), dim3(Cfg::kBlock), 0, s, d, (uint32_t)LPR, n, tiles);
}
template <class Cfg, int G>
Variant make_stream(const char *name) {
    Variant v;
    v.name = name;
    v.bags_per_tile = (64u / LPR) * (Cfg::kBlock / 64);
    v.fn = &do_launch_stream<Cfg, G>;
    return v;
}

template <class Cfg, bool WAVEBATCH>
Variant make_variant(const char *name, bool xcd = false) {
    Variant v;
    v.name = name;
    v.xcd = xcd;
    v.bags_per_tile = WAVEBATCH ? 64u * Cfg::kBatches * (Cfg::kBlock / 64) : (64u / LPR) * (Cfg::kBlock / 64);
    v.fn = &do_launch<Cfg, WAVEBATCH>;
    return v;
}

int main(int argc, char **argv) {
    uint32_t T = argc > 1 ? atoi(argv[1]) : 16;
    uint64_t rows = argc > 2 ? strtoull(argv[2], 0, 10) : 4000000ull;
    uint32_t B = argc > 3 ? atoi(argv[3]) : 16384;
    uint32_t L = argc > 4 ? atoi(argv[4]) : 32;
    double alpha = argc > 5 ? atof(argv[5]) : 1.2;
    int NB = argc > 6 ? atoi(argv[6]) : 2;
    int rounds = argc > 7 ? atoi(argv[7]) : 5;
    int iters = argc > 8 ? atoi(argv[8]) : 6;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    printf("pooled workload: %u tables x %llu rows x dim %d fp32 (%.1f GB), B=%u, L=%u, zipf %.2f, %d batches\n", T,
           (unsigned long long)rows, D, T * rows * D * 4 / 1e9, B, L, alpha, NB);

    std::vector<float *> tables(T);
    for (uint32_t t = 0; t < T; t++) {
        CK(hipMalloc((void **)&tables[t], rows * D * 4));
        hipLaunchKernelGGL(fill_table, dim3(4096), dim3(256), 0, s, tables[t], rows * D, t + 1);
    }
    const uint64_t n_idx = (uint64_t)B * L;
    std::vector<std::vector<uint32_t *>> d_idx(NB, std::vector<uint32_t *>(T)), d_off(NB, std::vector<uint32_t *>(T));
    std::vector<std::vector<float *>> d_out(NB, std::vector<float *>(T));
    std::vector<uint32_t> off(B), h(n_idx);
    for (uint32_t i = 0; i < B; i++) off[i] = i * L;
    const double oma = 1.0 - alpha, nn = std::pow((double)rows, oma) - 1.0;
    for (int b = 0; b < NB; b++)
        for (uint32_t t = 0; t < T; t++) {
            for (uint64_t i = 0; i < n_idx; i++) {
                uint64_t rank;
                if (alpha <= 0.0) rank = xorshift() % rows;          // uniform
                else rank = (uint64_t)std::pow(nn * urand() + 1.0, 1.0 / oma) - 1;  // bounded Pareto ~ Zipf
                if (rank >= rows) rank = rows - 1;
                h[i] = (uint32_t)((rank * 2654435761ull + 12345ull + t) % rows);     // scatter hot rows
            }
            CK(hipMalloc((void **)&d_idx[b][t], n_idx * 4));
            CK(hipMemcpy(d_idx[b][t], h.data(), n_idx * 4, hipMemcpyHostToDevice));
            CK(hipMalloc((void **)&d_off[b][t], B * 4));
            CK(hipMemcpy(d_off[b][t], off.data(), B * 4, hipMemcpyHostToDevice));
            CK(hipMalloc((void **)&d_out[b][t], (size_t)B * D * 4));
        }

    std::vector<Variant> vars;
    //                        BLOCK U  ntS   ntM  inflight minW batches ntRow spec  idxShuffle
    vars.push_back(make_variant<BagCfg<256, 8, true, false, 8, 1, 1, false, false, true>, false>("v1 group SHIP (blk256 U8 idxshfl)"));
    if (getenv("TUNE_SMALL")) {   // small pooled launches (reference presets): gathers in flight per lane x workgroup size
        vars.push_back(make_variant<BagCfg<256, 16, true, false, 8, 1, 1, false, false, true>, false>("v1 group blk256 U16"));
        vars.push_back(make_variant<BagCfg<256, 32, true, false, 8, 1, 1, false, false, true>, false>("v1 group blk256 U32"));
        vars.push_back(make_variant<BagCfg<64, 8, true, false, 8, 1, 1, false, false, true>, false>("v1 group blk64 U8"));
        vars.push_back(make_variant<BagCfg<64, 16, true, false, 8, 1, 1, false, false, true>, false>("v1 group blk64 U16"));
        vars.push_back(make_variant<BagCfg<64, 32, true, false, 8, 1, 1, false, false, true>, false>("v1 group blk64 U32"));
        vars.push_back(make_variant<BagCfg<128, 16, true, false, 8, 1, 1, false, false, true>, false>("v1 group blk128 U16"));
    }
    //                        BLOCK U  ntS   ntM  inflight minW
    if (getenv("TUNE_STREAM")) {   // the rejected persistent / prefetching / rolling-ring variant (see the note at the top)
        vars.push_back(make_stream<BagCfg<256, 8, true, false, 8, 8>, 2048>("v4 stream blk256 U8 minw8 G2048"));
        vars.push_back(make_stream<BagCfg<256, 8, true, false, 8, 8>, 4096>("v4 stream blk256 U8 minw8 G4096"));
        vars.push_back(make_stream<BagCfg<256, 8, true, false, 8, 8>, 1024>("v4 stream blk256 U8 minw8 G1024"));
        vars.push_back(make_stream<BagCfg<256, 8, true, false, 8, 1>, 2048>("v4 stream blk256 U8 minw1 G2048"));
        vars.push_back(make_stream<BagCfg<256, 16, true, false, 8, 1>, 2048>("v4 stream blk256 U16 minw1 G2048"));
        vars.push_back(make_stream<BagCfg<256, 16, true, false, 8, 1>, 1280>("v4 stream blk256 U16 minw1 G1280"));
        vars.push_back(make_stream<BagCfg<64, 8, true, false, 8, 8>, 8192>("v4 stream blk64 U8 minw8 G8192"));
    }
    if (getenv("TUNE_HOT")) {      // LDS hot rows, and cache policy of the rows that MISS the hot set
        const uint32_t wg = std::max(1u, 8192u / T);
        //                              BLOCK U  ntS   ntM  inflight minW batches ntRow  spec   idxShuffle
        vars.push_back(make_variant<BagCfg<256, 8, true, false, 8, 1, 1, true, false, true>, false>("v1 group, ALL rows nt"));
        vars.push_back(make_hot<BagCfg<1024, 8, true, false, 8, 1, 1, false, false, true>>("v3 hot100 SHIP", 100, wg));
        vars.push_back(make_hot<BagCfg<1024, 8, true, false, 8, 1, 1, true, false, true>>("v3 hot100, misses nt", 100, wg));
        vars.push_back(make_hot<BagCfg<1024, 8, true, false, 8, 1, 1, false, false, true>>("v3 hot32", 32, wg));
        vars.push_back(make_hot<BagCfg<1024, 8, true, false, 8, 1, 1, true, false, true>>("v3 hot32, misses nt", 32, wg));
        vars.push_back(make_hot<BagCfg<512, 8, true, false, 8, 1, 1, false, false, true>>("v3 hot100 blk512", 100, 2 * wg));
        vars.push_back(make_hot<BagCfg<512, 8, true, false, 8, 1, 1, true, false, true>>("v3 hot100 blk512, misses nt", 100, 2 * wg));
    }
    if (getenv("TUNE_REGHOT")) {   // round 5: the K hottest rows in registers
        using G8 = BagCfg<256, 8, true, false, 8, 1, 1, false, false, true>;
        vars.push_back(make_reghot<G8, 1>("v5 reghot K=1"));
        vars.push_back(make_reghot<G8, 2>("v5 reghot K=2"));
        vars.push_back(make_reghot<G8, 4>("v5 reghot K=4"));
        vars.push_back(make_reghot<BagCfg<256, 8, true, false, 8, 1, 1, false, false, false>, 2>("v5 reghot K=2 no idxshfl"));
        vars.push_back(make_hot<BagCfg<1024, 8, true, false, 8, 1, 1, false, false, true>>("v3 hot32 (LDS) for reference", 32, std::max(1u, 8192u / T)));
    }
    if (getenv("TUNE_HOT_SWEEP")) {   // how many hot rows, how many persistent workgroups
        static char names[64][64];
        int ni = 0;
        for (uint32_t hot : {8u, 16u, 32u, 64u})
            for (uint32_t total : {4096u, 8192u, 16384u}) {
                snprintf(names[ni], 64, "v3 blk1024 hot%u wgs%u", hot, total);
                vars.push_back(make_hot<BagCfg<1024, 8, true, false, 8, 1, 1, false, false, true>>(names[ni++], hot, std::max(1u, total / T)));
            }
        for (uint32_t hot : {16u, 32u})
            for (uint32_t total : {8192u, 16384u, 32768u}) {
                snprintf(names[ni], 64, "v3 blk256 hot%u wgs%u", hot, total);
                vars.push_back(make_hot<BagCfg<256, 8, true, false, 8, 1, 1, false, false, true>>(names[ni++], hot, std::max(1u, total / T)));
            }
    }
    if (!getenv("TUNE_ALL")) goto build_done;
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 8, 8, 1, false, true>, true>("v2 wave b1 SHIP (blk64 U8 minw8)"));
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 8, 8, 1, false, true>, true>("v2 wave b1 SHIP XCD", true));
    vars.push_back(make_variant<BagCfg<128, 4, true, false, 8, 8, 2, false, true>, true>("v2 wave b2 SHIP (blk128 U4 minw8)"));
    vars.push_back(make_variant<BagCfg<128, 4, true, false, 8, 8, 2, false, true>, true>("v2 wave b2 SHIP XCD", true));
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 16, 4, 1, false, true>, true>("v2 wave b1 inflight16 minw4"));
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 4, 8, 1, false, true>, true>("v2 wave b1 inflight4 minw8"));

build_done:
    // hot sets: the generator maps Zipf rank k of table t to row (k*2654435761 + 12345 + t) % rows
    std::vector<std::vector<void *>> hot_dev(vars.size(), std::vector<void *>(T, nullptr));
    std::vector<std::vector<uint64_t *>> hash_dev(vars.size(), std::vector<uint64_t *>(T, nullptr));
    std::vector<std::vector<uint32_t>> hot_n(vars.size(), std::vector<uint32_t>(T, 0)), hot_log2(vars.size(), std::vector<uint32_t>(T, 0));
    for (size_t v = 0; v < vars.size(); v++) {
        if (!vars[v].hot) continue;
        for (uint32_t t = 0; t < T; t++) {
            std::vector<uint64_t> ids(vars[v].hot);
            for (uint32_t k = 0; k < vars[v].hot; k++) ids[k] = ((uint64_t)k * 2654435761ull + 12345ull + t) % rows;
            HotSet hs = build_hot_set(ids.data(), vars[v].hot, rows, D * 4, 62 * 1024);
            if (vars[v].reghot) {          // registers: no hash, the compact copy = the K hottest rows in rank order
                hs.rows.assign(ids.begin(), ids.end());
                hs.hash.assign(2, ~0ull);
                hs.log2size = 1;
            }
            hot_n[v][t] = (uint32_t)hs.rows.size();
            hot_log2[v][t] = hs.log2size;
            vars[v].lds = std::max(vars[v].lds, hs.lds_bytes(D * 4));
            uint64_t *d_ids;
            CK(hipMalloc((void **)&d_ids, hs.rows.size() * 8));
            CK(hipMemcpy(d_ids, hs.rows.data(), hs.rows.size() * 8, hipMemcpyHostToDevice));
            CK(hipMalloc(&hot_dev[v][t], hs.rows.size() * D * 4));
            hipLaunchKernelGGL(copy_rows, dim3(64), dim3(256), 0, s, (u32x4 *)hot_dev[v][t], (const u32x4 *)tables[t], d_ids,
                               (uint32_t)hs.rows.size(), (uint32_t)LPR);
            CK(hipMalloc((void **)&hash_dev[v][t], hs.hash.size() * 8));
            CK(hipMemcpy(hash_dev[v][t], hs.hash.data(), hs.hash.size() * 8, hipMemcpyHostToDevice));
        }
        printf("variant %-28s hot rows accepted %u of %u, LDS %zu B\n", vars[v].name.c_str(), hot_n[v][0], vars[v].hot, vars[v].lds);
    }
    CK(hipStreamSynchronize(s));
    std::vector<std::vector<DevDesc *>> d_desc(vars.size(), std::vector<DevDesc *>(NB));
    std::vector<uint32_t> tiles(vars.size());
    for (size_t v = 0; v < vars.size(); v++) {
        tiles[v] = (B + vars[v].bags_per_tile - 1) / vars[v].bags_per_tile;
        for (int b = 0; b < NB; b++) {
            std::vector<DevDesc> hd(T);
            for (uint32_t t = 0; t < T; t++) {
                hd[t] = DevDesc{};
                hd[t].weights = tables[t];
                hd[t].indices = d_idx[b][t];
                hd[t].offsets = d_off[b][t];
                hd[t].out = d_out[b][t];
                hd[t].n_idx = n_idx;
                hd[t].n_bags = B;
                hd[t].nr_rows = rows;
                hd[t].n_tiles = tiles[v];
                if (vars[v].reghot) {
                    uint32_t *ids32 = reinterpret_cast<uint32_t *>(&hd[t].pad_[2]);
                    for (uint32_t k = 0; k < 4; k++)
                        ids32[k] = k < vars[v].hot ? (uint32_t)(((uint64_t)k * 2654435761ull + 12345ull + t) % rows) : 0xffffffffu;
                }
                if (vars[v].hot) {
                    hd[t].hot_rows = hot_dev[v][t];
                    hd[t].hot_hash = hash_dev[v][t];
                    hd[t].n_hot = hot_n[v][t];
                    hd[t].hot_log2 = hot_log2[v][t];
                }
            }
            CK(hipMalloc((void **)&d_desc[v][b], sizeof(DevDesc) * T));
            CK(hipMemcpy(d_desc[v][b], hd.data(), sizeof(DevDesc) * T, hipMemcpyHostToDevice));
        }
        if (vars[v].xcd) {
            std::vector<uint32_t> nt(T, tiles[v]), words;
            std::vector<uint64_t> bytes(T, rows * D * 4);
            vars[v].xgrid = build_xcd_map(nt, bytes, &words, 1, vars[v].cacheable);
            CK(hipMalloc((void **)&vars[v].d_xmap, words.size() * 4));
            CK(hipMemcpy(vars[v].d_xmap, words.data(), words.size() * 4, hipMemcpyHostToDevice));
        }
    }
    CK(hipDeviceSynchronize());

    size_t out_bytes = (size_t)B * D * 4;
    std::vector<char> ref(out_bytes * T), got(out_bytes * T);
    for (size_t v = 0; v < vars.size(); v++) {
        for (uint32_t t = 0; t < T; t++) CK(hipMemsetAsync(d_out[0][t], 0xff, out_bytes, s));
        g_lds = vars[v].lds; g_wgs = vars[v].wgs;
        vars[v].fn(d_desc[v][0], T, tiles[v], vars[v].d_xmap, vars[v].xgrid, s);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(s));
        for (uint32_t t = 0; t < T; t++)
            CK(hipMemcpy((v == 0 ? ref : got).data() + t * out_bytes, d_out[0][t], out_bytes, hipMemcpyDeviceToHost));
        if (v && memcmp(ref.data(), got.data(), out_bytes * T)) {
            fprintf(stderr, "MISMATCH variant %s\n", vars[v].name.c_str());
            return 2;
        }
    }
    printf("all variants bit-identical on batch 0\n");

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = -1; r < rounds; r++)
        for (size_t v = 0; v < vars.size(); v++) {
            g_lds = vars[v].lds; g_wgs = vars[v].wgs;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; i++) vars[v].fn(d_desc[v][i % NB], T, tiles[v], vars[v].d_xmap, vars[v].xgrid, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0) vars[v].us.push_back(ms * 1000.f / iters);
        }
    const double alg = (double)T * ((double)n_idx * (D * 4 + 4) + (double)B * 4 + (double)B * D * 4);
    printf("%-34s %10s %10s %9s %12s\n", "variant", "min us", "med us", "TB/s@med", "Gbags/s@med");
    for (auto &v : vars) {
        std::sort(v.us.begin(), v.us.end());
        float med = v.us[v.us.size() / 2];
        printf("%-34s %10.1f %10.1f %9.2f %12.3f\n", v.name.c_str(), v.us[0], med, alg / (med * 1e-6) / 1e12,
               (double)T * B / (med * 1e-6) / 1e9);
    }
    return 0;
}
