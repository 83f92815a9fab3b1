// pimemb_kernels.hip -- the device side of the hot path, written for gfx950 (MI355X, CDNA4).
//
// Replaces the UPMEM DPU program upmem/src/dpu/emb_dpu_lookup.c:36-138 (one DPU per (table,
// column), 14 tasklets striding over bags, one 8-byte MRAM read per (index, column)) and the host
// post-process upmem/include/emb_host.h:186-222 with ONE fused launch over all tables:
//
//   out[t][b][:] = sum_{p = off_t[b]}^{end-1} W_t[idx_t[p]][:]      end = off_t[b+1] | n_idx_t
//
// Mapping to the machine (bandwidth-bound indexing; no MFMA, there is no contraction here):
//   * rows stay row-major [nr_rows][dim] in HBM; a row is read as 16-byte pieces, one piece per
//     lane, so the LPR = row_bytes/16 lanes of a "lane group" fetch one whole row with a single
//     coalesced global_load_dwordx4 (64 B for dim 16 fp32, 512 B for dim 128 fp32);
//   * a 64-lane wavefront therefore works on 64/LPR bags at once, a 256-thread workgroup on
//     4 * 64/LPR bags per tile; blockIdx.y picks the table descriptor (scalar loads), blockIdx.x
//     strides over that table's tiles;
//   * every output element is accumulated by ONE lane in index order, starting from +0 -- the
//     same order as a sequential CPU EmbeddingBag, so fp32 results are bit-identical to the
//     oracle for any pooling factor (no cross-lane tree whose rounding would differ);
//   * UNROLL independent row loads are kept in flight per lane before the ordered adds, which is
//     what hides HBM latency for long bags; one-hot bags (Kaggle, L=1) rely on occupancy.
//   * the fixed-point mode keeps the reference arithmetic: int32 wrap-around accumulate
//     (emb_dpu_lookup.c:114) and out = (float)acc / 1e9 via double (emb_host.h:210).
#include "pimemb_internal.h"

namespace pimemb {
namespace {

using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
using f32x8 = __attribute__((ext_vector_type(8))) float;

constexpr int kBlock = 256;  // 4 wavefronts
constexpr int kWaves = kBlock / 64;
constexpr int kUnroll = 8;   // row loads in flight per lane inside a bag

// ---- per-dtype accumulate / store -----------------------------------------------------------
template <int DT>
struct RowOps;

template <>
struct RowOps<EMB_F32> {
    using Acc = f32x4;
    static constexpr uint32_t kFloatsPerLane = 4;
    static __device__ __forceinline__ Acc zero() { return Acc{0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void add(Acc &a, u32x4 raw) {
        a += __builtin_bit_cast(f32x4, raw);
    }
    static __device__ __forceinline__ void store(const Acc &a, float *dst) {
        *reinterpret_cast<f32x4 *>(dst) = a;
    }
};

template <>
struct RowOps<EMB_F16> {
    using Acc = f32x8;
    static constexpr uint32_t kFloatsPerLane = 8;
    static __device__ __forceinline__ Acc zero() { return Acc{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}; }
    static __device__ __forceinline__ void add(Acc &a, u32x4 raw) {
        a += __builtin_convertvector(__builtin_bit_cast(f16x8, raw), f32x8);
    }
    static __device__ __forceinline__ void store(const Acc &a, float *dst) {
        f32x4 lo = {a[0], a[1], a[2], a[3]}, hi = {a[4], a[5], a[6], a[7]};
        reinterpret_cast<f32x4 *>(dst)[0] = lo;
        reinterpret_cast<f32x4 *>(dst)[1] = hi;
    }
};

template <>
struct RowOps<EMB_FIXED32> {
    using Acc = u32x4;  // unsigned add == int32 two's-complement wrap (emb_dpu_lookup.c:114)
    static constexpr uint32_t kFloatsPerLane = 4;
    static __device__ __forceinline__ Acc zero() { return Acc{0u, 0u, 0u, 0u}; }
    static __device__ __forceinline__ void add(Acc &a, u32x4 raw) { a += raw; }
    static __device__ __forceinline__ float conv(uint32_t acc) {
        // emb_host.h:210: (float)tmp / pow(10,9): int32 -> float, divide in double, round to float
        return (float)((double)(float)(int32_t)acc / 1.0e9);
    }
    static __device__ __forceinline__ void store(const Acc &a, float *dst) {
        f32x4 o = {conv(a[0]), conv(a[1]), conv(a[2]), conv(a[3])};
        *reinterpret_cast<f32x4 *>(dst) = o;
    }
};

// ---- fused multi-table gather + segment-sum -------------------------------------------------
template <typename IdxT, int DT, int LPR>
__global__ void __launch_bounds__(kBlock)
bag_sum_kernel(const DevDesc *__restrict__ descs, uint32_t chunks) {
    using Ops = RowOps<DT>;
    constexpr uint32_t BPW = 64 / LPR;           // bags per wavefront
    constexpr uint32_t BAGS_PER_TILE = BPW * kWaves;

    const DevDesc *dp = descs + blockIdx.y;      // wave-uniform: scalar loads
    const char *__restrict__ weights = static_cast<const char *>(dp->weights);
    const IdxT *__restrict__ indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *__restrict__ offsets = static_cast<const IdxT *>(dp->offsets);
    float *__restrict__ out = dp->out;
    const uint64_t n_idx = dp->n_idx;
    const uint64_t n_bags = dp->n_bags;
    const uint32_t fixed_pooling = dp->fixed_pooling;
    const uint32_t n_tiles = dp->n_tiles;

    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = threadIdx.x >> 6;
    const uint32_t sub = lane & (LPR - 1);       // which 16-byte piece of the row
    const uint32_t grp = lane / LPR;             // which bag of this wavefront
    const uint32_t row_bytes = chunks * 16u;
    const uint32_t out_stride = chunks * Ops::kFloatsPerLane;  // = dim
    const char *__restrict__ wsub = weights + sub * 16u;

    for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const uint64_t bag = (uint64_t)tile * BAGS_PER_TILE + wave * BPW + grp;
        if (bag >= n_bags || sub >= chunks) continue;

        uint64_t p, e;
        if (offsets != nullptr) {
            p = (uint64_t)offsets[bag];
            e = (bag + 1 < n_bags) ? (uint64_t)offsets[bag + 1] : n_idx;  // emb_dpu_lookup.c:109-110
        } else {
            p = bag * fixed_pooling;
            e = p + fixed_pooling;
        }

        typename Ops::Acc acc = Ops::zero();     // empty bag -> 0 (emb_dpu_lookup.c:108)
        for (; p + kUnroll <= e; p += kUnroll) {
            uint64_t r[kUnroll];
#pragma unroll
            for (int k = 0; k < kUnroll; k++) r[k] = (uint64_t)indices[p + k];
            u32x4 v[kUnroll];
#pragma unroll
            for (int k = 0; k < kUnroll; k++)
                v[k] = *reinterpret_cast<const u32x4 *>(wsub + r[k] * row_bytes);
#pragma unroll
            for (int k = 0; k < kUnroll; k++) Ops::add(acc, v[k]);  // index order
        }
        for (; p < e; p++) {
            const uint64_t r = (uint64_t)indices[p];
            Ops::add(acc, *reinterpret_cast<const u32x4 *>(wsub + r * row_bytes));
        }
        Ops::store(acc, out + bag * out_stride + sub * Ops::kFloatsPerLane);
    }
}

template <typename IdxT, int DT>
hipError_t launch_lpr(const DevDesc *d, uint32_t n, uint32_t max_tiles, const LaunchGeom &g,
                      hipStream_t s) {
    dim3 grid(max_tiles, n, 1), block(kBlock, 1, 1);
    switch (g.lanes_per_row) {
#define PIMEMB_CASE(L)                                                                        \
    case L:                                                                                   \
        hipLaunchKernelGGL((bag_sum_kernel<IdxT, DT, L>), grid, block, 0, s, d, g.chunks);    \
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
                        const LaunchGeom &g, hipStream_t s) {
    switch (dtype) {
        case EMB_F32:
            return launch_lpr<IdxT, EMB_F32>(d, n, max_tiles, g, s);
        case EMB_F16:
            return launch_lpr<IdxT, EMB_F16>(d, n, max_tiles, g, s);
        case EMB_FIXED32:
            return launch_lpr<IdxT, EMB_FIXED32>(d, n, max_tiles, g, s);
    }
    return hipErrorInvalidValue;
}

// ---- column scatter for populate_mram-style uploads -----------------------------------------
__global__ void __launch_bounds__(kBlock)
scatter_column_kernel(int32_t *__restrict__ table, const int32_t *__restrict__ column,
                      uint64_t nr_rows, uint32_t dim, uint32_t col) {
    for (uint64_t r = (uint64_t)blockIdx.x * kBlock + threadIdx.x; r < nr_rows;
         r += (uint64_t)gridDim.x * kBlock)
        table[r * dim + col] = column[r];
}

// ---- debug validation ------------------------------------------------------------------------
template <typename IdxT>
__global__ void __launch_bounds__(kBlock)
validate_kernel(const DevDesc *__restrict__ descs, unsigned long long *__restrict__ bad) {
    const DevDesc *dp = descs + blockIdx.y;
    const IdxT *indices = static_cast<const IdxT *>(dp->indices);
    const IdxT *offsets = static_cast<const IdxT *>(dp->offsets);
    const uint64_t n_idx = dp->n_idx, n_bags = dp->n_bags, nr_rows = dp->nr_rows;
    unsigned long long local = 0;
    const uint64_t stride = (uint64_t)gridDim.x * kBlock;
    for (uint64_t i = (uint64_t)blockIdx.x * kBlock + threadIdx.x; i < n_idx; i += stride) {
        const IdxT v = indices[i];
        if (v < 0 || (uint64_t)v >= nr_rows) local++;
    }
    if (offsets != nullptr) {
        for (uint64_t b = (uint64_t)blockIdx.x * kBlock + threadIdx.x; b < n_bags; b += stride) {
            const IdxT o = offsets[b];
            const uint64_t nxt = (b + 1 < n_bags) ? (uint64_t)offsets[b + 1] : n_idx;
            if (o < 0 || (uint64_t)o > nxt || nxt > n_idx) local++;
        }
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        if ((uint64_t)dp->fixed_pooling * n_bags != n_idx) local++;
    }
    if (local) atomicAdd(bad, local);
}

}  // namespace

int geometry_for(emb_dtype dtype, uint32_t dim, LaunchGeom *g) {
    uint32_t elem = (dtype == EMB_F16) ? 2u : 4u;
    if (dtype != EMB_F32 && dtype != EMB_F16 && dtype != EMB_FIXED32) return EMB_ERR_INVALID;
    uint64_t row_bytes = (uint64_t)dim * elem;
    if (dim == 0 || row_bytes % 16 != 0 || row_bytes > 1024) return EMB_ERR_UNSUPPORTED;
    uint32_t chunks = (uint32_t)(row_bytes / 16);
    uint32_t lpr = 1;
    while (lpr < chunks) lpr <<= 1;
    g->lanes_per_row = lpr;
    g->chunks = chunks;
    g->bags_per_tile = (64 / lpr) * kWaves;
    return EMB_OK;
}

hipError_t launch_bag_sum(const DevDesc *d_descs, uint32_t n_descs, uint32_t max_tiles,
                          emb_dtype dtype, emb_index_type itype, const LaunchGeom &g,
                          hipStream_t stream) {
    if (n_descs == 0 || max_tiles == 0) return hipSuccess;
    if (n_descs > 65535u) return hipErrorInvalidValue;
    if (itype == EMB_IDX_U32)
        return launch_dtype<uint32_t>(d_descs, n_descs, max_tiles, dtype, g, stream);
    return launch_dtype<int64_t>(d_descs, n_descs, max_tiles, dtype, g, stream);
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

hipError_t launch_validate(const DevDesc *d_descs, uint32_t n_descs, emb_index_type itype,
                           unsigned long long *d_bad, hipStream_t stream) {
    if (n_descs == 0) return hipSuccess;
    dim3 grid(256, n_descs, 1), block(kBlock, 1, 1);
    if (itype == EMB_IDX_U32)
        hipLaunchKernelGGL(validate_kernel<uint32_t>, grid, block, 0, stream, d_descs, d_bad);
    else
        hipLaunchKernelGGL(validate_kernel<int64_t>, grid, block, 0, stream, d_descs, d_bad);
    return hipGetLastError();
}

}  // namespace pimemb
