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

#include "../pimemb_bag_kernels.h"
#include "../pimemb_xcd_map.h"
#include "../pimemb_hot_rows.h"

using namespace pimemb;

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
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 8, 8, 1, false, true>, true>("v2 wave b1 SHIP (blk64 U8 minw8)"));
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 8, 8, 1, false, true>, true>("v2 wave b1 SHIP XCD", true));
    vars.push_back(make_variant<BagCfg<128, 4, true, false, 8, 8, 2, false, true>, true>("v2 wave b2 SHIP (blk128 U4 minw8)"));
    vars.push_back(make_variant<BagCfg<128, 4, true, false, 8, 8, 2, false, true>, true>("v2 wave b2 SHIP XCD", true));
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 16, 4, 1, false, true>, true>("v2 wave b1 inflight16 minw4"));
    vars.push_back(make_variant<BagCfg<64, 8, true, false, 4, 8, 1, false, true>, true>("v2 wave b1 inflight4 minw8"));

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
