// tune_bag_kernels.hip -- developer tool: interleaved A/B timing of bag-kernel variants on the C2
// workload (26 Criteo-Kaggle tables, dim 16 fp32, B bags/table, one index per bag, u32 indices and
// offsets), all variants in ONE process, rounds interleaved, outputs cross-checked bit for bit.
// Not part of libpimemb.so.  Build: make -f Makefile.tools ; run on the GPU box.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

#include "pimemb_bag_kernels.h"
#include "pimemb_xcd_map.h"

using namespace pimemb;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static const uint64_t kKaggleRows[26] = {1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145,
                                         5683, 8351593, 3194, 27, 14992, 5461306, 10, 5652, 2173, 4,
                                         7046547, 18, 15, 286181, 105, 142572};

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

// ---- ablations (timing only; outputs are NOT the lookup result) -------------------------------
// store-only: every bag's 64-byte pooled row is written, nothing is gathered.
__global__ void __launch_bounds__(256) ablate_store_only(const DevDesc *descs, uint32_t) {
    const DevDesc *dp = descs + blockIdx.y;
    float *out = dp->out;
    const uint64_t n_bags = dp->n_bags;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, sub = lane & 3, grp = lane >> 2;
    const uint64_t bag_base = ((uint64_t)blockIdx.x * 4 + wave) * 64u;
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        const uint64_t bag = bag_base + j * 16 + grp;
        if (bag < n_bags) __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f32x4 *>(out + bag * 16 + sub * 4));
    }
}
// gather-only: bounds + index + row gather as in v2, but one 16-byte store per WAVE batch.
__global__ void __launch_bounds__(256) ablate_gather_only(const DevDesc *descs, uint32_t) {
    const DevDesc *dp = descs + blockIdx.y;
    const char *wsub = static_cast<const char *>(dp->weights) + (threadIdx.x & 3u) * 16u;
    const uint32_t *indices = static_cast<const uint32_t *>(dp->indices);
    const uint32_t *offsets = static_cast<const uint32_t *>(dp->offsets);
    float *out = dp->out;
    const uint64_t n_bags = dp->n_bags, n_idx = dp->n_idx;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, grp = lane >> 2;
    const uint64_t bag_base = ((uint64_t)blockIdx.x * 4 + wave) * 64u;
    if (bag_base >= n_bags) return;
    const uint64_t mb = bag_base + lane;
    uint64_t st = mb < n_bags ? offsets[mb] : n_idx, en = mb + 1 < n_bags ? offsets[mb + 1] : n_idx;
    uint32_t len = (uint32_t)(en - st);
    uint32_t my = len ? indices[st] : 0;
    f32x4 acc = {0, 0, 0, 0};
    u32x4 v[4];
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) {
        uint32_t r = shfl_u32(my, j * 16 + grp);
        v[j] = *reinterpret_cast<const u32x4 *>(wsub + (uint64_t)r * 64u);
    }
#pragma unroll
    for (uint32_t j = 0; j < 4; j++) acc += __builtin_bit_cast(f32x4, v[j]);
    if (acc[0] == 12345.678f) out[bag_base * 16] = acc[1] + acc[2] + acc[3];  // keeps the gathers live
}

using LaunchFn = void (*)(const DevDesc *, uint32_t, uint32_t, const uint32_t *, uint32_t, hipStream_t);

__global__ void __launch_bounds__(256) ablate_null(const DevDesc *descs, uint32_t) {
    if (descs == nullptr) __builtin_trap();
}
struct Variant {
    std::string name;
    uint32_t bags_per_tile = 256;
    LaunchFn fn = nullptr;
    bool xcd = false;            // use the XCD-aware 1-D map (pimemb_xcd_map.h)
    bool direct = false;         // ... expanded to a per-workgroup table
    uint32_t xrounds = 4;
    bool checked = true;         // ablations are timing-only
    bool ranged = false;         // the RANGED instantiation (every descriptor serves a row range; a whole table is the range from row 0)
    bool split_mid = false;      // round-5 experiment: tables of 4 .. 32 MiB as 8 row-range descriptors, one per XCD class, so that
                                 // each XCD's L2 keeps 1/8 of the table's rows across launches (every piece scans all the bags)
    uint32_t n_desc = 26;
    uint32_t *d_xmap = nullptr;
    uint32_t xgrid = 0;
    std::vector<float> us;
};

static uint32_t g_direct_flag = 0;
template <class Cfg, int LPR>
void do_launch_ranged(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *xmap, uint32_t xgrid, hipStream_t s) {
    dim3 grid = xmap ? dim3(xgrid, 1, 1) : dim3(tiles, n, 1), block(Cfg::kBlock, 1, 1);
    hipLaunchKernelGGL((bag_sum_wavebatch_kernel<uint32_t, EMB_F32, LPR, Cfg, true>), grid, block, 0, s, d, (uint32_t)LPR | g_direct_flag, xmap);
}
template <class Cfg, bool WAVEBATCH, int LPR>
void do_launch(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *xmap, uint32_t xgrid, hipStream_t s) {
    dim3 grid = xmap ? dim3(xgrid, 1, 1) : dim3(tiles, n, 1), block(Cfg::kBlock, 1, 1);
    if (WAVEBATCH)
        hipLaunchKernelGGL((bag_sum_wavebatch_kernel<uint32_t, EMB_F32, LPR, Cfg>), grid, block, 0, s, d, (uint32_t)LPR | g_direct_flag, xmap);
    else
        hipLaunchKernelGGL((bag_sum_group_kernel<uint32_t, EMB_F32, LPR, Cfg>), grid, block, 0, s, d, (uint32_t)LPR | g_direct_flag, xmap);
}

template <class Cfg, bool WAVEBATCH, int LPR = 4>
Variant make_variant(const char *name, bool xcd = false) {
    Variant v;
    v.name = name;
    v.xcd = xcd;
    v.bags_per_tile = WAVEBATCH ? 64u * Cfg::kBatches * (Cfg::kBlock / 64) : (64u / LPR) * (Cfg::kBlock / 64);
    v.fn = &do_launch<Cfg, WAVEBATCH, LPR>;
    return v;
}

void launch_store_only(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *, uint32_t, hipStream_t s) {
    hipLaunchKernelGGL(ablate_store_only, dim3(tiles, n, 1), dim3(256), 0, s, d, 0u);
}
void launch_null(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *, uint32_t, hipStream_t s) {
    hipLaunchKernelGGL(ablate_null, dim3(tiles, n, 1), dim3(256), 0, s, d, 0u);
}
void launch_gather_only(const DevDesc *d, uint32_t n, uint32_t tiles, const uint32_t *, uint32_t, hipStream_t s) {
    hipLaunchKernelGGL(ablate_gather_only, dim3(tiles, n, 1), dim3(256), 0, s, d, 0u);
}

int main(int argc, char **argv) {
    const uint32_t T = 26, D = 16;
    uint32_t B = argc > 1 ? (uint32_t)atoi(argv[1]) : 39292;
    int NB = argc > 2 ? atoi(argv[2]) : 8;
    int rounds = argc > 3 ? atoi(argv[3]) : 7;
    int iters = argc > 4 ? atoi(argv[4]) : 48;
    CK(hipSetDevice(0));
    hipStream_t s;
    CK(hipStreamCreate(&s));

    std::vector<float *> tables(T);
    for (uint32_t t = 0; t < T; t++) {
        CK(hipMalloc((void **)&tables[t], kKaggleRows[t] * D * 4));
        hipLaunchKernelGGL(fill_table, dim3(2048), dim3(256), 0, s, tables[t], kKaggleRows[t] * D, t + 1);
    }
    // batches
    std::vector<std::vector<uint32_t *>> d_idx(NB, std::vector<uint32_t *>(T));
    std::vector<std::vector<float *>> d_out(NB, std::vector<float *>(T));
    std::vector<std::vector<uint32_t *>> d_off(NB, std::vector<uint32_t *>(T));
    std::vector<uint32_t> off(B);
    for (uint32_t i = 0; i < B; i++) off[i] = i;
    std::vector<uint32_t> h(B);
    for (int b = 0; b < NB; b++)
        for (uint32_t t = 0; t < T; t++) {
            CK(hipMalloc((void **)&d_off[b][t], B * 4));
            CK(hipMemcpy(d_off[b][t], off.data(), B * 4, hipMemcpyHostToDevice));
            for (uint32_t i = 0; i < B; i++) h[i] = (uint32_t)(xorshift() % kKaggleRows[t]);
            CK(hipMalloc((void **)&d_idx[b][t], B * 4));
            CK(hipMemcpy(d_idx[b][t], h.data(), B * 4, hipMemcpyHostToDevice));
            CK(hipMalloc((void **)&d_out[b][t], (size_t)B * D * 4));
        }

    std::vector<Variant> vars;
    //                        BLOCK U  ntS   ntM  inflight minW batches ntRow spec  idxShuffle clamp
    vars.push_back(make_variant<BagCfg<256, 8, false, false, 8>, false>("v1 group blk256"));
    auto add = [&](Variant v, bool direct) { v.xrounds = 1; v.direct = direct; vars.push_back(v); };
    add(make_variant<BagCfg<128, 4, true, false, 8, 8, 2, false, true>, true>("v2 SHIP b2 XCD segments", true), false);
    add(make_variant<BagCfg<128, 4, true, false, 8, 8, 2, false, true>, true>("v2 SHIP b2 XCD direct", true), true);
    add(make_variant<BagCfg<64, 8, true, false, 8, 8, 1, false, true>, true>("v2 b1 XCD segments", true), false);
    add(make_variant<BagCfg<64, 8, true, false, 8, 8, 1, false, true>, true>("v2 b1 XCD direct", true), true);
    add(make_variant<BagCfg<64, 4, true, false, 8, 8, 2, false, true>, true>("v2 blk64 b2 U4 XCD direct", true), true);
    {   // round 5: the RANGED instantiation of the shipped kernel, whole tables (control) and with the 4 .. 32 MiB tables cut
        // into 8 row ranges pinned to the 8 XCD classes
        using Ship = BagCfg<128, 4, true, false, 8, 8, 2, false, true>;
        Variant c = make_variant<Ship, true>("v2 SHIP RANGED whole tables", true);
        c.fn = &do_launch_ranged<Ship, 4>; c.ranged = true; add(c, true);
        Variant m = make_variant<Ship, true>("v2 SHIP RANGED mid tables x8 rows", true);
        m.fn = &do_launch_ranged<Ship, 4>; m.ranged = true; m.split_mid = true; add(m, true);
    }
    { Variant v; v.name = "ABLATION null kernel same grid"; v.fn = launch_null; v.checked = false; vars.push_back(v); }
    {
        Variant v; v.name = "ABLATION store-only (ntS)"; v.fn = launch_store_only; v.checked = false; vars.push_back(v);
        Variant g; g.name = "ABLATION gather-only"; g.fn = launch_gather_only; g.checked = false; vars.push_back(g);
    }

    // descriptors per (variant, batch): n_tiles depends on the variant's tile size; split_mid variants have more descriptors
    std::vector<std::vector<DevDesc *>> d_desc(vars.size(), std::vector<DevDesc *>(NB));
    std::vector<uint32_t> tiles(vars.size());
    std::vector<std::vector<uint64_t>> desc_bytes(vars.size());
    for (size_t v = 0; v < vars.size(); v++) {
        tiles[v] = (B + vars[v].bags_per_tile - 1) / vars[v].bags_per_tile;
        for (int b = 0; b < NB; b++) {
            std::vector<DevDesc> hd;
            desc_bytes[v].clear();
            for (uint32_t t = 0; t < T; t++) {
                const uint64_t bytes = kKaggleRows[t] * D * 4;
                const uint32_t pieces = (vars[v].split_mid && bytes > (4ull << 20) && bytes <= (32ull << 20)) ? 8u : 1u;
                const uint64_t rps = (kKaggleRows[t] + pieces - 1) / pieces;
                for (uint32_t k = 0; k < pieces; k++) {
                    const uint64_t lo = k * rps, hi = std::min<uint64_t>(kKaggleRows[t], lo + rps);
                    DevDesc d{};
                    d.weights = reinterpret_cast<const char *>(tables[t]) + lo * D * 4;
                    d.indices = d_idx[b][t];
                    d.offsets = d_off[b][t];
                    d.out = d_out[b][t];
                    d.n_idx = B;
                    d.n_bags = B;
                    d.nr_rows = hi - lo;
                    d.fixed_pooling = 0;
                    d.n_tiles = tiles[v];
                    d.pad_[0] = lo;
                    hd.push_back(d);
                    desc_bytes[v].push_back((hi - lo) * D * 4);
                }
            }
            vars[v].n_desc = (uint32_t)hd.size();
            CK(hipMalloc((void **)&d_desc[v][b], sizeof(DevDesc) * hd.size()));
            CK(hipMemcpy(d_desc[v][b], hd.data(), sizeof(DevDesc) * hd.size(), hipMemcpyHostToDevice));
        }
    }
    for (size_t v = 0; v < vars.size(); v++) {
        if (!vars[v].xcd) continue;
        std::vector<uint32_t> nt(vars[v].n_desc, tiles[v]), words;
        vars[v].xgrid = build_xcd_map(nt, desc_bytes[v], &words, vars[v].xrounds);
        if (vars[v].direct) {
            std::vector<uint32_t> d;
            expand_xcd_map(words, vars[v].xgrid, &d);
            words.swap(d);
        }
        CK(hipMalloc((void **)&vars[v].d_xmap, words.size() * 4));
        CK(hipMemcpy(vars[v].d_xmap, words.data(), words.size() * 4, hipMemcpyHostToDevice));
    }
    CK(hipDeviceSynchronize());

    // correctness: every variant must reproduce variant 0 bit for bit on batch 0
    size_t out_bytes = (size_t)B * D * 4;
    std::vector<std::vector<char>> ref(T, std::vector<char>(out_bytes)), got(T, std::vector<char>(out_bytes));
    for (size_t v = 0; v < vars.size(); v++) {
        if (!vars[v].checked) continue;
        for (uint32_t t = 0; t < T; t++) CK(hipMemsetAsync(d_out[0][t], 0xff, out_bytes, s));
        g_direct_flag = vars[v].direct ? kXmapDirect : 0u;
        vars[v].fn(d_desc[v][0], vars[v].n_desc, tiles[v], vars[v].d_xmap, vars[v].xgrid, s);
        CK(hipGetLastError());
        CK(hipStreamSynchronize(s));
        for (uint32_t t = 0; t < T; t++) {
            CK(hipMemcpy((v == 0 ? ref : got)[t].data(), d_out[0][t], out_bytes, hipMemcpyDeviceToHost));
            if (v && memcmp(ref[t].data(), got[t].data(), out_bytes)) {
                fprintf(stderr, "MISMATCH variant %s table %u\n", vars[v].name.c_str(), t);
                return 2;
            }
        }
    }
    printf("all checked variants bit-identical on batch 0\n");

    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int r = -1; r < rounds; r++) {  // round -1 = warm-up
        for (size_t v = 0; v < vars.size(); v++) {
            g_direct_flag = vars[v].direct ? kXmapDirect : 0u;
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < iters; i++) vars[v].fn(d_desc[v][i % NB], vars[v].n_desc, tiles[v], vars[v].d_xmap, vars[v].xgrid, s);
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (r >= 0) vars[v].us.push_back(ms * 1000.f / iters);
        }
    }
    const double alg = (double)T * B * (D * 4 + 4 + 4 + D * 4);
    printf("%-34s %9s %9s %9s   (B=%u, NB=%d, %d rounds x %d launches)\n", "variant", "min us", "med us",
           "TB/s@med", B, NB, rounds, iters);
    for (auto &v : vars) {
        std::sort(v.us.begin(), v.us.end());
        float med = v.us[v.us.size() / 2];
        printf("%-34s %9.2f %9.2f %9.2f\n", v.name.c_str(), v.us[0], med, alg / (med * 1e-6) / 1e12);
    }
    return 0;
}
