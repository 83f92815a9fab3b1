// emb_threads_bench.cpp -- plan-less device-pointer lookups from K serving threads, each on a stream of its own: do the
// threads scale, or do they queue up behind one lock?  (Round 3's launch path held the engine-wide mutex across image copy +
// enqueue; round 4 draws launch images from a pool of rings picked by the calling thread.)  Every thread owns its buffers;
// results are checked at the end (one index per bag: the pooled row IS the table row).  Plain C++ over the C ABI.
//
//   emb_threads_bench [tables dim rows bags calls_per_thread]        (defaults: 26 16 100000 2048 4000)
#include <time.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "pimemb.h"

static double now_us() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3;
}
#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != EMB_OK) {                                                         \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, emb_last_error());         \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

int main(int argc, char **argv) {
    const uint32_t T = argc > 1 ? (uint32_t)atoi(argv[1]) : 26, D = argc > 2 ? (uint32_t)atoi(argv[2]) : 16;
    const uint32_t ROWS = argc > 3 ? (uint32_t)atoi(argv[3]) : 100000, B = argc > 4 ? (uint32_t)atoi(argv[4]) : 2048;
    const uint32_t CALLS = argc > 5 ? (uint32_t)atoi(argv[5]) : 4000;
    emb_engine *e = nullptr;
    emb_config cfg = {0, T, 0};
    CHECK(emb_create(&cfg, &e));
    std::vector<std::vector<float>> tab(T, std::vector<float>((size_t)ROWS * D));
    for (uint32_t t = 0; t < T; t++) {
        for (size_t i = 0; i < tab[t].size(); i++) tab[t][i] = (float)((i * 2654435761u + t * 40503u) % 1000003u) * 1e-3f;
        CHECK(emb_load_table(e, t, ROWS, D, EMB_F32, tab[t].data(), EMB_MEM_HOST));
    }
    struct Client {
        void *stream = nullptr, *d_idx = nullptr, *d_off = nullptr, *d_out = nullptr;
        std::vector<uint32_t> idx;
        std::vector<emb_lookup_desc> d;
    };
    const uint32_t KMAX = 8;
    std::vector<Client> cl(KMAX);
    std::vector<uint32_t> off(B);
    for (uint32_t b = 0; b < B; b++) off[b] = b;
    uint64_t x = 88172645463325252ull;
    for (Client &c : cl) {
        CHECK(emb_stream_create(e, &c.stream));
        c.idx.resize((size_t)T * B);
        for (uint32_t &v : c.idx) {
            x ^= x << 13; x ^= x >> 7; x ^= x << 17;
            v = (uint32_t)(x % ROWS);
        }
        CHECK(emb_device_alloc(e, c.idx.size() * 4, &c.d_idx));
        CHECK(emb_device_alloc(e, off.size() * 4, &c.d_off));
        CHECK(emb_device_alloc(e, (size_t)T * B * D * 4, &c.d_out));
        CHECK(emb_copy_to_device(e, c.d_idx, c.idx.data(), c.idx.size() * 4));
        CHECK(emb_copy_to_device(e, c.d_off, off.data(), off.size() * 4));
        c.d.assign(T, emb_lookup_desc{});
        for (uint32_t t = 0; t < T; t++) {
            c.d[t].table_id = t;
            c.d[t].indices = (const uint32_t *)c.d_idx + (size_t)t * B;
            c.d[t].offsets = c.d_off;
            c.d[t].n_indices = B;
            c.d[t].n_bags = B;
            c.d[t].pooled = (float *)c.d_out + (size_t)t * B * D;
        }
    }
    printf("%u tables x dim %u, %u bags per table, plan-less device-pointer calls, one stream per thread:\n", T, D, B);
    double base = 0;
    for (uint32_t K : {1u, 2u, 4u, 8u}) {
        for (uint32_t k = 0; k < K; k++) CHECK(emb_memset_device(e, cl[k].d_out, 0xff, (size_t)T * B * D * 4));
        const double t0 = now_us();
        std::vector<std::thread> th;
        for (uint32_t k = 0; k < K; k++)
            th.emplace_back([&, k]() {
                for (uint32_t i = 0; i < CALLS; i++) CHECK(emb_lookup_batched(e, cl[k].d.data(), T, EMB_IDX_U32, EMB_MEM_DEVICE, cl[k].stream));
                CHECK(emb_synchronize(e, cl[k].stream));
            });
        for (auto &t : th) t.join();
        const double us = now_us() - t0, rate = (double)K * CALLS / us * 1e6;
        if (K == 1) base = rate;
        uint64_t bad = 0;
        std::vector<float> out((size_t)T * B * D);
        for (uint32_t k = 0; k < K; k++) {
            CHECK(emb_copy_to_host(e, out.data(), cl[k].d_out, out.size() * 4));
            for (uint32_t t = 0; t < T; t++)
                for (uint32_t b = 0; b < B; b += 97)
                    bad += memcmp(&out[((size_t)t * B + b) * D], &tab[t][(size_t)cl[k].idx[(size_t)t * B + b] * D], D * 4) != 0;
        }
        printf("  %u thread(s): %.2f us per call per thread, %.3e calls/s in all (%.2fx one thread), %.3e pooled lookups/s, mismatches %llu\n",
               K, us / CALLS, rate, rate / base, rate * T * B, (unsigned long long)bad);
        if (bad) return 2;
    }
    for (Client &c : cl) {
        CHECK(emb_device_free(e, c.d_idx));
        CHECK(emb_device_free(e, c.d_off));
        CHECK(emb_device_free(e, c.d_out));
        CHECK(emb_stream_destroy(e, c.stream));
    }
    CHECK(emb_destroy(e));
    return 0;
}
