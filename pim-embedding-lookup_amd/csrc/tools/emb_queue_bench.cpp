// emb_queue_bench.cpp -- the reference's SERVING shapes on its own calling convention (host pointers), through the request
// queue: C client threads each post small requests (all T tables, B bags per table, one index per bag: mini-batch 1 is
// README.md:6 of the reference, 32 upmem/run.sh:119) and wait for their rows; a front-end thread flushes whatever is
// pending as ONE launch.  Against the same requests issued one by one through the plain host-pointer call
// (emb_lookup_batched, EMB_MEM_HOST -- what lookup() does per call, emb_host.h:234).  Every request's rows are checked
// against the table (one index per bag: the pooled row IS the table row).  Plain C++ over the C ABI, no HIP headers.
//
//   emb_queue_bench [tables dim rows bags clients requests_per_client window]   (defaults: 26 16 100000 1 8 2000 1)
//   window: requests a client keeps outstanding before it waits for them (1 = a closed loop: post one, wait, post the next)
#include <time.h>

#include <atomic>
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
    const uint32_t ROWS = argc > 3 ? (uint32_t)atoi(argv[3]) : 100000, B = argc > 4 ? (uint32_t)atoi(argv[4]) : 1;
    const uint32_t CLIENTS = argc > 5 ? (uint32_t)atoi(argv[5]) : 8, PER = argc > 6 ? (uint32_t)atoi(argv[6]) : 2000;
    const uint32_t WINDOW = argc > 7 && atoi(argv[7]) > 0 ? (uint32_t)atoi(argv[7]) : 1;
    emb_engine *e = nullptr;
    emb_config cfg = {0, T, 0};
    CHECK(emb_create(&cfg, &e));
    std::vector<std::vector<float>> tab(T, std::vector<float>((size_t)ROWS * D));
    for (uint32_t t = 0; t < T; t++) {
        for (size_t i = 0; i < tab[t].size(); i++) tab[t][i] = (float)((i * 2654435761u + t * 40503u) % 1000003u) * 1e-3f;
        CHECK(emb_load_table(e, t, ROWS, D, EMB_F32, tab[t].data(), EMB_MEM_HOST));
    }
    // every client's requests: indices / offsets / output rows of its own
    struct Req {
        std::vector<uint32_t> idx, off;
        std::vector<float> out;
    };
    std::vector<std::vector<Req>> reqs(CLIENTS, std::vector<Req>(PER));
    uint64_t x = 88172645463325252ull;
    for (auto &cl : reqs)
        for (Req &r : cl) {
            r.idx.resize((size_t)T * B);
            r.off.resize(B);
            r.out.assign((size_t)T * B * D, -1.0f);
            for (uint32_t b = 0; b < B; b++) r.off[b] = b;
            for (uint32_t &v : r.idx) {
                x ^= x << 13; x ^= x >> 7; x ^= x << 17;
                v = (uint32_t)(x % ROWS);
            }
        }
    auto descs_of = [&](Req &r, std::vector<emb_lookup_desc> &d) {
        d.assign(T, emb_lookup_desc{});
        for (uint32_t t = 0; t < T; t++) {
            d[t].table_id = t;
            d[t].indices = r.idx.data() + (size_t)t * B;
            d[t].offsets = r.off.data();
            d[t].n_indices = B;
            d[t].n_bags = B;
            d[t].pooled = r.out.data() + (size_t)t * B * D;
        }
    };
    auto check_all = [&]() {
        uint64_t bad = 0;
        for (auto &cl : reqs)
            for (Req &r : cl)
                for (uint32_t t = 0; t < T; t++)
                    for (uint32_t b = 0; b < B; b++)
                        bad += memcmp(&r.out[((size_t)t * B + b) * D], &tab[t][(size_t)r.idx[(size_t)t * B + b] * D], D * 4) != 0;
        return bad;
    };
    const uint64_t total = (uint64_t)CLIENTS * PER;

    // ---- one by one: every client thread makes the plain host-pointer call per request (calls are serialised inside the engine)
    double t0 = now_us();
    {
        std::vector<std::thread> th;
        for (uint32_t c = 0; c < CLIENTS; c++)
            th.emplace_back([&, c]() {
                std::vector<emb_lookup_desc> d;
                for (Req &r : reqs[c]) {
                    descs_of(r, d);
                    CHECK(emb_lookup_batched(e, d.data(), T, EMB_IDX_U32, EMB_MEM_HOST, nullptr));
                }
            });
        for (auto &t : th) t.join();
    }
    const double us_one = (now_us() - t0) / (double)total;
    const uint64_t bad_one = check_all();
    for (auto &cl : reqs)
        for (Req &r : cl) std::fill(r.out.begin(), r.out.end(), -1.0f);

    // ---- queued: clients add and wait; a front-end thread flushes whatever is pending (one launch per flush)
    emb_queue *q = nullptr;
    CHECK(emb_queue_create(e, EMB_IDX_U32, EMB_MEM_HOST, &q));
    std::atomic<uint64_t> done{0};
    std::atomic<bool> stop{false};
    uint64_t flushes = 0, flushed = 0;
    t0 = now_us();
    std::thread front([&]() {
        while (!stop.load(std::memory_order_acquire)) {
            uint32_t n = 0;
            CHECK(emb_queue_flush(q, nullptr, &n));
            if (n) {
                flushes++;
                flushed += n;
            }
        }
    });
    {
        std::vector<std::thread> th;
        for (uint32_t c = 0; c < CLIENTS; c++)
            th.emplace_back([&, c]() {
                std::vector<emb_lookup_desc> d;
                std::vector<uint64_t> tickets;
                for (size_t i0 = 0; i0 < reqs[c].size(); i0 += WINDOW) {
                    tickets.clear();
                    for (size_t i = i0; i < reqs[c].size() && i < i0 + WINDOW; i++) {      // post a window of requests ...
                        descs_of(reqs[c][i], d);
                        uint64_t ticket = 0;
                        CHECK(emb_queue_add(q, d.data(), T, &ticket));
                        tickets.push_back(ticket);
                    }
                    for (uint64_t ticket : tickets) {                                        // ... then collect them
                        int rc;
                        while ((rc = emb_queue_wait(q, ticket)) == EMB_ERR_INVALID && strstr(emb_last_error(), "not been flushed")) {
                        }                                     // (the front end has not picked it up yet)
                        if (rc != EMB_OK) {
                            fprintf(stderr, "emb_queue_wait -> %d: %s\n", rc, emb_last_error());
                            exit(1);
                        }
                        done.fetch_add(1);
                    }
                }
            });
        for (auto &t : th) t.join();
    }
    const double us_q = (now_us() - t0) / (double)total;
    stop.store(true, std::memory_order_release);
    front.join();
    const uint64_t bad_q = check_all();
    CHECK(emb_queue_destroy(q));
    CHECK(emb_destroy(e));
    printf("%u tables x dim %u, %u bags per table per request, %u client threads x %u requests, %u outstanding per client (host pointers):\n"
           "  one by one (emb_lookup_batched per request): %.2f us / request = %.3e pooled lookups/s, mismatches %llu\n"
           "  request queue (add + wait; front end flushes): %.2f us / request = %.3e pooled lookups/s, %.1f requests per launch, mismatches %llu\n"
           "  speed-up %.1fx\n",
           T, D, B, CLIENTS, PER, WINDOW, us_one, (double)T * B / us_one * 1e6, (unsigned long long)bad_one, us_q, (double)T * B / us_q * 1e6,
           flushes ? (double)flushed / (double)flushes : 0.0, (unsigned long long)bad_q, us_one / us_q);
    return (bad_one || bad_q) ? 2 : 0;
}
