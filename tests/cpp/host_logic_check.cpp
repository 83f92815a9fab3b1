// TEST INFRASTRUCTURE: the HOST side of libpimemb's sources over tests/cpp/hip_runtime_stub.cpp (kernels are no-ops there),
// under ThreadSanitizer / AddressSanitizer + UBSan -- see tests/test_host_side_sanitizers.py.  Through the public C ABI only.
// Checks return codes, tickets and ordering; no lookup result exists here to be looked at.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>

#include <hip/hip_runtime.h>      // (hipGetDevice of the stub: "is the caller's current device untouched?")

#include "pimemb.h"

#define CHECK(x)                                                                                          \
    do {                                                                                                  \
        int rc_ = (x);                                                                                    \
        if (rc_ != EMB_OK) {                                                                              \
            fprintf(stderr, "%s:%d: %s -> %d (%s)\n", __FILE__, __LINE__, #x, rc_, emb_last_error());     \
            exit(1);                                                                                      \
        }                                                                                                 \
    } while (0)
#define EXPECT(c)                                                                       \
    do {                                                                                \
        if (!(c)) {                                                                     \
            fprintf(stderr, "%s:%d: expected %s\n", __FILE__, __LINE__, #c);            \
            exit(1);                                                                    \
        }                                                                               \
    } while (0)

namespace {

constexpr uint32_t kTables = 6, kDim = 16, kRows = 1000;
constexpr uint32_t kMaxTables = 26;

// test controls of tests/cpp/hip_runtime_stub.cpp (several devices, "which device was current at every HIP call")
extern "C" void pimemb_stub_set_device_count(int n);
extern "C" void pimemb_stub_expect_device(int d);
extern "C" long pimemb_stub_violations(void);
extern "C" long pimemb_stub_tracked_calls(void);
extern "C" long pimemb_stub_device_syncs(void);
extern "C" void pimemb_stub_no_finegrained(int on);
extern "C" void pimemb_stub_ipc_open_hangs(int on);

struct Rng {
    uint64_t s;
    uint32_t next() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 11); }
};

emb_engine *make_engine(uint32_t flags, int device = 0, uint32_t n_tables = kTables) {
    emb_config cfg{};
    cfg.device = device;
    cfg.max_tables = n_tables + 4;
    cfg.flags = flags;
    emb_engine *e = nullptr;
    CHECK(emb_create(&cfg, &e));
    std::vector<float> rows((size_t)kRows * kDim, 0.5f);
    for (uint32_t t = 0; t < n_tables; t++) CHECK(emb_load_table(e, t, kRows, kDim, EMB_F32, rows.data(), EMB_MEM_HOST));
    return e;
}

// ---- several caller threads on one engine: transient launches, plans, host-pointer calls, checked calls -------------------
void engine_threads(int device = 0, int n_threads = 4) {
    emb_engine *e = make_engine(0, device);
    std::atomic<int> range_errors{0};
    auto worker = [&](int tid) {
        Rng rng{0x9E3779B97F4A7C15ull * (uint64_t)(tid + 1)};
        void *stream = nullptr;
        CHECK(emb_stream_create(e, &stream));
        const uint32_t B = 64 + 13 * (uint32_t)tid;
        std::vector<std::vector<uint32_t>> idx(kTables), off(kTables);
        std::vector<std::vector<float>> out(kTables);
        std::vector<void *> d_idx(kTables), d_off(kTables), d_out(kTables);
        for (uint32_t t = 0; t < kTables; t++) {
            off[t].resize(B);
            uint32_t at = 0;
            for (uint32_t b = 0; b < B; b++) { off[t][b] = at; at += rng.next() % 4; }
            idx[t].resize(at);
            for (auto &v : idx[t]) v = rng.next() % kRows;
            out[t].assign((size_t)B * kDim, 0.f);
            CHECK(emb_device_alloc(e, at * 4 + 4, &d_idx[t]));
            CHECK(emb_device_alloc(e, B * 4, &d_off[t]));
            CHECK(emb_device_alloc(e, (size_t)B * kDim * 4, &d_out[t]));
            CHECK(emb_copy_to_device(e, d_idx[t], idx[t].data(), at * 4));
            CHECK(emb_copy_to_device(e, d_off[t], off[t].data(), B * 4));
        }
        std::vector<emb_lookup_desc> dev(kTables), host(kTables);
        for (uint32_t t = 0; t < kTables; t++) {
            dev[t] = emb_lookup_desc{t, 0, d_idx[t], d_off[t], idx[t].size(), B, static_cast<float *>(d_out[t])};
            host[t] = emb_lookup_desc{t, 0, idx[t].data(), off[t].data(), idx[t].size(), B, out[t].data()};
        }
        emb_plan *plan = nullptr;
        CHECK(emb_plan_create(e, dev.data(), kTables, EMB_IDX_U32, &plan));
        for (int it = 0; it < 200; it++) {
            CHECK(emb_lookup_batched(e, dev.data(), kTables, EMB_IDX_U32, EMB_MEM_DEVICE, stream));
            CHECK(emb_plan_launch(plan, stream));
            if (it % 4 == 0) CHECK(emb_lookup_batched(e, host.data(), kTables, EMB_IDX_U32, EMB_MEM_HOST, nullptr));
            if (it % 8 == 0) CHECK(emb_lookup(e, (uint32_t)tid % kTables, idx[0].data(), idx[0].size(), off[0].data(), B, out[0].data(),
                                             EMB_IDX_U32, EMB_MEM_HOST, nullptr));
            uint64_t bad = 0;
            if (it % 16 == 5 && tid == 1) {         // ONE thread hands in a bad index now and then: only its call fails
                uint32_t spoiled = kRows + 7;
                CHECK(emb_copy_to_device(e, d_idx[2], &spoiled, 4));
                int rc = emb_lookup_batched_checked(e, dev.data(), kTables, EMB_IDX_U32, EMB_MEM_DEVICE, stream, &bad);
                EXPECT(rc == EMB_ERR_RANGE && bad == 1);
                range_errors++;
                CHECK(emb_copy_to_device(e, d_idx[2], idx[2].data(), 4));
            } else if (idx[2].size()) {
                CHECK(emb_lookup_batched_checked(e, dev.data(), kTables, EMB_IDX_U32, EMB_MEM_DEVICE, stream, &bad));
                EXPECT(bad == 0);
            }
            if (it % 50 == 49) CHECK(emb_synchronize(e, stream));
        }
        CHECK(emb_synchronize(e, stream));
        CHECK(emb_plan_destroy(plan));
        for (uint32_t t = 0; t < kTables; t++) {
            CHECK(emb_device_free(e, d_idx[t]));
            CHECK(emb_device_free(e, d_off[t]));
            CHECK(emb_device_free(e, d_out[t]));
        }
        CHECK(emb_stream_destroy(e, stream));
    };
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; t++) th.emplace_back(worker, t);
    for (auto &t : th) t.join();
    EXPECT(range_errors.load() > 0);
    emb_stats st{};
    CHECK(emb_get_stats(e, &st));
    EXPECT(st.n_kernel_launches > 0 && st.n_lookup_calls > 0);
    CHECK(emb_destroy(e));
    printf("engine threads ok\n");
}

// ---- plan-less launches whose SHAPES do not recur: the XCD-map cache must not wait for the device or allocate per launch --------
// (round 5: the routed sharded step fell from 76 to 129 us per step once its pieces stopped cycling through 8 sizes.)  Big one-index
// launches of 6 tables, a different bag count each time: first sightings carry their map in the launch image; shapes that come
// back are cached on the second sighting; a burst of new recurring shapes evicts idle ones into a graveyard that is freed 64 at a
// time.  Thousands of launches, a handful of device-wide waits -- and AddressSanitizer watches the cache's bookkeeping.
void xmap_cache_shapes() {
    emb_engine *e = make_engine(0);
    void *d_idx = nullptr, *d_out = nullptr;
    CHECK(emb_device_alloc(e, 4096, &d_idx));
    CHECK(emb_device_alloc(e, 4096, &d_out));
    // shape number i: every table its own bag count, spread over 40 000 .. 440 000 bags so that the shapes stay distinct after the
    // engine has rounded their tile counts to 1/16 of a power of two (kernels are no-ops here: only the launch bookkeeping runs,
    // the buffers are never touched)
    auto launch = [&](uint32_t i) {
        emb_lookup_desc d[kTables];
        for (uint32_t t = 0; t < kTables; t++) {
            const uint32_t bags = 40000 + 4096 * ((i * (t + 1) * 7 + 13 * t) % 97);
            d[t] = emb_lookup_desc{t, 1, d_idx, nullptr, bags, bags, static_cast<float *>(d_out)};
        }
        CHECK(emb_lookup_batched(e, d, kTables, EMB_IDX_U32, EMB_MEM_DEVICE, nullptr));
    };
    const long syncs0 = pimemb_stub_device_syncs();
    for (uint32_t i = 0; i < 97; i++) launch(i);                                            // 97 shapes, none seen before: maps in the launch image
    for (uint32_t i = 0; i < 1400; i++) launch(100 + (i * 31) % 97 + 97 * (i / 97));        // (shape i + 97 = shape i) the same 97 in scrambled order:
                                                                                            // second sightings fill the 32 places, the rest stay inline
    EXPECT(pimemb_stub_device_syncs() - syncs0 == 0);
    for (int round = 0; round < 4; round++)
        for (uint32_t i = 0; i < 48; i++) launch(i);                                        // 48 shapes in rotation, 32 entries: no thrash
    EXPECT(pimemb_stub_device_syncs() - syncs0 <= 3);
    for (uint32_t burst = 0; burst < 6; burst++)                                            // bursts of new recurring shapes push idle ones out
        for (int twice = 0; twice < 6; twice++)
            for (uint32_t i = 0; i < 40; i++) launch(48 + (burst * 40 + i) % 49);
    EXPECT(pimemb_stub_device_syncs() - syncs0 == 0);                                       // (evicted maps are freed behind events recorded at their eviction: never a device-wide wait)
    CHECK(emb_synchronize(e, nullptr));
    CHECK(emb_device_free(e, d_idx));
    CHECK(emb_device_free(e, d_out));
    CHECK(emb_destroy(e));
    printf("xmap cache under shapes that do not recur ok (%ld device-wide waits in 3 129 launches)\n", pimemb_stub_device_syncs() - syncs0);
}

// ---- a big host-pointer call: copy-in and lookup in two parts, the second on a stream of the engine's own (lookup_host_split) ---
// Every table returns >= 1.5 MB of rows and the indices are >= 1 MB: the staging offsets of the two parts, the second stream and
// its events, the order of the copies -- under the sanitizers; and a stage-timed call of the same shape (one chain, as before).
void big_host_call() {
    emb_engine *e = make_engine(0);
    const uint32_t T = 5, B = 40000;
    std::vector<std::vector<uint32_t>> idx(T), off(T);
    std::vector<std::vector<float>> out(T);
    std::vector<emb_lookup_desc> d(T);
    Rng rng{0x1234567ull};
    for (uint32_t t = 0; t < T; t++) {
        const uint32_t L = t == 2 ? 0u : 1u;                  // one table with explicit (ragged) offsets, the others fixed pooling
        off[t].resize(B);
        uint32_t at = 0;
        for (uint32_t b = 0; b < B; b++) { off[t][b] = at; at += L ? 1u : rng.next() % 3; }
        idx[t].resize(at);
        for (auto &v : idx[t]) v = rng.next() % kRows;
        out[t].assign((size_t)B * kDim, -1.f);
        d[t] = emb_lookup_desc{t, L, idx[t].data(), L ? nullptr : off[t].data(), idx[t].size(), B, out[t].data()};
    }
    for (int it = 0; it < 3; it++) CHECK(emb_lookup_batched(e, d.data(), T, EMB_IDX_U32, EMB_MEM_HOST, nullptr));
    CHECK(emb_set_stage_timing(e, 1));
    CHECK(emb_lookup_batched(e, d.data(), T, EMB_IDX_U32, EMB_MEM_HOST, nullptr));
    CHECK(emb_set_stage_timing(e, 0));
    void *stream = nullptr;
    CHECK(emb_stream_create(e, &stream));
    CHECK(emb_lookup_batched(e, d.data(), T, EMB_IDX_U32, EMB_MEM_HOST, stream));
    CHECK(emb_lookup_batched(e, d.data(), 2, EMB_IDX_U32, EMB_MEM_HOST, stream));        // two descriptors: one per part
    CHECK(emb_stream_destroy(e, stream));
    emb_stats st{};
    CHECK(emb_get_stats(e, &st));
    EXPECT(st.n_lookup_calls == 6 && st.n_bags == (uint64_t)B * (5 * T + 2));
    CHECK(emb_destroy(e));
    printf("big host-pointer call in two parts ok\n");
}

// ---- checked calls whose verdict is deferred (EMB_FLAG_DEFER_CHECK / emb_lookup_batched_checked_deferred / emb_check_report) -----
// The finding of a call is returned by a LATER call or by emb_check_report, once, naming the call; more calls than verdict
// slots between two reports; a synchronous checked call behind a deferred one reports the earlier finding.
void deferred_check() {
    for (int via_flag = 0; via_flag < 2; via_flag++) {
        emb_engine *e = make_engine(via_flag ? (EMB_FLAG_CHECK_INPUTS | EMB_FLAG_DEFER_CHECK) : 0u);
        const uint32_t B = 96;
        std::vector<uint32_t> idx(B), off(B);
        for (uint32_t b = 0; b < B; b++) { idx[b] = (b * 37u) % kRows; off[b] = b; }
        void *d_idx = nullptr, *d_off = nullptr, *d_out = nullptr, *stream = nullptr;
        CHECK(emb_stream_create(e, &stream));
        CHECK(emb_device_alloc(e, B * 4, &d_idx));
        CHECK(emb_device_alloc(e, B * 4, &d_off));
        CHECK(emb_device_alloc(e, (size_t)B * kDim * 4, &d_out));
        CHECK(emb_copy_to_device(e, d_idx, idx.data(), B * 4));
        CHECK(emb_copy_to_device(e, d_off, off.data(), B * 4));
        emb_lookup_desc d{1, 0, d_idx, d_off, B, B, static_cast<float *>(d_out)};
        auto call = [&]() {
            return via_flag ? emb_lookup_batched(e, &d, 1, EMB_IDX_U32, EMB_MEM_DEVICE, stream)
                            : emb_lookup_batched_checked_deferred(e, &d, 1, EMB_IDX_U32, EMB_MEM_DEVICE, stream);
        };
        auto spoil = [&](bool on) {
            uint32_t v = on ? kRows + 3 : idx[5];
            CHECK(emb_copy_to_device(e, static_cast<char *>(d_idx) + 20, &v, 4));
        };
        uint64_t bad = 7;
        for (int i = 0; i < 150; i++) CHECK(call());                 // more clean calls than verdict slots: every one is read in turn
        CHECK(emb_check_report(e, &bad));
        EXPECT(bad == 0);
        emb_stats before{}, after{};
        CHECK(emb_get_stats(e, &before));
        spoil(true);
        CHECK(call());                                               // the bad call itself: its verdict is not waited for
        spoil(false);
        int rc = call();                                             // the next one brings the finding (and has launched its own work)
        EXPECT(rc == EMB_ERR_RANGE && strstr(emb_last_error(), "EARLIER") != nullptr && strstr(emb_last_error(), "number 151") != nullptr);
        CHECK(call());                                               // ... once
        CHECK(emb_check_report(e, &bad));
        EXPECT(bad == 0);
        CHECK(emb_get_stats(e, &after));
        EXPECT(after.n_lookup_calls == before.n_lookup_calls + 3);
        EXPECT(after.n_kernel_launches == before.n_kernel_launches + 2);       // the refused call's lookup did nothing
        // a finding nobody met in a later call: emb_check_report has it, once, with the count
        spoil(true);
        CHECK(call());
        spoil(false);
        rc = emb_check_report(e, &bad);
        EXPECT(rc == EMB_ERR_RANGE && bad == 1);
        CHECK(emb_check_report(e, &bad));
        EXPECT(bad == 0);
        // a synchronous checked call behind a deferred bad one: the earlier finding comes out of it; its own verdict is clean
        spoil(true);
        CHECK(call());
        spoil(false);
        rc = emb_lookup_batched_checked(e, &d, 1, EMB_IDX_U32, EMB_MEM_DEVICE, stream, &bad);
        EXPECT(rc == EMB_ERR_RANGE && bad == 1 && strstr(emb_last_error(), "EARLIER") != nullptr);
        CHECK(emb_lookup_batched_checked(e, &d, 1, EMB_IDX_U32, EMB_MEM_DEVICE, stream, &bad));
        EXPECT(bad == 0);
        // emb_validate_inputs is never deferred
        spoil(true);
        rc = emb_validate_inputs_on(e, &d, 1, EMB_IDX_U32, EMB_MEM_DEVICE, stream, &bad);
        EXPECT(rc == EMB_ERR_RANGE && bad == 1);
        spoil(false);
        CHECK(emb_synchronize(e, stream));
        CHECK(emb_device_free(e, d_idx));
        CHECK(emb_device_free(e, d_off));
        CHECK(emb_device_free(e, d_out));
        CHECK(emb_stream_destroy(e, stream));
        CHECK(emb_destroy(e));
    }
    printf("deferred verdicts of checked calls ok\n");
}

// ---- the engine picks a table's hot rows itself (emb_learn_hot_rows): sampling, counting, the min_share rule -------------------
void learn_hot_rows_check() {
    emb_engine *e = make_engine(0);
    Rng rng{99};
    // a skewed batch over table 1 (device copy) and the same as int64 from host memory: ids 7, 3, 11 dominate
    std::vector<uint32_t> ix(300000);
    for (size_t i = 0; i < ix.size(); i++) {
        const uint32_t r = rng.next() % 100;
        ix[i] = r < 40 ? 7u : (r < 65 ? 3u : (r < 80 ? 11u : rng.next() % kRows));
    }
    ix[5] = kRows + 9;                                   // an id outside the table is nobody's hot row
    void *d_ix = nullptr;
    CHECK(emb_device_alloc(e, ix.size() * 4, &d_ix));
    CHECK(emb_copy_to_device(e, d_ix, ix.data(), ix.size() * 4));
    uint32_t n = 0;
    float share = 0.f;
    CHECK(emb_learn_hot_rows(e, 1, d_ix, ix.size(), EMB_IDX_U32, EMB_MEM_DEVICE, 3, 0.05f, nullptr, &n, &share));
    EXPECT(n == 3 && share > 0.75f && share < 0.85f);    // four runs of 65 536 of the 300 000 ids were counted
    std::vector<int64_t> wide(ix.begin(), ix.begin() + 5000);
    wide[1] = -4;
    CHECK(emb_learn_hot_rows(e, 1, wide.data(), wide.size(), EMB_IDX_I64, EMB_MEM_HOST, 2, 0.05f, nullptr, &n, &share));
    EXPECT(n == 2 && share > 0.55f && share < 0.75f);
    // near-uniform accesses: the three most frequent ids cover far less than 5 % -- the set is cleared
    for (size_t i = 0; i < ix.size(); i++) ix[i] = rng.next() % kRows;
    CHECK(emb_copy_to_device(e, d_ix, ix.data(), ix.size() * 4));
    CHECK(emb_learn_hot_rows(e, 1, d_ix, ix.size(), EMB_IDX_U32, EMB_MEM_DEVICE, 3, 0.05f, nullptr, &n, &share));
    EXPECT(n == 0 && share < 0.05f);
    CHECK(emb_learn_hot_rows(e, 1, nullptr, 0, EMB_IDX_U32, EMB_MEM_DEVICE, 3, 0.05f, nullptr, &n, &share));      // an empty batch: cleared, no error
    EXPECT(n == 0);
    EXPECT(emb_learn_hot_rows(e, 63, d_ix, 4, EMB_IDX_U32, EMB_MEM_DEVICE, 3, 0.f, nullptr, nullptr, nullptr) == EMB_ERR_INVALID);
    CHECK(emb_device_free(e, d_ix));
    CHECK(emb_destroy(e));
    printf("engine-side hot rows ok\n");
}

// ---- the request queue: adders, a free-running flusher, waiters that collect late ------------------------------------------
void queue_threads(int device = 0) {
    emb_engine *e = make_engine(0, device);
    emb_queue *q = nullptr;
    CHECK(emb_queue_create(e, EMB_IDX_U32, EMB_MEM_HOST, &q));
    std::atomic<bool> stop{false};
    std::atomic<uint64_t> flushed{0};
    std::thread flusher([&] {
        void *stream = nullptr;
        CHECK(emb_stream_create(e, &stream));
        while (!stop.load()) {
            uint32_t n = 0;
            CHECK(emb_queue_flush(q, stream, &n));
            flushed += n;
            std::this_thread::yield();
        }
        uint32_t n = 0;
        CHECK(emb_queue_flush(q, stream, &n));
        flushed += n;
        CHECK(emb_synchronize(e, stream));
        CHECK(emb_stream_destroy(e, stream));
    });
    constexpr int kClients = 3, kRequests = 400;
    auto client = [&](int tid) {
        Rng rng{77ull + (uint64_t)tid};
        for (int r = 0; r < kRequests; r++) {
            const uint32_t B = 1 + rng.next() % 8, n_t = 1 + rng.next() % kTables;
            std::vector<std::vector<uint32_t>> idx(n_t), off(n_t);
            std::vector<std::vector<float>> out(n_t);
            std::vector<emb_lookup_desc> d(n_t);
            for (uint32_t t = 0; t < n_t; t++) {
                off[t].resize(B);
                uint32_t at = 0;
                for (uint32_t b = 0; b < B; b++) { off[t][b] = at; at += rng.next() % 3; }
                idx[t].resize(at + 1);
                for (auto &v : idx[t]) v = rng.next() % kRows;
                out[t].assign((size_t)B * kDim, -1.f);
                d[t] = emb_lookup_desc{t, 0, idx[t].data(), off[t].data(), at, B, out[t].data()};
            }
            uint64_t ticket = 0;
            CHECK(emb_queue_add(q, d.data(), n_t, &ticket));
            if (r % 7 == 3) std::this_thread::yield();      // collect late now and then: generations must be held for us
            int rc;         // (a wait ahead of the flush that carries the request is refused, not blocked: the client tries again)
            while ((rc = emb_queue_wait(q, ticket)) == EMB_ERR_INVALID && strstr(emb_last_error(), "not been flushed")) std::this_thread::yield();
            CHECK(rc);
        }
    };
    std::vector<std::thread> th;
    for (int t = 0; t < kClients; t++) th.emplace_back(client, t);
    for (auto &t : th) t.join();
    stop = true;
    flusher.join();
    EXPECT(flushed.load() == (uint64_t)kClients * kRequests);
    uint64_t never = 0;
    EXPECT(emb_queue_wait(q, (uint64_t)kClients * kRequests + 5) == EMB_ERR_INVALID);
    (void)never;
    CHECK(emb_queue_destroy(q));

    // a device queue from two threads + add_many
    CHECK(emb_queue_create(e, EMB_IDX_U32, EMB_MEM_DEVICE, &q));
    void *d_idx = nullptr, *d_out = nullptr;
    CHECK(emb_device_alloc(e, 4096, &d_idx));
    CHECK(emb_device_alloc(e, 64 * kDim * 4, &d_out));
    auto adder = [&](int) {
        for (int r = 0; r < 300; r++) {
            emb_lookup_desc d[2] = {{0, 1, d_idx, nullptr, 8, 8, static_cast<float *>(d_out)}, {1, 2, d_idx, nullptr, 16, 8, static_cast<float *>(d_out)}};
            uint32_t n[2] = {1, 1};
            uint64_t first = 0;
            CHECK(emb_queue_add_many(q, d, n, 2, &first));
            if (r % 16 == 15) {
                uint32_t got = 0;
                CHECK(emb_queue_flush(q, nullptr, &got));
            }
        }
    };
    std::thread a(adder, 0), b(adder, 1);
    a.join();
    b.join();
    uint32_t got = 0;
    CHECK(emb_queue_flush(q, nullptr, &got));
    CHECK(emb_queue_destroy(q));
    CHECK(emb_device_free(e, d_idx));
    CHECK(emb_device_free(e, d_out));
    CHECK(emb_destroy(e));
    printf("queue threads ok\n");
}

// ---- the sharded call -------------------------------------------------------------------------------------------------------
// Placement mixes: SMALL = the six tables of the earlier rounds (2 replicated, 2 whole, 2 row-split); C4 = the mix the planner
// gives BASELINE configs[3] at 8 ranks (26 tables: 20 replicated, 6 split by row range); C5 = configs[4]'s (every table
// whole on an owner rank, two per rank at world 8).
enum Mix { MIX_SMALL = 0, MIX_C4 = 1, MIX_C5 = 2 };
uint32_t tables_of(Mix m) { return m == MIX_SMALL ? kTables : (m == MIX_C4 ? 26u : 16u); }
void fill_tables(Mix m, int world, emb_shard_table *tabs) {
    const uint32_t n = tables_of(m);
    for (uint32_t t = 0; t < n; t++) {
        tabs[t].owner = (int32_t)(t % (uint32_t)world);
        tabs[t].engine_table = t;
        tabs[t].rows_per_shard = kRows;
        if (m == MIX_SMALL) tabs[t].placement = t < 2 ? EMB_PLACE_REPLICATED : (t < 4 ? EMB_PLACE_WHOLE : EMB_PLACE_ROWS);
        else if (m == MIX_C4) tabs[t].placement = (t == 0 || t == 9 || t == 10 || t == 19 || t == 20 || t == 21) ? EMB_PLACE_ROWS : EMB_PLACE_REPLICATED;
        else tabs[t].placement = EMB_PLACE_WHOLE;
    }
}

// One rank's life with a shard object: `steps` batches of its OWN (ragged / one-index / empty) bags through submit / wait /
// flush or the synchronous call, for every depth in `depths`.  comm: the RCCL stand-in (or NULL); peer: the peer group (or
// NULL).  checked: EMB_SHARD_CHECK_SERVED -- and, where a one-index batch takes the direct path (one rank, or peer stores),
// rank 0 hands in ONE index no shard holds at step `bad_step`: exactly that rank gets EMB_ERR_RANGE, nobody hangs, the
// batches after it are clean.
struct Drive {
    emb_engine *e; int rank, world; Mix mix; emb_comm *comm; emb_peer *peer; bool checked; std::vector<uint32_t> depths; int steps;
    bool idx64 = false;      // int64 index / offset arrays (torch's width), used in place
    bool defer = false;      // EMB_SHARD_DEFER_REPORT: a requester-side finding surfaces at the next call / flush, never inside the call
};
// index / offset arrays of either width from the same numbers (a negative `bad` id only exists as int64)
struct IdxBuf {
    std::vector<uint32_t> u;
    std::vector<int64_t> l;
    bool wide;
    explicit IdxBuf(bool w) : wide(w) {}
    void push(uint64_t v) { if (wide) l.push_back((int64_t)v); else u.push_back((uint32_t)v); }
    size_t size() const { return wide ? l.size() : u.size(); }
    void set(size_t i, int64_t v) { if (wide) l[i] = v; else u[i] = (uint32_t)v; }
    const void *data() const { return wide ? (const void *)l.data() : (const void *)u.data(); }
    size_t bytes() const { return size() * (wide ? 8 : 4); }
};
void drive_shard(const Drive &D) {
    emb_engine *e = D.e;
    const uint32_t T = tables_of(D.mix);
    emb_shard_table tabs[kMaxTables];
    fill_tables(D.mix, D.world, tabs);
    const bool direct_possible = D.comm == nullptr;      // (no peer behind RCCL: one rank, or peer stores)
    auto dev_alloc = [&](size_t bytes) {
        void *p = nullptr;
        if (D.peer) CHECK(emb_peer_alloc(D.peer, bytes ? bytes : 4, &p));
        else CHECK(emb_device_alloc(e, bytes ? bytes : 4, &p));
        return p;
    };
    const uint32_t Bmax = 64;
    constexpr int kSlots = 8;
    std::vector<std::vector<void *>> idx(kSlots), off(kSlots), out(kSlots);
    for (int k = 0; k < kSlots; k++)
        for (uint32_t t = 0; t < T; t++) {
            idx[k].push_back(dev_alloc(Bmax * 4 * 8));
            off[k].push_back(dev_alloc(Bmax * 8));
            out[k].push_back(dev_alloc((size_t)Bmax * kDim * 4));
        }
    for (uint32_t depth : D.depths) {
        emb_shard_config cfg{};
        cfg.n_tables = T;
        cfg.dim = kDim;
        cfg.depth = depth;
        cfg.flags = (D.peer ? EMB_SHARD_PEER_STORES : 0u) | (D.checked ? EMB_SHARD_CHECK_SERVED : 0u) | (D.defer ? EMB_SHARD_DEFER_REPORT : 0u);
        cfg.tables = tabs;
        cfg.peer = D.peer;
        emb_shard *s = nullptr;
        CHECK(emb_shard_create(e, D.comm, &cfg, &s));
        CHECK(emb_shard_set_kernel_timing(s, depth == 2));
        Rng rng{1000ull * (uint64_t)(D.rank + 1) + depth + 17ull * (uint64_t)D.mix};
        std::vector<uint64_t> seqs;
        const int bad_step = (D.checked && direct_possible) ? 9 : -1;        // (9 % 4 == 1: a one-index batch)
        int range_seen = 0;
        auto note = [&](int rc) {           // a deferred EMB_ERR_RANGE may surface at the submit that un-routes the batch, or at the flush
            if (rc == EMB_ERR_RANGE) range_seen++;
            else CHECK(rc);
        };
        for (int j = 0; j < D.steps; j++) {
            const int sl = j % kSlots;
            const uint32_t B = (j == 5 && D.rank == 1) ? 0u : 1 + rng.next() % Bmax;       // every rank has its OWN bags; one is empty once
            const bool one_hot = j % 2 == 1;
            emb_shard_input in[kMaxTables];
            for (uint32_t t = 0; t < T; t++) {
                IdxBuf o(D.idx64), ix(D.idx64);
                const uint32_t range = tabs[t].placement == EMB_PLACE_ROWS ? kRows * (uint32_t)D.world : kRows;
                for (uint32_t b = 0; b < B; b++) {
                    o.push(ix.size());
                    const uint32_t len = one_hot ? 1u : rng.next() % 4;
                    for (uint32_t k = 0; k < len; k++) ix.push(rng.next() % range);
                }
                // a row nobody holds (last table: row-split / whole); int64: a NEGATIVE id, or one beyond 2^32, by depth
                if (j == bad_step && D.rank == 0 && t == T - 1 && B)
                    ix.set(B / 2, !D.idx64 ? (int64_t)range + 5 : (depth % 2 ? -3 : (int64_t)(1ull << 32) + 7));
                if (ix.size()) CHECK(emb_copy_to_device(e, idx[sl][t], ix.data(), ix.bytes()));
                if (B) CHECK(emb_copy_to_device(e, off[sl][t], o.data(), o.bytes()));
                in[t] = emb_shard_input{idx[sl][t], one_hot ? nullptr : off[sl][t], ix.size(), one_hot ? 1u : 0u,
                                        D.idx64 ? (uint32_t)EMB_IDX_I64 : (uint32_t)EMB_IDX_U32, static_cast<float *>(out[sl][t])};
            }
            if (depth == 0 || j % 7 == 6) {
                note(emb_shard_lookup(s, in, B, nullptr));
                seqs.clear();
            } else {
                uint64_t seq = 0;
                note(emb_shard_submit(s, in, B, nullptr, &seq));
                seqs.push_back(seq);
                if (seqs.size() > depth) {
                    CHECK(emb_shard_wait(s, seqs.front(), nullptr));
                    seqs.erase(seqs.begin());
                }
            }
            if (j == bad_step && D.defer) EXPECT(range_seen == 0);       // (deferred: the call that completed the batch said nothing)
            if (j == bad_step) {            // every rank alike: flush is collective
                note(emb_shard_flush(s));
                for (uint64_t q : seqs) CHECK(emb_shard_wait(s, q, nullptr));
                seqs.clear();
                if (D.rank == 0) {
                    EXPECT(range_seen == 1);
                    EXPECT(strstr(emb_last_error(), "bags were served") != nullptr);
                } else {
                    EXPECT(range_seen == 0);
                }
            }
        }
        CHECK(emb_shard_flush(s));
        for (uint64_t q : seqs) CHECK(emb_shard_wait(s, q, nullptr));
        EXPECT(range_seen == (bad_step >= 0 && bad_step < D.steps && D.rank == 0 ? 1 : 0));
        emb_shard_stats st{};
        CHECK(emb_shard_get_stats(s, &st, 0));
        EXPECT(st.n_batches == (uint64_t)D.steps);
        if (D.world > 1) EXPECT(st.bytes_to_peers > 0);
        if (D.world > 1 && D.mix != MIX_C5 && D.comm) EXPECT(st.served_sub_bags > 0 && st.served_indices >= st.served_sub_bags);   // row pieces really travelled (the stub counts like the router)
        uint32_t counts[2 * 64 * 8];
        (void)emb_shard_sent_counts(s, 0, counts, 2 * 64 * 8);
        if (D.peer) CHECK(emb_peer_barrier(D.peer));         // nobody tears its shard down while a peer may still be gathering from this arena
        CHECK(emb_shard_destroy(s));
    }
    if (!D.peer)
        for (int k = 0; k < kSlots; k++)
            for (uint32_t t = 0; t < T; t++) {
                CHECK(emb_device_free(e, idx[k][t]));
                CHECK(emb_device_free(e, off[k][t]));
                CHECK(emb_device_free(e, out[k][t]));
            }
}

// ---- one rank: every placement, every depth, ragged and one-index batches, peer-store mode, checked ---------------------------
void shard_one_rank(bool peer_mode, bool checked, emb_engine *shared = nullptr, int device = 0, bool idx64 = false, bool defer = false) {
    emb_engine *e = shared ? shared : make_engine(0, device);
    emb_peer *peer = nullptr;
    if (peer_mode) {
        char tag[64];
        snprintf(tag, sizeof tag, "hostcheck-%d-%d", (int)getpid(), (int)checked);
        CHECK(emb_peer_create(e, tag, 0, 1, 64ull << 20, &peer));
    }
    Drive D{e, 0, 1, MIX_SMALL, nullptr, peer, checked, {0, 1, 2, 3}, 20};
    D.idx64 = idx64;
    D.defer = defer;
    drive_shard(D);
    if (peer) {
        CHECK(emb_peer_barrier(peer));
        CHECK(emb_peer_destroy(peer));
    }
    if (!shared) {
        CHECK(emb_destroy(e));
        printf("shard one rank%s%s%s%s ok\n", peer_mode ? " (peer stores)" : "", checked ? " (checked)" : "", idx64 ? " (int64 ids)" : "", defer ? " (deferred report)" : "");
    }
}

// A batch's finding is returned ONCE (round 5 returned the serving side's twice: by the call that ran S(b) and again by the one
// that ran U(b) -- two IndexErrors in Python for one bad batch, the second on a clean submit).  One rank, checked, every depth:
// a RAGGED batch (validated up front, not counted) hands in one index beyond a replicated table at step 8.  The launch that
// finds it gathers nothing: the batch whose replicated tables rode in it (8) and the batch whose pieces it served (8 at depth 0;
// 7 at depth 1; 6 at depth 2 / 3) hold zero rows -- one EMB_ERR_RANGE each, none for any other submit, and the batches after it
// are clean.
void report_once(bool idx64) {
    emb_engine *e = make_engine(0);
    emb_shard_table tabs[kMaxTables];
    fill_tables(MIX_SMALL, 1, tabs);
    const uint32_t T = kTables, B = 32;
    std::vector<void *> idx(T), off(T), out(T);
    for (uint32_t t = 0; t < T; t++) {
        CHECK(emb_device_alloc(e, B * 2 * 8, &idx[t]));
        CHECK(emb_device_alloc(e, B * 8, &off[t]));
        CHECK(emb_device_alloc(e, (size_t)B * kDim * 4, &out[t]));
    }
    for (uint32_t depth = 0; depth < 4; depth++) {
        emb_shard_config cfg{};
        cfg.n_tables = T;
        cfg.dim = kDim;
        cfg.depth = depth;
        cfg.flags = EMB_SHARD_CHECK_SERVED;
        cfg.tables = tabs;
        emb_shard *s = nullptr;
        CHECK(emb_shard_create(e, nullptr, &cfg, &s));
        int range_seen = 0, first = -1, last = -1;
        for (int j = 0; j < 16; j++) {
            emb_shard_input in[kMaxTables];
            for (uint32_t t = 0; t < T; t++) {
                IdxBuf o(idx64), ix(idx64);
                for (uint32_t b = 0; b < B; b++) {
                    o.push(ix.size());
                    ix.push((b * 7 + j) % kRows);
                    ix.push((b * 13 + t) % kRows);
                }
                if (j == 8 && t == 0) ix.set(5, (int64_t)kRows + 1);
                CHECK(emb_copy_to_device(e, idx[t], ix.data(), ix.bytes()));
                CHECK(emb_copy_to_device(e, off[t], o.data(), o.bytes()));
                in[t] = emb_shard_input{idx[t], off[t], ix.size(), 0u, idx64 ? (uint32_t)EMB_IDX_I64 : (uint32_t)EMB_IDX_U32, static_cast<float *>(out[t])};
            }
            uint64_t seq = 0;
            const int rc = emb_shard_submit(s, in, B, nullptr, &seq);
            if (rc == EMB_ERR_RANGE) {
                range_seen++;
                if (first < 0) first = j;
                last = j;
            } else CHECK(rc);
            CHECK(emb_synchronize(e, nullptr));        // (one set of buffers: the batch must be through before they are rewritten)
        }
        const int rc = emb_shard_flush(s);
        if (rc == EMB_ERR_RANGE) range_seen++;
        else CHECK(rc);
        EXPECT(range_seen == (depth == 0 ? 1 : 2));
        EXPECT(first >= 8 && last <= 8 + 3);            // nothing before the bad batch, nothing once it and its companion are through
        CHECK(emb_shard_destroy(s));
    }
    for (uint32_t t = 0; t < T; t++) {
        CHECK(emb_device_free(e, idx[t]));
        CHECK(emb_device_free(e, off[t]));
        CHECK(emb_device_free(e, out[t]));
    }
    CHECK(emb_destroy(e));
    printf("one report per bad batch%s ok\n", idx64 ? " (int64 ids)" : "");
}

// several shard objects, each with its own caller thread, on ONE engine (pimemb.h: "several shard objects may share an engine")
void shards_sharing_an_engine() {
    emb_engine *e = make_engine(0);
    std::vector<std::thread> th;
    for (int t = 0; t < 3; t++) th.emplace_back([e, t] { shard_one_rank(false, t == 1, e); });
    for (auto &t : th) t.join();
    CHECK(emb_destroy(e));
    printf("three shard objects on one engine ok\n");
}

// ---- several ranks as THREADS over a stand-in for emb_comm (pimemb_comm.cpp, the RCCL binding, is not linked here) ----------
// Sends are copied on the spot, receives wait for the matching send of the same pair (pieces of a pair match in order, as
// grouped ncclSend / ncclRecv do) and insist on the same byte count: both sides' size arithmetic must agree -- for the counts
// messages, the whole-table transfers and, since the stub's routers count like the real ones, ragged row pieces both ways.
struct Hub {
    std::mutex mu;
    std::condition_variable cv;
    std::vector<std::deque<std::vector<char>>> q;     // [src * world + dst]
    int world;
    explicit Hub(int w) : q((size_t)w * w), world(w) {}
};
}  // namespace
struct emb_comm { int rank; Hub *hub; };
extern "C" int emb_comm_rank(const emb_comm *c, int32_t *rank, int32_t *world) {
    *rank = c->rank;
    *world = c->hub->world;
    return EMB_OK;
}
extern "C" int emb_comm_exchange(emb_comm *c, const emb_comm_op *ops, uint32_t n_ops, void *) {
    Hub &h = *c->hub;
    {
        std::lock_guard<std::mutex> lk(h.mu);
        for (uint32_t i = 0; i < n_ops; i++)
            if (!ops[i].is_recv) {
                const char *p = static_cast<const char *>(ops[i].ptr);
                h.q[(size_t)c->rank * h.world + ops[i].peer].emplace_back(p, p + ops[i].bytes);
            }
    }
    h.cv.notify_all();
    for (uint32_t i = 0; i < n_ops; i++)
        if (ops[i].is_recv) {
            std::unique_lock<std::mutex> lk(h.mu);
            auto &q = h.q[(size_t)ops[i].peer * h.world + c->rank];
            if (!h.cv.wait_for(lk, std::chrono::seconds(120), [&] { return !q.empty(); })) {
                fprintf(stderr, "rank %d: nothing arrived from rank %d\n", c->rank, ops[i].peer);
                exit(1);
            }
            if (q.front().size() != ops[i].bytes) {
                fprintf(stderr, "rank %d expects %llu bytes from rank %d, which sent %zu\n", c->rank, (unsigned long long)ops[i].bytes, ops[i].peer, q.front().size());
                exit(1);
            }
            memcpy(ops[i].ptr, q.front().data(), ops[i].bytes);
            q.pop_front();
        }
    return EMB_OK;
}
namespace {

void shard_ranks_as_threads(int world, Mix mix, bool checked, std::vector<uint32_t> depths, bool idx64 = false) {
    Hub hub(world);
    std::atomic<int> finished{0};
    auto rank_main = [&](int rank) {
        emb_engine *e = make_engine(0, 0, tables_of(mix));
        emb_comm comm{rank, &hub};
        Drive D{e, rank, world, mix, &comm, nullptr, checked, depths, 16};
        D.idx64 = idx64;
        drive_shard(D);
        CHECK(emb_destroy(e));
        finished++;
    };
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, r);
    for (auto &t : th) t.join();
    EXPECT(finished.load() == world);
    for (auto &q : hub.q) EXPECT(q.empty());           // every piece sent was received
    printf("shard %d ranks as threads, mix %d%s%s ok\n", world, (int)mix, checked ? ", checked" : "", idx64 ? ", int64 ids" : "");
}

// ---- the collective-free exchange with several ranks as threads: one peer group, an IPC handle is the pointer itself --------
// (AddressSanitizer build only: the hosts poll mailbox words that a peer's KERNEL writes -- here another thread's stub -- with
// plain volatile loads, which ThreadSanitizer rightly calls a race between threads and which is none between a GPU and a host)
#if defined(__has_feature)
#if __has_feature(thread_sanitizer)
#define HOST_CHECK_TSAN 1
#endif
#endif
void peer_ranks_as_threads(int world, Mix mix, bool checked, std::vector<uint32_t> depths, bool idx64 = false, bool defer = false) {
#ifdef HOST_CHECK_TSAN
    (void)mix; (void)checked; (void)depths; (void)idx64; (void)defer;
    printf("peer stores, %d ranks as threads: skipped under ThreadSanitizer\n", world);
#else
    char tag[64];
    snprintf(tag, sizeof tag, "hostcheck-peers-%d-%d-%d-%d-%d%d", (int)getpid(), world, (int)mix, (int)checked, (int)idx64, (int)defer);
    std::atomic<int> finished{0};
    auto rank_main = [&](int rank) {
        emb_engine *e = make_engine(0, 0, tables_of(mix));
        emb_peer *peer = nullptr;
        CHECK(emb_peer_create(e, tag, rank, world, 96ull << 20, &peer));
        Drive D{e, rank, world, mix, nullptr, peer, checked, depths, 24};
        D.idx64 = idx64;
        D.defer = defer;
        drive_shard(D);
        CHECK(emb_peer_barrier(peer));
        CHECK(emb_peer_destroy(peer));
        CHECK(emb_destroy(e));
        finished++;
    };
    std::vector<std::thread> th;
    for (int r = 0; r < world; r++) th.emplace_back(rank_main, r);
    for (auto &t : th) t.join();
    EXPECT(finished.load() == world);
    printf("peer stores, %d ranks as threads, mix %d%s%s%s ok\n", world, (int)mix, checked ? ", checked" : "", idx64 ? ", int64 ids" : "", defer ? ", deferred report" : "");
#endif
}

// ---- a node with several GPUs: everything on device 1 while the calling threads' current device stays 0 ------------------------
// The stub counts every HIP call made with another device current (or on a stream / event of another device).  Engine (four
// caller threads), request queues (host + device, with the flusher and client threads), the sharded call with one rank
// (routed, direct, checked, peer stores) and a two-rank peer group: no violation, and the callers' current device is untouched.
void everything_on_device_one() {
    pimemb_stub_set_device_count(2);
    int cur = -1;
    EXPECT(hipGetDevice(&cur) == hipSuccess && cur == 0);
    pimemb_stub_expect_device(1);
    const long before = pimemb_stub_tracked_calls();
    engine_threads(1, 2);
    queue_threads(1);
    shard_one_rank(false, false, nullptr, 1);
    shard_one_rank(false, true, nullptr, 1);
    shard_one_rank(true, true, nullptr, 1);
#ifndef HOST_CHECK_TSAN
    {       // two peer-store ranks as threads, both engines on device 1 (the stub has one address space)
        char tag[64];
        snprintf(tag, sizeof tag, "hostcheck-dev1-%d", (int)getpid());
        auto rank_main = [&](int rank) {
            emb_engine *e = make_engine(0, 1, tables_of(MIX_SMALL));
            emb_peer *peer = nullptr;
            CHECK(emb_peer_create(e, tag, rank, 2, 96ull << 20, &peer));
            drive_shard(Drive{e, rank, 2, MIX_SMALL, nullptr, peer, true, {0, 3}, 16});
            CHECK(emb_peer_barrier(peer));
            CHECK(emb_peer_destroy(peer));
            CHECK(emb_destroy(e));
            int c = -1;
            EXPECT(hipGetDevice(&c) == hipSuccess && c == 0);      // a rank thread's own current device was never changed for good
        };
        std::thread a(rank_main, 0), b(rank_main, 1);
        a.join();
        b.join();
    }
#endif
    {       // ranks behind the RCCL stand-in
        Hub hub(2);
        auto rank_main = [&](int rank) {
            emb_engine *e = make_engine(0, 1, tables_of(MIX_SMALL));
            emb_comm comm{rank, &hub};
            drive_shard(Drive{e, rank, 2, MIX_SMALL, &comm, nullptr, false, {0, 3}, 16});
            CHECK(emb_destroy(e));
        };
        std::thread a(rank_main, 0), b(rank_main, 1);
        a.join();
        b.join();
    }
    EXPECT(hipGetDevice(&cur) == hipSuccess && cur == 0);
    const long calls = pimemb_stub_tracked_calls() - before, bad = pimemb_stub_violations();
    pimemb_stub_expect_device(-1);
    pimemb_stub_set_device_count(1);
    if (bad) {
        fprintf(stderr, "%ld HIP calls were made with the wrong device current\n", bad);
        exit(1);
    }
    EXPECT(calls > 10000);
    printf("everything on device 1: %ld HIP calls, every one with device 1 current ok\n", calls);
}

// ---- the peer group's two exposures -------------------------------------------------------------------
// (a) a runtime without fine-grained device memory: a group of one rank falls back to ordinary memory (nobody else stores into
//     it), a group with peers refuses -- unless ordinary memory was asked for by name (PIMEMB_PEER_ARENA=coarse).
void peer_group_without_finegrained_memory() {
    pimemb_stub_no_finegrained(1);
    emb_engine *e = make_engine(0);
    emb_peer *peer = nullptr;
    char tag[64];
    snprintf(tag, sizeof tag, "hostcheck-nofg-%d", (int)getpid());
    CHECK(emb_peer_create(e, tag, 0, 1, 8ull << 20, &peer));
    int32_t fg = 1;
    CHECK(emb_peer_info(peer, nullptr, nullptr, nullptr, nullptr, nullptr, &fg));
    EXPECT(fg == 0);
    CHECK(emb_peer_destroy(peer));
    std::atomic<int> refused{0}, taken{0};
    auto join2 = [&](const char *suffix, bool expect_ok) {
        char t2[96];
        snprintf(t2, sizeof t2, "%s-%s", tag, suffix);
        auto rank_main = [&, expect_ok](int rank) {
            emb_engine *e2 = make_engine(0);
            emb_peer *p2 = nullptr;
            const int rc = emb_peer_create(e2, t2, rank, 2, 8ull << 20, &p2);
            if (expect_ok) {
                CHECK(rc);
                int32_t f = 1;
                CHECK(emb_peer_info(p2, nullptr, nullptr, nullptr, nullptr, nullptr, &f));
                EXPECT(f == 0);
                taken++;
                CHECK(emb_peer_barrier(p2));
                CHECK(emb_peer_destroy(p2));
            } else {
                EXPECT(rc == EMB_ERR_UNSUPPORTED && p2 == nullptr && strstr(emb_last_error(), "PIMEMB_PEER_ARENA=coarse") != nullptr);
                refused++;
            }
            CHECK(emb_destroy(e2));
        };
        std::thread a(rank_main, 0), b(rank_main, 1);
        a.join();
        b.join();
    };
    join2("refuse", false);
    EXPECT(refused.load() == 2);
    setenv("PIMEMB_PEER_ARENA", "coarse", 1);
    join2("coarse", true);
    unsetenv("PIMEMB_PEER_ARENA");
    EXPECT(taken.load() == 2);
    CHECK(emb_destroy(e));
    pimemb_stub_no_finegrained(0);
    printf("peer group without fine-grained memory ok\n");
}

// (b) hipIpcOpenMemHandle that never returns: the watchdog ends the PROCESS with a non-zero status and a message naming rank,
//     peer and chunk (run as `host_logic_check ipc-hang` by the test, which expects exactly that)
[[noreturn]] void peer_group_ipc_mapping_hangs() {
    setenv("PIMEMB_SHARD_TIMEOUT_S", "1", 1);
    pimemb_stub_ipc_open_hangs(1);
    char tag[64];
    snprintf(tag, sizeof tag, "hostcheck-hang-%d", (int)getpid());
    auto rank_main = [&](int rank) {
        emb_engine *e = make_engine(0);
        emb_peer *peer = nullptr;
        (void)emb_peer_create(e, tag, rank, 2, 8ull << 20, &peer);
        fprintf(stderr, "emb_peer_create returned although the mapping call hangs\n");
    };
    std::thread a(rank_main, 0), b(rank_main, 1);
    a.join();
    b.join();
    exit(0);        // (not reached: the watchdog ends the process first)
}

// ---- the reference's two entry points -------------------------------------------------------------------------------------
void compat_calls() {
    const uint32_t nt = 3, nc = 8, nb = 16, per = 4;
    CHECK(emb_configure(nt, nc, nb, per));
    std::vector<int32_t> col(500, 1000000);
    struct dpu_set_t *set = nullptr;
    for (uint32_t t = 0; t < nt; t++)
        for (uint32_t c = 0; c < nc; c++) {
            set = populate_mram(t, 500, c, col.data(), nullptr);
            EXPECT(set != nullptr);
        }
    std::vector<std::vector<uint32_t>> idx(nt, std::vector<uint32_t>(nb * per, 3)), off(nt, std::vector<uint32_t>(nb));
    std::vector<std::vector<float>> out(nt, std::vector<float>(nb * nc));
    std::vector<uint32_t *> pi(nt), po(nt);
    std::vector<float *> pr(nt);
    for (uint32_t t = 0; t < nt; t++) {
        for (uint32_t b = 0; b < nb; b++) off[t][b] = b * per;
        pi[t] = idx[t].data(); po[t] = off[t].data(); pr[t] = out[t].data();
    }
    for (int it = 0; it < 5; it++) EXPECT(lookup(pi.data(), po.data(), pr.data(), set, 0) == nullptr);      // (an error would be printed: the test greps for it)
    CHECK(emb_compat_reset());
    printf("compat ok\n");
}

}  // namespace

int main(int argc, char **argv) {
    if (argc > 1 && strcmp(argv[1], "ipc-hang") == 0) peer_group_ipc_mapping_hangs();
    const bool only_world8 = argc > 1 && strcmp(argv[1], "world8") == 0;
    if (!only_world8) {
        engine_threads();
        xmap_cache_shapes();
        learn_hot_rows_check();
        deferred_check();
        big_host_call();
        queue_threads();
        shard_one_rank(false, false);
        shard_one_rank(false, true);
        shard_one_rank(true, false);
        shard_one_rank(true, true);
        shards_sharing_an_engine();
        // int64 ids in place (the router, the ranged launches, whole tables' arrays at 8 bytes per id), the deferred report
        shard_one_rank(false, true, nullptr, 0, /*idx64=*/true);
        shard_one_rank(false, true, nullptr, 0, /*idx64=*/false, /*defer=*/true);
        shard_one_rank(true, true, nullptr, 0, /*idx64=*/true, /*defer=*/true);
        report_once(false);
        report_once(true);
        shard_ranks_as_threads(2, MIX_SMALL, true, {0, 3}, /*idx64=*/true);
        shard_ranks_as_threads(3, MIX_SMALL, false, {1, 2}, /*idx64=*/true);
        peer_ranks_as_threads(3, MIX_SMALL, true, {0, 2, 3}, /*idx64=*/true, /*defer=*/true);
        shard_ranks_as_threads(2, MIX_SMALL, false, {0, 1, 2, 3});
        shard_ranks_as_threads(3, MIX_SMALL, true, {0, 1, 2, 3});
        peer_ranks_as_threads(2, MIX_SMALL, true, {0, 1, 2, 3});
        peer_ranks_as_threads(3, MIX_SMALL, false, {0, 1, 2, 3});
        everything_on_device_one();
        peer_group_without_finegrained_memory();
        compat_calls();
    }
    // world 8 -- the sizes that change there (counts[N][K+1][2], mailboxes [dst][src][slot], N * K descriptors, slots (d << 24))
    // walked with the C4 placement mix (20 replicated / 6 row-split), the C5 mix (every table whole on an owner) and the small one
    shard_ranks_as_threads(8, MIX_C4, false, {0, 3});
    shard_ranks_as_threads(8, MIX_C4, true, {2});
    shard_ranks_as_threads(8, MIX_C5, false, {0, 3});
    shard_ranks_as_threads(8, MIX_SMALL, false, {1, 3});
    peer_ranks_as_threads(8, MIX_C4, false, {0, 3});
    peer_ranks_as_threads(8, MIX_C4, true, {0, 3});
    peer_ranks_as_threads(8, MIX_C5, true, {0, 2});
    peer_ranks_as_threads(8, MIX_SMALL, true, {1, 3});
    shard_ranks_as_threads(8, MIX_C4, true, {3}, /*idx64=*/true);
    peer_ranks_as_threads(8, MIX_C4, true, {0, 3}, /*idx64=*/true, /*defer=*/true);
    printf("host logic ok\n");
    return 0;
}
