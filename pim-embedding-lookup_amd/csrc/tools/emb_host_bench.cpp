// emb_host_bench.cpp -- synthetic driver over the reference-compatible entry points of libpimemb.so.
//
// Counterpart of the reference's upmem/src/load_generator.c (the APP_MAIN of both `emb_host` and
// `emblib.so`, upmem/Makefile:36,117-124): same three roles, same generator shapes --
//   synthetic_populate   (:27-38)  random int32 tables pushed through populate_mram, column by column
//   synthetic_inference  (:67-122) uniform indices, offsets[i] = i*L, 100 timed lookup() calls
//   validate_result      (:40-65)  CPU re-computation, |out*1e9 - sum| <= 1000            (enabled here)
// with the reference's slips not reproduced: every column is uploaded (it uploads col 0 only, :34),
// wall time is CLOCK_MONOTONIC over whole calls (it sums tv_nsec of CPU time, :98-103), the last bag
// is validated (it reads offsets[j+1] past the end, :50).
//
//   emb_host_bench [nr_tables nr_cols nr_rows nr_batches indices_per_batch iters validate_tables]
//   validate_tables: host copies kept for the CPU check (default: all; the big points of the table-size
//   sweep keep a few, 32 tables x 13.9M rows x 64 columns are 114 GB)
//   (defaults: $NR_TABLES $NR_COLS 50000 $MAX_NR_BATCHES $MAX_INDICES_PER_BATCH 100, else the toy
//    preset of upmem/run.sh:93-101 with load_generator.c:125-127: 9 64 50000 64 32 100)
#include <time.h>

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pimemb.h"

static double now_ms() {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

static uint32_t env_or(const char *name, uint32_t dflt) {
    const char *v = getenv(name);
    return v ? (uint32_t)strtoul(v, nullptr, 10) : dflt;
}

// checker only (never produces results): load_generator.c:40-65 with the indexing slips repaired
static uint64_t validate_result(const std::vector<std::vector<int32_t>> &emb_tables, uint32_t nr_cols,
                                const std::vector<std::vector<uint32_t>> &indices,
                                const std::vector<std::vector<uint32_t>> &offsets, uint32_t indices_len,
                                uint32_t nr_batches, const std::vector<std::vector<float>> &results) {
    uint64_t bad = 0;
    for (size_t i = 0; i < emb_tables.size(); i++)
        for (uint32_t j = 0; j < nr_batches; j++) {
            const uint32_t p0 = offsets[i][j], e = (j + 1 < nr_batches) ? offsets[i][j + 1] : indices_len;
            for (uint32_t t = 0; t < nr_cols; t++) {
                uint32_t acc = 0;
                for (uint32_t p = p0; p < e; p++) acc += (uint32_t)emb_tables[i][(uint64_t)indices[i][p] * nr_cols + t];
                if (std::fabs((double)results[i][(uint64_t)j * nr_cols + t] * 1e9 - (double)(int32_t)acc) > 1000.0) bad++;
            }
        }
    return bad;
}

int main(int argc, char **argv) {
    uint32_t nr_tables = env_or("NR_TABLES", 9), nr_cols = env_or("NR_COLS", 64), nr_rows = 50000;
    uint32_t nr_batches = env_or("MAX_NR_BATCHES", 64), per_batch = env_or("MAX_INDICES_PER_BATCH", 32);
    uint32_t iters = 100;
    if (argc > 1) nr_tables = (uint32_t)atoi(argv[1]);
    if (argc > 2) nr_cols = (uint32_t)atoi(argv[2]);
    if (argc > 3) nr_rows = (uint32_t)atoi(argv[3]);
    if (argc > 4) nr_batches = (uint32_t)atoi(argv[4]);
    if (argc > 5) per_batch = (uint32_t)atoi(argv[5]);
    if (argc > 6) iters = (uint32_t)atoi(argv[6]);
    uint32_t keep = nr_tables;
    if (argc > 7) keep = std::min<uint32_t>(nr_tables, (uint32_t)atoi(argv[7]));
    const uint32_t indices_len = nr_batches * per_batch;
    printf("emb_host_bench: %u tables x %u cols x %u rows, %u bags x %u indices, %u lookups (%s)\n", nr_tables,
           nr_cols, nr_rows, nr_batches, per_batch, iters, emb_version());
    if (emb_configure(nr_tables, nr_cols, nr_batches, per_batch) != EMB_OK) {
        fprintf(stderr, "emb_configure: %s\n", emb_last_error());
        return 1;
    }

    // synthetic_populate: rand() int32 tables; small magnitudes so sums stay far from the wrap point
    // is NOT assumed -- full-range values, wrap-around is part of the arithmetic being checked.
    srand(1);
    uint64_t xs = 88172645463325252ull;   // table contents: xorshift64 (rand() would take minutes at sweep sizes)
    std::vector<std::vector<int32_t>> emb_tables(nr_tables);
    struct dpu_set_t *handle = nullptr;
    dpu_runtime_totals rt = {};
    double t0 = now_ms();
    std::vector<int32_t> colmajor((size_t)nr_rows * nr_cols);   // the table split into columns (alloc_buffers' job)
    for (uint32_t k = 0; k < nr_tables; k++) {
        emb_tables[k].resize((size_t)nr_rows * nr_cols);
        for (auto &v : emb_tables[k]) {
            xs ^= xs << 13; xs ^= xs >> 7; xs ^= xs << 17;
            v = (int32_t)(xs >> 16);
        }
        for (uint32_t r0 = 0; r0 < nr_rows; r0 += 512) {         // blocked transpose: rows stay in cache
            const uint32_t r1 = std::min(nr_rows, r0 + 512);
            for (uint32_t c = 0; c < nr_cols; c++)
                for (uint32_t r = r0; r < r1; r++) colmajor[(size_t)c * nr_rows + r] = emb_tables[k][(size_t)r * nr_cols + c];
        }
        for (uint32_t c = 0; c < nr_cols; c++) {
            handle = populate_mram(k, nr_rows, c, colmajor.data() + (size_t)c * nr_rows, &rt);
            if (!handle) {
                fprintf(stderr, "populate_mram: %s\n", emb_last_error());
                return 1;
            }
        }
        if (k >= keep) std::vector<int32_t>().swap(emb_tables[k]);   // uploaded; not needed for the check
    }
    printf("populate: %.1f ms total (%u populate_mram calls, copy-in %.1f ms)\n", now_ms() - t0,
           nr_tables * nr_cols, rt.execution_time_populate_copy_in);

    // synthetic_inference
    std::vector<std::vector<uint32_t>> indices(nr_tables), offsets(nr_tables);
    std::vector<std::vector<float>> results(nr_tables);
    std::vector<uint32_t *> pi(nr_tables), po(nr_tables);
    std::vector<float *> pr(nr_tables);
    for (uint32_t k = 0; k < nr_tables; k++) {
        indices[k].resize(indices_len);
        offsets[k].resize(nr_batches);
        results[k].assign((size_t)nr_batches * nr_cols, NAN);
        for (uint32_t i = 0; i < nr_batches; i++) {
            offsets[k][i] = i * per_batch;
            for (uint32_t j = 0; j < per_batch; j++) {
                uint32_t v = (uint32_t)((double)rand() / RAND_MAX * nr_rows);
                indices[k][i * per_batch + j] = v < nr_rows ? v : nr_rows - 1;
            }
        }
        pi[k] = indices[k].data();
        po[k] = offsets[k].data();
        pr[k] = results[k].data();
    }
    lookup(pi.data(), po.data(), pr.data(), handle, 1);  // warm-up, prints the six stage latencies
    std::vector<double> lat(iters);
    emb_engine *e = emb_compat_engine();
    emb_reset_stats(e);
    for (uint32_t i = 0; i < iters; i++) {
        double a = now_ms();
        lookup(pi.data(), po.data(), pr.data(), handle, 0);
        lat[i] = now_ms() - a;
    }
    std::sort(lat.begin(), lat.end());
    double mean = 0;
    for (double v : lat) mean += v;
    mean /= iters;
    emb_stats st;
    emb_get_stats(e, &st);
    printf("lookup(): mean %.3f ms, median %.3f ms, min %.3f ms, max %.3f ms over %u calls\n", mean,
           lat[iters / 2], lat[0], lat[iters - 1], iters);
    printf("  -> %.3e pooled (table,bag) outputs/s, %.3e row gathers/s (host pointers, PCIe inclusive)\n",
           nr_tables * (double)nr_batches / (mean * 1e-3), nr_tables * (double)indices_len / (mean * 1e-3));
    // stage breakdown: a second, stage-timed loop (the host waits after every stage, so calls are slower)
    emb_reset_stats(e);
    emb_set_stage_timing(e, 1);
    for (uint32_t i = 0; i < iters; i++) lookup(pi.data(), po.data(), pr.data(), handle, 0);
    emb_set_stage_timing(e, 0);
    emb_get_stats(e, &st);
    printf("stage means, stage-timed calls (us): copy-in indices+offsets %.1f | descriptors+enqueue %.1f | kernel %.1f | "
           "copy-out enqueue %.1f | post-process %.1f | final wait %.1f\n",
           st.us_copy_in_indices / iters, st.us_copy_in_lengths / iters, st.us_launch / iters,
           st.us_copy_out / iters, st.us_post_process / iters, st.us_sync / iters);

    emb_tables.resize(keep);
    uint64_t bad = validate_result(emb_tables, nr_cols, indices, offsets, indices_len, nr_batches, results);
    printf("Validation result: %s (%llu cells beyond 1000 fixed-point units = 1e-6; %u of %u tables checked)\n",
           bad ? "false" : "true", (unsigned long long)bad, keep, nr_tables);
    emb_compat_reset();
    return bad ? 2 : 0;
}
