// pimemb_compat.cpp -- the two reference entry points, same names and signatures, on top of the
// native engine, so that a caller of the reference's emblib.so (upmem/Makefile:122-124) relinks
// against libpimemb.so unchanged:
//
//   struct dpu_set_t* populate_mram(uint32_t table_id, uint64_t nr_rows, uint32_t col,
//                                   int32_t* table_data, dpu_runtime_totals* runtime)   emb_host.h:136
//   int32_t* lookup(uint32_t** indices, uint32_t** offsets, float** final_results,
//                   void* dpu_set_ptr_untyped, int64_t latency_print)                   emb_host.h:234
//
// Differences that are deliberate:
//   * NR_TABLES / NR_COLS / MAX_NR_BATCHES / MAX_INDICES_PER_BATCH are runtime values
//     (emb_configure or same-named environment variables) instead of -D macros;
//   * tables live row-major int32 [nr_rows][NR_COLS] in HBM; a populate_mram call scatters its
//     column into that layout (one DPU per column does not exist here);
//   * arithmetic is the reference's (int32 wrap-around sum, (float)acc/1e9) minus its bugs: no
//     dropped last bag for odd nr_batches (emb_dpu_lookup.c:118-123), every table is converted
//     (post_process only converts tables < number of ranks, emb_host.h:207), nothing is leaked
//     (emb_host.h:305-318,386-393);
//   * errors do not exit() the process: populate_mram returns NULL, lookup leaves final_results
//     untouched, and emb_last_error() holds the text.
// Kept as the reference has it: the BAG-0 START RULE -- the first bag of every table starts at index 0
// whatever offsets[t][0] says (tasklet 0: `indices_ptr[me()]=0`, emb_dpu_lookup.c:60-63; the reference's own
// validator walks from ind_ptr = 0 too, load_generator.c:46).  The native emb_lookup* API follows
// nn.EmbeddingBag instead (bag 0 starts at offsets[0], which torch requires to be 0): lookup() hands the
// engine a copy of a table's offsets with entry 0 zeroed when a caller passes anything else.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

#include "pimemb_internal.h"

namespace {

struct CompatConfig {
    uint32_t nr_tables = 0, nr_cols = 0, max_nr_batches = 0, max_indices_per_batch = 0;
    bool set = false;
};

std::mutex g_mu;
CompatConfig g_cfg;
emb_engine *g_engine = nullptr;  // the reference's global `dpu_set` (emb_host.h:33)
std::vector<uint64_t> g_rows;    // rows allocated per table (0 = not yet populated)

uint32_t env_u32(const char *name) {
    const char *v = getenv(name);
    return v ? (uint32_t)strtoul(v, nullptr, 10) : 0u;
}

double now_us() {
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

// The reference aborts on error (DPU_ASSERT, emb_host.h:157); here the text goes to stderr once
// and stays readable through emb_last_error().
int compat_fail(const char *msg) {
    fprintf(stderr, "pimemb: %s\n", msg);
    return pimemb::fail(EMB_ERR_INVALID, "%s", msg);
}

bool config_ready_locked() {
    if (g_cfg.set) return true;
    // run.sh:40-45 / r.sh:6-10 export exactly these names for the reference's Makefile
    CompatConfig c;
    c.nr_tables = env_u32("NR_TABLES");
    c.nr_cols = env_u32("NR_COLS");
    c.max_nr_batches = env_u32("MAX_NR_BATCHES");
    c.max_indices_per_batch = env_u32("MAX_INDICES_PER_BATCH");
    if (!c.nr_tables || !c.nr_cols || !c.max_nr_batches || !c.max_indices_per_batch) return false;
    c.set = true;
    g_cfg = c;
    return true;
}

void drop_engine_locked() {
    if (g_engine) emb_destroy(g_engine);
    g_engine = nullptr;
    g_rows.clear();
}

}  // namespace

extern "C" {

int emb_configure(uint32_t nr_tables, uint32_t nr_cols, uint32_t max_nr_batches,
                  uint32_t max_indices_per_batch) {
    if (!nr_tables || !nr_cols || !max_nr_batches || !max_indices_per_batch)
        return compat_fail("emb_configure: all four shape values must be non-zero");
    std::lock_guard<std::mutex> lk(g_mu);
    drop_engine_locked();
    g_cfg.nr_tables = nr_tables;
    g_cfg.nr_cols = nr_cols;
    g_cfg.max_nr_batches = max_nr_batches;
    g_cfg.max_indices_per_batch = max_indices_per_batch;
    g_cfg.set = true;
    return EMB_OK;
}

emb_engine *emb_compat_engine(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    return g_engine;
}

int emb_compat_reset(void) {
    std::lock_guard<std::mutex> lk(g_mu);
    drop_engine_locked();
    g_cfg = CompatConfig{};
    return EMB_OK;
}

struct dpu_set_t *populate_mram(uint32_t table_id, uint64_t nr_rows, uint32_t col,
                                int32_t *table_data, dpu_runtime_totals *runtime) {
    std::lock_guard<std::mutex> lk(g_mu);
    if (!config_ready_locked()) {
        compat_fail("populate_mram: call emb_configure() or export NR_TABLES, NR_COLS, "
                    "MAX_NR_BATCHES, MAX_INDICES_PER_BATCH first");
        return nullptr;
    }
    double t0 = now_us();
    if (!g_engine) {  // first_run, emb_host.h:155-160
        emb_config cfg{};
        cfg.device = -1;
        cfg.max_tables = g_cfg.nr_tables;
        const char *chk = getenv("PIMEMB_CHECK_INPUTS");   // the reference interface has no flags argument
        if (chk && chk[0] && chk[0] != '0') cfg.flags |= EMB_FLAG_CHECK_INPUTS;
        if (emb_create(&cfg, &g_engine) != EMB_OK) {
            fprintf(stderr, "pimemb: populate_mram: %s\n", emb_last_error());
            g_engine = nullptr;
            return nullptr;
        }
        g_rows.assign(g_cfg.nr_tables, 0);
    }
    if (table_id >= g_cfg.nr_tables || col >= g_cfg.nr_cols || !table_data || nr_rows == 0) {
        compat_fail("populate_mram: table_id/col out of range, NULL data or zero rows");
        return nullptr;
    }
    if (g_rows[table_id] != nr_rows) {  // first column of this table (or a resize): allocate, zeroed
        if (emb_alloc_table(g_engine, table_id, nr_rows, g_cfg.nr_cols, EMB_FIXED32) != EMB_OK) {
            fprintf(stderr, "pimemb: populate_mram: %s\n", emb_last_error());
            return nullptr;
        }
        g_rows[table_id] = nr_rows;
    }
    if (emb_load_table_column(g_engine, table_id, col, table_data, nr_rows) != EMB_OK) {
        fprintf(stderr, "pimemb: populate_mram: %s\n", emb_last_error());
        return nullptr;
    }
    if (runtime) runtime->execution_time_populate_copy_in += (now_us() - t0) / 1000.0;
    return reinterpret_cast<struct dpu_set_t *>(g_engine);  // emb_host.h:182
}

int32_t *lookup(uint32_t **indices, uint32_t **offsets, float **final_results,
                void *dpu_set_ptr_untyped, int64_t latency_print) {
    std::lock_guard<std::mutex> lk(g_mu);
    emb_engine *e = reinterpret_cast<emb_engine *>(dpu_set_ptr_untyped);
    if (!e || e != g_engine || !g_cfg.set || !indices || !offsets || !final_results) {
        compat_fail("lookup: bad handle or NULL argument");
        return nullptr;
    }
    // emb_host.h:30,282-283: lengths are always the configured maxima
    const uint64_t indices_len = (uint64_t)g_cfg.max_indices_per_batch * g_cfg.max_nr_batches;
    std::vector<emb_lookup_desc> descs(g_cfg.nr_tables);
    std::vector<std::vector<uint32_t>> patched;       // offsets of tables whose caller passed offsets[t][0] != 0
    for (uint32_t t = 0; t < g_cfg.nr_tables; t++) {
        descs[t] = emb_lookup_desc{};
        descs[t].table_id = t;
        descs[t].indices = indices[t];
        descs[t].offsets = offsets[t];
        if (offsets[t] && g_cfg.max_nr_batches && offsets[t][0] != 0) {     // emb_dpu_lookup.c:60-63: bag 0 starts at index 0
            patched.emplace_back(offsets[t], offsets[t] + g_cfg.max_nr_batches);
            patched.back()[0] = 0;
        }
        descs[t].n_indices = indices_len;
        descs[t].n_bags = g_cfg.max_nr_batches;
        descs[t].pooled = final_results[t];
    }
    for (uint32_t t = 0, k = 0; t < g_cfg.nr_tables && k < patched.size(); t++)       // (the vectors no longer move)
        if (offsets[t] && g_cfg.max_nr_batches && offsets[t][0] != 0) descs[t].offsets = patched[k++].data();
    emb_stats before{}, after{};
    emb_get_stats(e, &before);
    if (latency_print == 1) emb_set_stage_timing(e, 1);   // the reference's per-stage TIME_NOW brackets
    int rc = emb_lookup_batched(e, descs.data(), g_cfg.nr_tables, EMB_IDX_U32, EMB_MEM_HOST, nullptr);
    if (latency_print == 1) emb_set_stage_timing(e, 0);
    if (rc != EMB_OK) {
        fprintf(stderr, "pimemb: lookup: %s\n", emb_last_error());
        return nullptr;
    }
    if (latency_print == 1) {  // emb_host.h:395-402, same six lines
        emb_get_stats(e, &after);
        printf("C: Indices and offsets copying latency: %ldμs\n",
               (long)(after.us_copy_in_indices - before.us_copy_in_indices));
        printf("C: Query copying latency: %ldμs\n",
               (long)(after.us_copy_in_lengths - before.us_copy_in_lengths));
        printf("C: Dpu launch latency: %ldμs\n", (long)(after.us_launch - before.us_launch));
        printf("C: Results copy latency: %ldμs\n", (long)(after.us_copy_out - before.us_copy_out));
        printf("C: Callback prep latency: %ldμs\n", 0L);
        printf("C: DPU sync latency: %ldμs\n", (long)(after.us_sync - before.us_sync));
    }
    return nullptr;  // emb_host.h:403
}

}  // extern "C"
