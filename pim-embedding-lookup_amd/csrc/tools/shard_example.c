/* shard_example.c -- the sharded lookup as ONE call per batch, and the request queue, from plain C99 (no HIP headers):
 * INTEGRATION.md section 3's two snippets as a complete program (tests run it on the GPU).
 *
 * One rank here (comm == NULL: everything is "self"), so the three placements can be shown and checked in one process:
 * table 0 replicated, table 1 whole on rank 0, table 2 split by row range (one shard holding every row).  With N ranks the
 * only additions are emb_comm_unique_id / emb_comm_create (or emb_peer_create) and that every rank makes the same calls. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pimemb.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != EMB_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, emb_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

enum { T = 3, ROWS = 500, DIM = 16, BAGS = 6, NIDX = 11 };

static float tab[T][ROWS][DIM];

static float expect(int t, const uint32_t *idx, int lo, int hi, int c) {
    float acc = 0.0f;
    for (int p = lo; p < hi; p++) acc += tab[t][idx[p]][c];
    return acc;
}

int main(void) {
    for (int t = 0; t < T; t++)
        for (int r = 0; r < ROWS; r++)
            for (int c = 0; c < DIM; c++) tab[t][r][c] = (float)((t + 1) * 1000 + r) + 0.25f * (float)c;
    emb_engine *e = NULL;
    emb_config cfg = {0, 8, 0};
    CHECK(emb_create(&cfg, &e));
    for (int t = 0; t < T; t++) CHECK(emb_load_table(e, (uint32_t)t, ROWS, DIM, EMB_F32, tab[t], EMB_MEM_HOST));

    /* ---- the sharded lookup: name every table's placement once ... */
    emb_shard_table place[T] = {{EMB_PLACE_REPLICATED, -1, 0, 0}, {EMB_PLACE_WHOLE, 0, 1, 0}, {EMB_PLACE_ROWS, -1, 2, ROWS}};
    emb_shard_config scfg;
    memset(&scfg, 0, sizeof scfg);
    scfg.n_tables = T;
    scfg.dim = DIM;
    scfg.depth = 0;
    scfg.flags = EMB_SHARD_CHECK_SERVED;
    scfg.tables = place;
    emb_shard *s = NULL;
    CHECK(emb_shard_create(e, NULL, &scfg, &s));

    /* ... then one call per batch: this rank's bags for every table in, its pooled rows out (buffers resident in HBM) */
    const uint32_t idx[NIDX] = {3, 499, 7, 7, 250, 0, 1, 2, 3, 4, 123}, off[BAGS] = {0, 1, 1, 4, 5, 10}; /* bags: {3} {} {499,7,7} {250} {0..4} {123} */
    void *d_idx, *d_off, *d_out[T];
    CHECK(emb_device_alloc(e, sizeof idx, &d_idx));
    CHECK(emb_device_alloc(e, sizeof off, &d_off));
    CHECK(emb_copy_to_device(e, d_idx, idx, sizeof idx));
    CHECK(emb_copy_to_device(e, d_off, off, sizeof off));
    emb_shard_input in[T];
    memset(in, 0, sizeof in);
    for (int t = 0; t < T; t++) {
        CHECK(emb_device_alloc(e, BAGS * DIM * sizeof(float), &d_out[t]));
        in[t].indices = (const uint32_t *)d_idx;
        in[t].offsets = (const uint32_t *)d_off;
        in[t].n_indices = NIDX;
        in[t].pooled = (float *)d_out[t];
    }
    CHECK(emb_shard_lookup(s, in, BAGS, NULL));
    CHECK(emb_synchronize(e, NULL));
    int bad = 0;
    static float got[BAGS][DIM];
    for (int t = 0; t < T; t++) {
        CHECK(emb_copy_to_host(e, got, d_out[t], sizeof got));
        for (int b = 0; b < BAGS; b++)
            for (int c = 0; c < DIM; c++) bad += got[b][c] != expect(t, idx, (int)off[b], b + 1 < BAGS ? (int)off[b + 1] : NIDX, c);
    }
    emb_shard_stats ss;
    CHECK(emb_shard_get_stats(s, &ss, 0));

    /* a row id outside its table: the serving rank reports it after the batch went through all its stages */
    uint32_t bad_idx[NIDX];
    memcpy(bad_idx, idx, sizeof idx);
    bad_idx[4] = ROWS + 7;
    CHECK(emb_copy_to_device(e, d_idx, bad_idx, sizeof bad_idx));
    const int rc_bad = emb_shard_lookup(s, in, BAGS, NULL);
    bad += rc_bad != EMB_ERR_RANGE;
    CHECK(emb_shard_destroy(s));

    /* ---- the request queue: three small requests (mini-batch 1, host pointers) -> ONE launch */
    emb_queue *q = NULL;
    CHECK(emb_queue_create(e, EMB_IDX_U32, EMB_MEM_HOST, &q));
    uint32_t one_idx[3][T] = {{5, 6, 7}, {100, 200, 300}, {499, 0, 42}}, zero = 0;
    static float one_out[3][T][DIM];
    uint64_t ticket[3];
    for (int r = 0; r < 3; r++) {
        emb_lookup_desc d[T];
        memset(d, 0, sizeof d);
        for (int t = 0; t < T; t++) {
            d[t].table_id = (uint32_t)t;
            d[t].indices = &one_idx[r][t];
            d[t].offsets = &zero;
            d[t].n_indices = 1;
            d[t].n_bags = 1;
            d[t].pooled = one_out[r][t];
        }
        CHECK(emb_queue_add(q, d, T, &ticket[r]));
    }
    uint32_t n_req = 0;
    emb_stats st0, st1;
    CHECK(emb_get_stats(e, &st0));
    CHECK(emb_queue_flush(q, NULL, &n_req));
    for (int r = 0; r < 3; r++) CHECK(emb_queue_wait(q, ticket[r]));
    CHECK(emb_get_stats(e, &st1));
    for (int r = 0; r < 3; r++)
        for (int t = 0; t < T; t++)
            for (int c = 0; c < DIM; c++) bad += one_out[r][t][c] != tab[t][one_idx[r][t]][c];
    bad += n_req != 3 || st1.n_kernel_launches - st0.n_kernel_launches != 1;
    CHECK(emb_queue_destroy(q));

    printf("sharded batch: %llu batch(es), %llu sub-bags served; bad index -> %d (EMB_ERR_RANGE = %d); queue: %u requests in %llu launch; mismatches: %d\n",
           (unsigned long long)ss.n_batches, (unsigned long long)ss.served_sub_bags, rc_bad, EMB_ERR_RANGE, n_req,
           (unsigned long long)(st1.n_kernel_launches - st0.n_kernel_launches), bad);
    CHECK(emb_device_free(e, d_idx));
    CHECK(emb_device_free(e, d_off));
    for (int t = 0; t < T; t++) CHECK(emb_device_free(e, d_out[t]));
    CHECK(emb_destroy(e));
    return bad ? 2 : 0;
}
