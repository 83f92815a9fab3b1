/* native_example.c -- the runtime-shaped native API from plain C99 (no HIP headers, no C++):
 * create an engine, upload two fp32 tables, run one fused two-table lookup with host pointers and
 * one with buffers kept in HBM through the emb_device_* helpers, print the pooled rows.
 * This is the INTEGRATION.md section 3 snippet as a complete program (tests run it on the GPU). */
#include <stdio.h>
#include <stdlib.h>

#include "pimemb.h"

#define CHECK(call)                                                              \
    do {                                                                         \
        int rc_ = (call);                                                        \
        if (rc_ != EMB_OK) {                                                     \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, emb_last_error());     \
            return 1;                                                            \
        }                                                                        \
    } while (0)

int main(void) {
    enum { ROWS = 100, DIM = 16, BAGS = 4 };
    static float t0[ROWS][DIM], t1[ROWS][DIM];
    for (int r = 0; r < ROWS; r++)
        for (int c = 0; c < DIM; c++) {
            t0[r][c] = (float)(r * 100 + c);
            t1[r][c] = -(float)(r + c);
        }
    emb_engine *e = NULL;
    emb_config cfg = {0, 2, 0};
    CHECK(emb_create(&cfg, &e));
    CHECK(emb_load_table(e, 0, ROWS, DIM, EMB_F32, t0, EMB_MEM_HOST));
    CHECK(emb_load_table(e, 1, ROWS, DIM, EMB_F32, t1, EMB_MEM_HOST));

    /* bags: {3}, {}, {5, 5, 7}, {99}  -- last bag runs to n_indices */
    uint32_t idx[5] = {3, 5, 5, 7, 99}, off[BAGS] = {0, 1, 1, 4};
    static float out0[BAGS][DIM], out1[BAGS][DIM];
    emb_lookup_desc d[2] = {{0, 0, idx, off, 5, BAGS, &out0[0][0]}, {1, 0, idx, off, 5, BAGS, &out1[0][0]}};
    CHECK(emb_lookup_batched(e, d, 2, EMB_IDX_U32, EMB_MEM_HOST, NULL));
    int bad = 0;
    for (int c = 0; c < DIM; c++) {
        bad += out0[0][c] != t0[3][c];
        bad += out0[1][c] != 0.0f;
        bad += out0[2][c] != (t0[5][c] + t0[5][c]) + t0[7][c];
        bad += out0[3][c] != t0[99][c];
        bad += out1[2][c] != (t1[5][c] + t1[5][c]) + t1[7][c];
    }

    /* the same lookup with everything resident in HBM: a prepared plan, launched twice */
    void *d_idx, *d_off, *d_out;
    CHECK(emb_device_alloc(e, sizeof idx, &d_idx));
    CHECK(emb_device_alloc(e, sizeof off, &d_off));
    CHECK(emb_device_alloc(e, sizeof out0, &d_out));
    CHECK(emb_copy_to_device(e, d_idx, idx, sizeof idx));
    CHECK(emb_copy_to_device(e, d_off, off, sizeof off));
    emb_lookup_desc dd = {0, 0, d_idx, d_off, 5, BAGS, (float *)d_out};
    emb_plan *p = NULL;
    CHECK(emb_plan_create(e, &dd, 1, EMB_IDX_U32, &p));
    CHECK(emb_plan_launch(p, NULL));
    CHECK(emb_plan_launch(p, NULL));
    CHECK(emb_synchronize(e, NULL));
    static float back[BAGS][DIM];
    CHECK(emb_copy_to_host(e, back, d_out, sizeof back));
    for (int b = 0; b < BAGS; b++)
        for (int c = 0; c < DIM; c++) bad += back[b][c] != out0[b][c];

    emb_stats st;
    CHECK(emb_get_stats(e, &st));
    printf("pooled[2][0..3] = %g %g %g %g ; %llu lookups, %llu kernel launches, mismatches: %d\n", out0[2][0],
           out0[2][1], out0[2][2], out0[2][3], (unsigned long long)st.n_lookup_calls,
           (unsigned long long)st.n_kernel_launches, bad);
    CHECK(emb_plan_destroy(p));
    CHECK(emb_device_free(e, d_idx));
    CHECK(emb_device_free(e, d_off));
    CHECK(emb_device_free(e, d_out));
    CHECK(emb_destroy(e));
    return bad ? 2 : 0;
}
