// gather_floor.hip -- the simplest possible one-index-per-bag lookup over WIDE rows: out[b][:] = table[ids[b]][:], one 16-byte
// piece per thread (or PER pieces), non-temporal stores, rows of ROW bytes drawn uniformly from an 8-GB table.  What does plain
// code reach on the C4 one-of-8 share's shape (26 x 16 384 bags x 512-byte rows: 218 MB of rows read at random, 218 MB written),
// and how far is the shipped wave-batch kernel (57-59 us) from it?  Eight batches rotated, HIP events, as bench.py does.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int PER>
__global__ void __launch_bounds__(256) gather_rows(const char *__restrict__ table, const uint32_t *__restrict__ ids, f32x4 *__restrict__ out,
                                                   uint64_t n_pieces, uint32_t pieces_per_row, uint32_t row_bytes) {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * PER) + threadIdx.x;
    f32x4 v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint64_t p = base + (uint64_t)k * 256;
        if (p < n_pieces) v[k] = *reinterpret_cast<const f32x4 *>(table + (uint64_t)ids[p / pieces_per_row] * row_bytes + (p % pieces_per_row) * 16);
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint64_t p = base + (uint64_t)k * 256;
        if (p < n_pieces) __builtin_nontemporal_store(v[k], out + p);
    }
}

template <int PER>
static void run(const char *name, const char *table, uint32_t *const *ids, f32x4 *const *out, uint64_t n_rows, uint32_t row_bytes) {
    const uint32_t ppr = row_bytes / 16;
    const uint64_t n_pieces = n_rows * ppr;
    dim3 grid((uint32_t)((n_pieces + 256 * PER - 1) / (256 * PER))), block(256);
    for (int i = 0; i < 16; i++) hipLaunchKernelGGL((gather_rows<PER>), grid, block, 0, 0, table, ids[i & 7], out[i & 7], n_pieces, ppr, row_bytes);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 400;
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((gather_rows<PER>), grid, block, 0, 0, table, ids[i & 7], out[i & 7], n_pieces, ppr, row_bytes);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const float us = ms * 1000.f / iters;
    printf("%-34s rows %8llu x %4u B  %7.2f us/launch  %6.2f TB/s (rows in + rows out + ids)\n", name, (unsigned long long)n_rows, row_bytes, us,
           (n_rows * (2.0 * row_bytes + 4)) / (us * 1e-6) / 1e12);
}

int main(int argc, char **argv) {
    const uint32_t row_bytes = argc > 1 ? (uint32_t)atoi(argv[1]) : 512;
    const uint64_t n_rows = argc > 2 ? strtoull(argv[2], 0, 10) : 26ull * 16384;
    const uint64_t table_bytes = 8ull << 30, table_rows = table_bytes / row_bytes;
    char *table; CK(hipMalloc((void **)&table, table_bytes)); CK(hipMemset(table, 1, table_bytes));
    uint32_t *ids[8]; f32x4 *out[8];
    std::vector<uint32_t> h(n_rows);
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (int i = 0; i < 8; i++) {
        for (auto &v : h) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; v = (uint32_t)((st >> 11) % table_rows); }
        CK(hipMalloc((void **)&ids[i], n_rows * 4)); CK(hipMemcpy(ids[i], h.data(), n_rows * 4, hipMemcpyHostToDevice));
        CK(hipMalloc((void **)&out[i], n_rows * row_bytes));
    }
    CK(hipDeviceSynchronize());
    for (int rep = 0; rep < 2; rep++) {
        run<1>("16 B per thread", table, ids, out, n_rows, row_bytes);
        run<2>("32 B per thread", table, ids, out, n_rows, row_bytes);
        run<4>("64 B per thread", table, ids, out, n_rows, row_bytes);
        run<8>("128 B per thread", table, ids, out, n_rows, row_bytes);
    }
    return 0;
}
