// stream_floor.hip -- practical speed of light for the C2 launch's HBM-side traffic mix.
// The shipped kernel moves 48.9 MB of reads and 65.4 MB of writes per launch across the fabric
// (profiles/traffic.json).  How long do perfectly coalesced streaming kernels take to move the same
// bytes?  Back-to-back launches on one stream, rotating over 8 buffer sets (as bench.py rotates 8
// batches), HIP events over the whole run -- the same clock bench.py uses for roofline.kernel_us.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

// every thread moves PER 16-byte pieces, consecutive threads touch consecutive pieces
template <int PER, bool NT>
__global__ void __launch_bounds__(256) stream_mix(const u32x4 *__restrict__ src, uint64_t n_read,
                                                  u32x4 *__restrict__ dst, uint64_t n_write, uint32_t *sink) {
    const uint64_t base = (uint64_t)blockIdx.x * (256 * PER) + threadIdx.x;
    u32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint64_t i = base + (uint64_t)k * 256;
        if (i < n_read) acc ^= src[i];
    }
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const uint64_t i = base + (uint64_t)k * 256;
        if (i < n_write) {
            u32x4 v = {(uint32_t)i, acc[1], 2u, 3u};
            if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v;
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

template <int PER, bool NT>
static float run(const char *name, u32x4 *const *src, uint64_t rbytes, u32x4 *const *dst, uint64_t wbytes,
                 uint32_t *sink, int iters) {
    const uint64_t nr = rbytes / 16, nw = wbytes / 16, n = nr > nw ? nr : nw;
    dim3 grid((uint32_t)((n + 256 * PER - 1) / (256 * PER))), block(256);
    for (int i = 0; i < 16; i++) hipLaunchKernelGGL((stream_mix<PER, NT>), grid, block, 0, 0, src[i & 7], nr, dst[i & 7], nw, sink);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < iters; i++) hipLaunchKernelGGL((stream_mix<PER, NT>), grid, block, 0, 0, src[i & 7], nr, dst[i & 7], nw, sink);
    CK(hipEventRecord(b, 0));
    CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const float us = ms * 1000.f / iters;
    printf("%-44s read %6.1f MB  write %6.1f MB  %7.2f us/launch  %6.2f TB/s  (grid %u)\n", name, rbytes / 1e6,
           wbytes / 1e6, us, (rbytes + wbytes) / (us * 1e-6) / 1e12, grid.x);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    return us;
}

int main(int argc, char **argv) {
    const uint64_t R = argc > 1 ? strtoull(argv[1], 0, 10) : 48869523, W = argc > 2 ? strtoull(argv[2], 0, 10) : 65440153;
    const uint64_t cap = ((R > W ? R : W) * 2 + 4095) / 4096 * 4096;
    u32x4 *src[8], *dst[8]; uint32_t *sink;
    for (int i = 0; i < 8; i++) {
        CK(hipMalloc((void **)&src[i], cap)); CK(hipMemset(src[i], 1, cap));
        CK(hipMalloc((void **)&dst[i], cap)); CK(hipMemset(dst[i], 2, cap));
    }
    CK(hipMalloc((void **)&sink, 64));
    CK(hipDeviceSynchronize());
    const int it = 400;
    for (int rep = 0; rep < 2; rep++) {
        run<4, true>("C2 mix, nt stores, 64 B/thread", src, R, dst, W, sink, it);
        run<4, false>("C2 mix, plain stores, 64 B/thread", src, R, dst, W, sink, it);
        run<1, true>("C2 mix, nt stores, 16 B/thread", src, R, dst, W, sink, it);
        run<8, true>("C2 mix, nt stores, 128 B/thread", src, R, dst, W, sink, it);
        run<4, true>("writes only (the 65.4 MB of pooled rows)", src, 0, dst, W, sink, it);
        run<4, true>("reads only (48.9 MB)", src, R, dst, 0, sink, it);
        run<4, true>("copy, same total (57.2 MB each way)", src, (R + W) / 2, dst, (R + W) / 2, sink, it);
        run<4, true>("copy, 2x the larger stream each way", src, cap, dst, cap, sink, 50);
    }
    return 0;
}
