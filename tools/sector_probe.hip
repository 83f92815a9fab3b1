// sector_probe.hip -- does any load flavour make the L2 fetch less than a 128-byte line for a
// 64-byte random row?  Run under rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum (and
// _64B / _128B in a second pass); each flavour is its own kernel so the counters separate.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
using u32x4 = __attribute__((ext_vector_type(4))) uint32_t;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *p, uint32_t bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

template <int MODE>
__global__ void __launch_bounds__(256) gather64(const char *__restrict__ table, const uint32_t *__restrict__ idx,
                                                uint32_t n, uint32_t *__restrict__ sink, uint64_t table_bytes) {
    const uint32_t gid = blockIdx.x * 256 + threadIdx.x, row_slot = gid >> 2, sub = gid & 3;
    if (row_slot >= n) return;
    const uint64_t off = (uint64_t)idx[row_slot] * 64u + sub * 16u;
    u32x4 v;
    if (MODE == 0) v = *reinterpret_cast<const u32x4 *>(table + off);
    else if (MODE == 1) v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(table + off));
    else if (MODE == 2) v = __hip_atomic_load(reinterpret_cast<const uint32_t *>(table + off), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + u32x4{0, 0, 0, 0};
    else if (MODE == 3) {  // only the first 32 bytes of each row are touched (2 lanes of 4)
        v = u32x4{0, 0, 0, 0};
        if (sub < 2) v = *reinterpret_cast<const u32x4 *>(table + off);
    } else {  // raw buffer loads with cache-policy aux bits
        // rows live in the first 4 GiB only for this mode (32-bit buffer offset)
        __amdgpu_buffer_rsrc_t r = make_rsrc(table, 0xfffffff0u);
        const uint32_t o32 = (uint32_t)off;
        if (MODE == 4) v = __builtin_amdgcn_raw_buffer_load_b128(r, o32, 0, 0);
        else if (MODE == 5) v = __builtin_amdgcn_raw_buffer_load_b128(r, o32, 0, 1);
        else if (MODE == 6) v = __builtin_amdgcn_raw_buffer_load_b128(r, o32, 0, 2);
        else if (MODE == 7) v = __builtin_amdgcn_raw_buffer_load_b128(r, o32, 0, 16);
        else v = __builtin_amdgcn_raw_buffer_load_b128(r, o32, 0, 17);
    }
    if ((v[0] ^ v[1] ^ v[2] ^ v[3]) == 0x12345678u) sink[0] = gid;  // keep the load live
}

int main() {
    const uint64_t rows = 48ull << 20, bytes = rows * 64;  // 3 GiB table, far beyond L2 + Infinity Cache
    const uint32_t n = 1u << 20;
    char *table; uint32_t *idx, *sink;
    CK(hipMalloc((void **)&table, bytes));
    CK(hipMemset(table, 1, bytes));
    CK(hipMalloc((void **)&idx, n * 4));
    CK(hipMalloc((void **)&sink, 64));
    std::vector<uint32_t> h(n);
    uint64_t s = 88172645463325252ull;
    for (auto &x : h) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; x = (uint32_t)(s % rows); }
    CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice));
    dim3 grid(n * 4 / 256), block(256);
#define RUN(M) for (int r = 0; r < 3; r++) hipLaunchKernelGGL(gather64<M>, grid, block, 0, 0, table, idx, n, sink, bytes); CK(hipDeviceSynchronize());
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8)
    printf("done: %u random 64-B rows per launch (=%u KiB useful)\n", n, n * 64 / 1024);
    return 0;
}
