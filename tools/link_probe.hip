// link_probe.hip -- what the host link gives: 64 MB device -> host into PINNED and into PAGEABLE memory, host -> device from pinned,
// both directions at once (two streams), and a kernel storing rows straight into pinned host memory -- the ceilings of the
// host-pointer lookup path (lookup_host / lookup_host_split in pimemb_engine.cpp).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
__global__ void store_rows(float4 *dst, const float4 *src, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t n = 64u << 20, in_n = 8u << 20;
    char *d, *d2, *pin, *pin_in;
    CK(hipMalloc((void **)&d, n)); CK(hipMalloc((void **)&d2, n)); CK(hipMemset(d, 1, n));
    CK(hipHostMalloc((void **)&pin, n, hipHostMallocMapped)); CK(hipHostMalloc((void **)&pin_in, in_n, hipHostMallocMapped));
    memset(pin, 0, n); memset(pin_in, 1, in_n);
    std::vector<char> page(n + 8192, 0), page2(n + 8192, 0);
    hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    auto timed = [&](const char *what, size_t bytes, auto fn) {
        for (int w = 0; w < 3; w++) fn();
        const int reps = 10;
        const double t0 = now_us();
        for (int r = 0; r < reps; r++) fn();
        const double us = (now_us() - t0) / reps;
        printf("%-78s %8.1f us  %6.2f GB/s\n", what, us, bytes / us / 1e3);
    };
    timed("D2H 64 MB -> pinned", n, [&] { CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
    timed("D2H 64 MB -> pageable", n, [&] { CK(hipMemcpyAsync(page.data(), d, n, hipMemcpyDeviceToHost, s1)); CK(hipStreamSynchronize(s1)); });
    timed("D2H 26 x 2.5 MB -> pageable (as the lookup does)", 26 * 2514688ull, [&] {
        for (int t = 0; t < 26; t++) CK(hipMemcpyAsync(page.data() + (size_t)t * 2514688, d + (size_t)t * 2514688, 2514688, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1)); });
    timed("D2H 26 x 2.5 MB -> pinned", 26 * 2514688ull, [&] {
        for (int t = 0; t < 26; t++) CK(hipMemcpyAsync(pin + (size_t)t * 2514688, d + (size_t)t * 2514688, 2514688, hipMemcpyDeviceToHost, s1));
        CK(hipStreamSynchronize(s1)); });
    timed("H2D 8 MB <- pinned", in_n, [&] { CK(hipMemcpyAsync(d2, pin_in, in_n, hipMemcpyHostToDevice, s2)); CK(hipStreamSynchronize(s2)); });
    timed("D2H 64 MB -> pinned WHILE H2D 8 MB x 8 <- pinned (bytes: the D2H's)", n, [&] {
        CK(hipMemcpyAsync(pin, d, n, hipMemcpyDeviceToHost, s1));
        for (int k = 0; k < 8; k++) CK(hipMemcpyAsync(d2, pin_in, in_n, hipMemcpyHostToDevice, s2));
        CK(hipStreamSynchronize(s1)); CK(hipStreamSynchronize(s2)); });
    timed("kernel stores 64 MB into pinned host memory", n, [&] {
        hipLaunchKernelGGL(store_rows, dim3(2048), dim3(256), 0, s1, (float4 *)pin, (const float4 *)d, n / 16); CK(hipStreamSynchronize(s1)); });
    for (int nt : {1, 2, 4, 8, 16}) {
        char buf[128];
        snprintf(buf, sizeof buf, "host memcpy 64 MB pinned -> pageable, %d threads", nt);
        timed(buf, n, [&] {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back([&, k] { memcpy(page2.data() + n / nt * k, pin + n / nt * k, n / nt); });
            for (auto &t : th) t.join(); });
    }
    // the 26 per-table copies issued by several host threads, each on a stream of its own
    for (int nt : {2, 3, 4}) {
        static hipStream_t ts[4] = {nullptr, nullptr, nullptr, nullptr};
        for (int k = 0; k < nt; k++) if (!ts[k]) CK(hipStreamCreateWithFlags(&ts[k], hipStreamNonBlocking));
        char buf[160];
        snprintf(buf, sizeof buf, "D2H 26 x 2.5 MB -> pageable, %d host threads x 1 stream each (tables dealt round robin)", nt);
        timed(buf, 26 * 2514688ull, [&] {
            std::vector<std::thread> th;
            for (int k = 0; k < nt; k++) th.emplace_back([&, k] {
                for (int t = k; t < 26; t += nt) CK(hipMemcpyAsync(page.data() + (size_t)t * 2514688, d + (size_t)t * 2514688, 2514688, hipMemcpyDeviceToHost, ts[k]));
                CK(hipStreamSynchronize(ts[k])); });
            for (auto &t : th) t.join(); });
    }
    // one big D2H into pinned staging, then 4 threads unpack (no overlap), and with the copy cut in 2 / 4 pieces (unpack k behind copy k+1)
    for (int np : {1, 2, 4}) {
        char buf[160];
        snprintf(buf, sizeof buf, "D2H 64 MB -> pinned in %d piece(s), 4 threads unpack piece k while k+1 arrives (-> pageable)", np);
        timed(buf, n, [&] {
            const size_t piece = n / np;
            hipEvent_t ev[4];
            for (int k = 0; k < np; k++) { CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); CK(hipMemcpyAsync(pin + k * piece, d + k * piece, piece, hipMemcpyDeviceToHost, s1)); CK(hipEventRecord(ev[k], s1)); }
            for (int k = 0; k < np; k++) {
                CK(hipEventSynchronize(ev[k]));
                std::vector<std::thread> th;
                for (int j = 0; j < 4; j++) th.emplace_back([&, j, k] { memcpy(page2.data() + k * piece + piece / 4 * j, pin + k * piece + piece / 4 * j, piece / 4); });
                for (auto &t : th) t.join();
                CK(hipEventDestroy(ev[k]));
            } });
    }
    // D2H into pinned staging in 4 MB pieces, host threads unpacking piece k while piece k+1 arrives
    for (int nt : {2, 4, 8}) {
        char buf[160];
        snprintf(buf, sizeof buf, "D2H 64 MB -> pinned in 4 MB pieces + %d threads unpacking behind it (-> pageable)", nt);
        timed(buf, n, [&] {
            const size_t piece = 4u << 20, np = n / piece;
            std::vector<hipEvent_t> ev(np);
            for (size_t k = 0; k < np; k++) { CK(hipEventCreateWithFlags(&ev[k], hipEventDisableTiming)); CK(hipMemcpyAsync(pin + k * piece, d + k * piece, piece, hipMemcpyDeviceToHost, s1)); CK(hipEventRecord(ev[k], s1)); }
            for (size_t k = 0; k < np; k++) {
                CK(hipEventSynchronize(ev[k]));
                std::vector<std::thread> th;
                for (int j = 0; j < nt; j++) th.emplace_back([&, j, k] { memcpy(page2.data() + k * piece + piece / nt * j, pin + k * piece + piece / nt * j, piece / nt); });
                for (auto &t : th) t.join();
                CK(hipEventDestroy(ev[k]));
            } });
    }
    return 0;
}
