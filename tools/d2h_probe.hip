// d2h_probe.hip -- how fast does hipMemcpyAsync move N bytes from HBM into PAGEABLE host memory?
// (the host-pointer lookup path copies each table's pooled rows that way when the pieces are large)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1);} } while (0)
int main() {
    const size_t cap = 64u << 20;
    char *d; CK(hipMalloc((void **)&d, cap)); CK(hipMemset(d, 1, cap));
    std::vector<char> h(cap, 0);
    const size_t sizes[] = {64u << 10, 256u << 10, 512u << 10, 1000000, 1u << 20, 1200000, 1300000, 1400000, 1500000, 1536000,
                            3u << 19, 1700000, 2u << 20, 4u << 20, 16u << 20, 64u << 20};
    for (size_t n : sizes) {
        for (int w = 0; w < 3; w++) { CK(hipMemcpyAsync(h.data(), d, n, hipMemcpyDeviceToHost, 0)); }
        CK(hipStreamSynchronize(0));
        const int reps = 20;
        auto t0 = std::chrono::steady_clock::now();
        for (int r = 0; r < reps; r++) CK(hipMemcpyAsync(h.data() + (r % 2) * 4096, d, n, hipMemcpyDeviceToHost, 0));
        CK(hipStreamSynchronize(0));
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / reps;
        printf("%9zu B  %8.1f us  %6.2f GB/s\n", n, us, n / us / 1e3);
    }
    return 0;
}
