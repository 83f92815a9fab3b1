// CPU-only check of csrc/pimemb_hostcopy.h (built with -fsanitize=thread by tests/test_hostcopy.py):
// many rounds of multi-piece copies of varying sizes, contents verified, copier destroyed and re-created.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "pimemb_hostcopy.h"

int main() {
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int life = 0; life < 3; life++) {
        pimemb::HostCopier copier;
        for (int round = 0; round < 24; round++) {
            const int n_pieces = 1 + (int)(rnd() % 9);
            std::vector<std::vector<unsigned char>> src(n_pieces), dst(n_pieces);
            std::vector<pimemb::CopyPiece> pieces;
            for (int i = 0; i < n_pieces; i++) {
                size_t bytes = (round % 3 == 0) ? rnd() % 5000 : rnd() % (3u << 20);   // small rounds stay single-threaded
                if (rnd() % 7 == 0) bytes = 0;
                src[i].resize(bytes);
                dst[i].assign(bytes + 8, 0xAB);
                for (size_t k = 0; k < bytes; k += 97) src[i][k] = (unsigned char)(rnd() & 0xff);
                pieces.push_back({dst[i].data(), src[i].data(), bytes});
            }
            copier.copy(pieces);
            for (int i = 0; i < n_pieces; i++) {
                for (size_t k = 0; k < src[i].size(); k++)
                    if (dst[i][k] != src[i][k]) { fprintf(stderr, "mismatch life %d round %d piece %d byte %zu\n", life, round, i, k); return 1; }
                for (size_t k = src[i].size(); k < dst[i].size(); k++)
                    if (dst[i][k] != 0xAB) { fprintf(stderr, "overrun life %d round %d piece %d\n", life, round, i); return 1; }
            }
        }
    }
    printf("hostcopy ok\n");
    return 0;
}
