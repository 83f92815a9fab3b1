// pimemb_hot_rows.h -- host-side builder of a table's hot-row set for bag_sum_hot_kernel:
// de-duplicates the caller's list, caps it to the LDS budget, and lays the ids out in a small
// open-addressing hash (linear probing, at most kHotProbes probes -- a row that cannot be placed
// that close to its home slot is dropped from the hot set, it is then simply read from HBM/L2).
#pragma once

#include <stdint.h>

#include <unordered_set>
#include <vector>

namespace pimemb {

struct HotSet {
    std::vector<uint64_t> rows;   // accepted hot rows, slot order (slot i <-> rows[i])
    std::vector<uint64_t> hash;   // entries (slot << 32) | row id, empty = ~0ull
    uint32_t log2size = 0;
    size_t lds_bytes(uint32_t row_bytes) const { return rows.size() * (size_t)row_bytes + hash.size() * 8; }
};

// ids: candidate hot rows, hottest first.  lds_budget: bytes of LDS the launch may use per workgroup.
inline HotSet build_hot_set(const uint64_t *ids, uint32_t n, uint64_t nr_rows, uint32_t row_bytes,
                            size_t lds_budget) {
    HotSet hs;
    if (n == 0 || row_bytes == 0) return hs;
    // hash gets 4 slots per row (8 B each): budget = rows*(row_bytes + 32) (+ rounding to a power of two)
    size_t max_rows = lds_budget / ((size_t)row_bytes + 64);
    if (max_rows > n) max_rows = n;
    if (max_rows == 0) return hs;
    uint32_t log2 = 2;
    while ((1ull << log2) < 4 * max_rows) log2++;
    while (max_rows && max_rows * (size_t)row_bytes + (8ull << log2) > lds_budget) max_rows--;
    const uint32_t mask = (1u << log2) - 1u;
    hs.log2size = log2;
    hs.hash.assign((size_t)1 << log2, ~0ull);
    std::unordered_set<uint64_t> seen;
    for (uint32_t i = 0; i < n && hs.rows.size() < max_rows; i++) {
        const uint64_t r = ids[i];
        if (r >= nr_rows || r >= 0xffffffffull || !seen.insert(r).second) continue;
        const uint32_t key = (uint32_t)r, home = (key * 0x9E3779B1u) >> (32u - log2);
        for (uint32_t p = 0; p < 2; p++) {   // kHotProbes
            uint64_t &e = hs.hash[(home + p) & mask];
            if (e == ~0ull) {
                e = ((uint64_t)hs.rows.size() << 32) | key;
                hs.rows.push_back(r);
                break;
            }
        }
    }
    if (hs.rows.empty()) {
        hs.hash.clear();
        hs.log2size = 0;
    }
    return hs;
}

}  // namespace pimemb
