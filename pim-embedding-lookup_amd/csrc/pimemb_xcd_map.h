// pimemb_xcd_map.h -- host-side planner of the XCD-aware workgroup -> (table, tile) map.
//
// MI355X deals the workgroups of a 1-D grid round-robin over its 8 XCDs (blocks n and n+8 share
// one XCD and therefore one 4 MiB L2; MI355X_MICROARCH.md "Workgroup dispatch").  A fused
// multi-table lookup that walks tables in grid order spreads every table over all 8 L2s, so each
// L2 has to hold the hot rows of EVERY table.  The map below gives each residue class n % 8 a
// contiguous share of the concatenated (table, tile) list instead: a table is then served by one
// (at most two) XCDs, whose L2 keeps its rows across launches, while every class gets the same
// number of tiles (+-1) so the XCDs finish together.  Placement is a speed matter only: results do
// not depend on which XCD runs a workgroup.
#pragma once

#include <stdint.h>

#include <algorithm>
#include <numeric>
#include <vector>

namespace pimemb {

constexpr uint32_t kXcdGroups = 8;

// One contiguous run of tiles of one descriptor, owned by one residue class.
struct XcdSeg {
    uint32_t desc;        // descriptor index within the launch group
    uint32_t tile0;       // first tile of the run
    uint32_t slot_begin;  // class-local slot of tile0
    uint32_t slot_end;    // one past the last slot of the run
};

// Tables no larger than this are "cacheable": worth pinning to one XCD's L2 (4 MiB, shared with
// whatever else that XCD streams).  Larger tables miss in L2 wherever they run.
constexpr uint64_t kXcdCacheableBytes = 32ull << 20;

// Device image: words[0..7] = segments per class, words[8..15] = first segment of the class,
// words[16..] = XcdSeg array.  Returns the 1-D grid size (8 * slots per class).
//
// Balance rules (measured on MI355X: a class that only holds HBM-missing tables runs far longer
// than one that only holds L2-resident ones, so equal tile COUNTS are not enough):
//   * streaming (non-cacheable) tables: every class gets the same share of EACH of them, so the
//     L2-miss traffic is spread evenly over the 8 XCDs' fabric ports;
//   * cacheable tables: placed whole, largest first, on the class with the fewest cacheable bytes
//     that still has tile room; a table that fits nowhere whole is split over two classes;
//   * within a class the two kinds are interleaved in `rounds` rounds so misses and hits overlap.
inline uint32_t build_xcd_map(const std::vector<uint32_t> &n_tiles, const std::vector<uint64_t> &table_bytes,
                              std::vector<uint32_t> *words, uint32_t rounds = 1,
                              uint64_t cacheable_bytes = kXcdCacheableBytes) {
    const uint32_t n = (uint32_t)n_tiles.size();
    struct Run { uint32_t desc, tile0, count; };
    std::vector<std::vector<Run>> cache_runs(kXcdGroups), stream_runs(kXcdGroups);

    // streaming tables: equal share of each to every class
    std::vector<uint32_t> cacheable;
    uint64_t cache_tiles = 0;
    for (uint32_t d = 0; d < n; d++) {
        if (table_bytes[d] <= cacheable_bytes) {
            cacheable.push_back(d);
            cache_tiles += n_tiles[d];
            continue;
        }
        uint32_t tile = 0;
        for (uint32_t c = 0; c < kXcdGroups; c++) {
            const uint32_t take = n_tiles[d] / kXcdGroups + (c < n_tiles[d] % kXcdGroups ? 1u : 0u);
            if (take) stream_runs[(c + d) % kXcdGroups].push_back(Run{d, tile, take});
            tile += take;
        }
    }
    // cacheable tables: whole tables, largest first, least-loaded (by bytes) class with room
    std::stable_sort(cacheable.begin(), cacheable.end(),
                     [&](uint32_t a, uint32_t b) { return table_bytes[a] > table_bytes[b]; });
    std::vector<uint32_t> room(kXcdGroups);
    for (uint32_t c = 0; c < kXcdGroups; c++)
        room[c] = (uint32_t)(cache_tiles / kXcdGroups + (c < cache_tiles % kXcdGroups ? 1u : 0u));
    std::vector<uint64_t> held(kXcdGroups, 0);
    for (uint32_t d : cacheable) {
        uint32_t tile = 0, left = n_tiles[d];
        while (left) {
            int best = -1;
            for (uint32_t c = 0; c < kXcdGroups; c++)  // whole fit, fewest bytes held
                if (room[c] >= left && (best < 0 || held[c] < held[best])) best = (int)c;
            if (best < 0)                                // no whole fit: most room
                for (uint32_t c = 0; c < kXcdGroups; c++)
                    if (room[c] && (best < 0 || room[c] > room[best])) best = (int)c;
            const uint32_t take = std::min(left, room[best]);
            cache_runs[best].push_back(Run{d, tile, take});
            held[best] += table_bytes[d];
            room[best] -= take;
            tile += take;
            left -= take;
        }
    }
    // interleave the two kinds per class
    std::vector<std::vector<XcdSeg>> segs(kXcdGroups);
    uint32_t max_slots = 0;
    for (uint32_t c = 0; c < kXcdGroups; c++) {
        uint32_t slot = 0;
        auto emit = [&](std::vector<Run> &runs, size_t &ri, uint32_t &used, uint32_t want) {
            while (want && ri < runs.size()) {
                const uint32_t take = std::min(want, runs[ri].count - used);
                segs[c].push_back(XcdSeg{runs[ri].desc, runs[ri].tile0 + used, slot, slot + take});
                slot += take;
                used += take;
                want -= take;
                if (used == runs[ri].count) {
                    ri++;
                    used = 0;
                }
            }
        };
        uint32_t tot_s = 0, tot_c = 0;
        for (const Run &r : stream_runs[c]) tot_s += r.count;
        for (const Run &r : cache_runs[c]) tot_c += r.count;
        size_t si = 0, ci = 0;
        uint32_t su = 0, cu = 0;
        for (uint32_t r = 0; r < rounds; r++) {
            emit(stream_runs[c], si, su, tot_s / rounds + (r < tot_s % rounds ? 1u : 0u));
            emit(cache_runs[c], ci, cu, tot_c / rounds + (r < tot_c % rounds ? 1u : 0u));
        }
        max_slots = std::max(max_slots, slot);
    }
    words->assign(16, 0u);
    uint32_t base = 0;
    for (uint32_t c = 0; c < kXcdGroups; c++) {
        (*words)[c] = (uint32_t)segs[c].size();
        (*words)[8 + c] = base;
        base += (uint32_t)segs[c].size();
        for (const XcdSeg &s : segs[c]) {
            words->push_back(s.desc);
            words->push_back(s.tile0);
            words->push_back(s.slot_begin);
            words->push_back(s.slot_end);
        }
    }
    return max_slots * kXcdGroups;
}

// Expand a segment map (build_xcd_map) into the per-workgroup table the kernels read with ONE scalar
// load: entry[block] = {descriptor, tile}, 0xffffffff = idle slot.  8 bytes per workgroup.
inline void expand_xcd_map(const std::vector<uint32_t> &words, uint32_t grid, std::vector<uint32_t> *direct) {
    direct->assign((size_t)grid * 2, 0xffffffffu);
    for (uint32_t cls = 0; cls < kXcdGroups; cls++) {
        const uint32_t ns = words[cls], base = words[8 + cls];
        for (uint32_t i = 0; i < ns; i++) {
            const uint32_t *sg = &words[16 + 4 * (base + i)];
            for (uint32_t slot = sg[2]; slot < sg[3]; slot++) {
                const uint32_t block = slot * kXcdGroups + cls;
                if (block < grid) {
                    (*direct)[2 * (size_t)block] = sg[0];
                    (*direct)[2 * (size_t)block + 1] = sg[1] + (slot - sg[2]);
                }
            }
        }
    }
}

}  // namespace pimemb
