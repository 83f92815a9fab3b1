"""CPU: the XCD-aware workgroup map (host logic, header-only C++): every (table, tile) is covered
exactly once, residue classes are balanced, streaming tables are spread over all classes and
cacheable tables stay together.  The device-side decode is mirrored here line for line."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHECKER = r'''
#include <cstdio>
#include <cstdlib>
#include <map>
#include <set>
#include "pimemb_xcd_map.h"
using namespace pimemb;
// mirror of decode_block() in pimemb_bag_kernels.h
static bool decode(const std::vector<uint32_t>& w, uint32_t block, uint32_t* desc, uint32_t* tile) {
    uint32_t cls = block & 7u, slot = block >> 3, ns = w[cls], base = w[8 + cls];
    uint32_t lo = 0, hi = ns;
    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; if (slot < w[16 + 4 * (base + mid) + 3]) hi = mid; else lo = mid + 1; }
    if (lo >= ns) return false;
    const uint32_t* sg = &w[16 + 4 * (base + lo)];
    *desc = sg[0]; *tile = sg[1] + (slot - sg[2]);
    return true;
}
int main(int argc, char** argv) {
    std::vector<uint32_t> tiles; std::vector<uint64_t> bytes;
    for (int i = 1; i + 1 < argc; i += 2) { tiles.push_back(atoi(argv[i])); bytes.push_back(strtoull(argv[i + 1], 0, 10)); }
    std::vector<uint32_t> w;
    uint32_t grid = build_xcd_map(tiles, bytes, &w, 1);
    std::vector<uint32_t> direct;
    expand_xcd_map(w, grid, &direct);   // the per-workgroup table must agree with the segment decode
    for (uint32_t b = 0; b < grid; b++) {
        uint32_t d, t;
        bool ok = decode(w, b, &d, &t);
        if (ok != (direct[2 * b] != 0xffffffffu) || (ok && (direct[2 * b] != d || direct[2 * b + 1] != t))) {
            printf("DIRECT_MISMATCH at %u\n", b);
            return 1;
        }
    }
    std::map<std::pair<uint32_t, uint32_t>, int> seen;
    std::vector<uint32_t> per_class(8, 0);
    std::vector<std::set<uint32_t>> classes_of(tiles.size());
    for (uint32_t b = 0; b < grid; b++) {
        uint32_t d, t;
        if (!decode(w, b, &d, &t)) continue;
        if (d >= tiles.size() || t >= tiles[d]) { printf("OUT_OF_RANGE\n"); return 1; }
        seen[{d, t}]++;
        per_class[b & 7]++;
        classes_of[d].insert(b & 7);
    }
    uint64_t total = 0; for (auto t : tiles) total += t;
    if (seen.size() != total) { printf("COVERAGE %zu != %llu\n", seen.size(), (unsigned long long)total); return 1; }
    for (auto& kv : seen) if (kv.second != 1) { printf("DUPLICATE\n"); return 1; }
    uint32_t mn = ~0u, mx = 0; for (auto c : per_class) { mn = c < mn ? c : mn; mx = c > mx ? c : mx; }
    printf("grid %u total %llu min %u max %u\n", grid, (unsigned long long)total, mn, mx);
    for (size_t d = 0; d < tiles.size(); d++) printf("table %zu classes %zu\n", d, classes_of[d].size());
    return 0;
}
'''


def _run(tmp_path, pairs):
    src = tmp_path / "chk.cpp"
    exe = tmp_path / "chk"
    if not exe.exists():
        src.write_text(CHECKER)
        subprocess.check_call(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined",
                               "-fno-sanitize-recover=all", "-I",
                               os.path.join(ROOT, "pim-embedding-lookup_amd", "csrc"), str(src), "-o", str(exe)])
    args = [str(x) for p in pairs for x in p]
    out = subprocess.check_output([str(exe)] + args, text=True)
    head = out.splitlines()[0].split()
    info = dict(grid=int(head[1]), total=int(head[3]), mn=int(head[5]), mx=int(head[7]))
    info["classes"] = [int(l.split()[-1]) for l in out.splitlines()[1:]]
    return info


def test_kaggle_shape(tmp_path, pel):
    rows = pel.workloads.KAGGLE_ROWS
    tiles = -(-39292 // 64)                       # 64-bag tiles of the shipped wave-batch kernel
    info = _run(tmp_path, [(tiles, n * 64) for n in rows])
    assert info["total"] == 26 * tiles
    assert info["mx"] - info["mn"] <= 8           # balanced residue classes
    assert info["grid"] <= info["total"] + 8 * 8
    for n, c in zip(rows, info["classes"]):
        if n * 64 > (32 << 20):
            assert c == 8                          # streaming tables: a share on every class
        elif n * 64 > (1 << 20):
            assert c <= 2                          # mid-size cacheable tables: one (at most two) L2s
        else:
            assert c <= 4                          # the smallest ones fill the gaps (cheap to duplicate)


def test_degenerate_shapes(tmp_path):
    assert _run(tmp_path, [(1, 64)])["total"] == 1
    info = _run(tmp_path, [(5, 1 << 40), (3, 100), (0, 100), (1000, 1 << 20)])
    assert info["total"] == 1008 and info["classes"][2] == 0
    info = _run(tmp_path, [(7, 10)] * 40)          # many small equal tables
    assert info["total"] == 280 and info["mx"] - info["mn"] <= 1
