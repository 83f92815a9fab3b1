"""CPU: the host-side parallel copier of the host-pointer path (csrc/pimemb_hostcopy.h) under
ThreadSanitizer -- contents, no overruns, no data races, clean start/stop of the worker threads."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("sanitizer", ["thread", "address"])
def test_host_copier_under_sanitizers(tmp_path, sanitizer):
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    exe = tmp_path / ("hostcopy_" + sanitizer)
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=" + sanitizer, "-pthread",
           "-I", os.path.join(ROOT, "pim-embedding-lookup_amd", "csrc"),
           os.path.join(ROOT, "tests", "cpp", "hostcopy_check.cpp"), "-o", str(exe)]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1")
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0 and "hostcopy ok" in run.stdout, run.stdout[-1000:] + run.stderr[-3000:]


def test_hot_set_builder_under_sanitizers(tmp_path):
    """csrc/pimemb_hot_rows.h (host side of emb_set_hot_rows) with ASan + UBSan: every accepted row is
    found again within two probes, duplicates / out-of-range ids are dropped, the LDS budget holds."""
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = tmp_path / "hot.cpp"
    src.write_text(r"""
#include <cstdio>
#include <vector>
#include "pimemb_hot_rows.h"
int main() {
    uint64_t s = 1234567;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int it = 0; it < 200; it++) {
        const uint64_t nr_rows = 1 + rnd() % 100000;
        const uint32_t row_bytes = 16u * (1 + (uint32_t)(rnd() % 64)), n = (uint32_t)(rnd() % 3000);
        const size_t budget = 1024 + rnd() % (62u << 10);
        std::vector<uint64_t> ids(n);
        for (auto &v : ids) v = (rnd() % 5 == 0) ? nr_rows + rnd() % 10 : rnd() % nr_rows;
        if (n > 4) { ids[3] = ids[0]; ids[n - 1] = 0xffffffffffull; }
        pimemb::HotSet hs = pimemb::build_hot_set(ids.data(), n, nr_rows, row_bytes, budget);
        if (hs.rows.empty()) continue;
        if (hs.lds_bytes(row_bytes) > budget) { printf("over budget\n"); return 1; }
        const uint32_t mask = (1u << hs.log2size) - 1u;
        for (size_t slot = 0; slot < hs.rows.size(); slot++) {
            const uint64_t r = hs.rows[slot];
            if (r >= nr_rows) { printf("out-of-range row accepted\n"); return 1; }
            const uint32_t key = (uint32_t)r, home = (key * 0x9E3779B1u) >> (32u - hs.log2size);
            bool found = false;
            for (uint32_t p = 0; p < 2; p++) {
                const uint64_t e = hs.hash[(home + p) & mask];
                if ((uint32_t)e == key && (e >> 32) == slot) found = true;
            }
            if (!found) { printf("row %llu not within two probes\n", (unsigned long long)r); return 1; }
            for (size_t o = 0; o < slot; o++) if (hs.rows[o] == r) { printf("duplicate\n"); return 1; }
        }
    }
    printf("hot set ok\n");
    return 0;
}
""")
    exe = tmp_path / "hot"
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
                           "-I", os.path.join(ROOT, "pim-embedding-lookup_amd", "csrc"), str(src), "-o", str(exe)])
    run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and "hot set ok" in run.stdout, run.stdout + run.stderr[-2000:]


@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_library_host_logic_under_sanitizers(tmp_path, sanitizer):
    """The HOST side of the library's own sources -- engine (launch-image rings from several caller threads, plans,
    host-pointer calls, checked calls with one thread handing in bad indices), the request queue (adders, a free-running
    flusher, late collectors), the sharded call with one rank (every placement, depth 0-3, routed and direct batches,
    peer-store mode, CHECKED shards whose counted direct path must name the one bad index on the requesting rank),
    with 2 / 3 and with EIGHT ranks as threads over a stand-in for emb_comm (the C4 placement mix: 20 replicated / 6
    row-split; the C5 mix: every table whole on an owner; ragged, one-index, empty batches) and -- ASan leg -- in one peer
    group of 2 / 3 / 8 ranks (the collective-free exchange: the stub plays the mailbox kernels and the served counters),
    EVERYTHING AGAIN ON DEVICE 1 of a two-device stub that counts every HIP call made with another device current (none may
    be: engine, queues and their client / flusher threads, shards, peer groups), a runtime without fine-grained memory
    (a group with peers refuses the silent fallback), populate_mram / lookup -- compiled host-only (`--cuda-host-only`) and linked against
    tests/cpp/hip_runtime_stub.cpp instead of the HIP runtime: kernels are no-ops there except the signalling ones, so what
    is checked is return codes, tickets, ordering and that ThreadSanitizer / AddressSanitizer + UBSan stay silent.  (Removing
    the launch-ring lock makes the TSan leg fail: tried.)  Nothing of this is linked into libpimemb.so."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "obj"
    build = subprocess.run(["bash", os.path.join(ROOT, "tests", "cpp", "build_host_logic_check.sh"), sanitizer, str(out)],
                           capture_output=True, text=True, timeout=900)
    if build.returncode != 0 and "libclang_rt" in build.stderr and "No such file" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, TSAN_OPTIONS="halt_on_error=1", ASAN_OPTIONS="detect_leaks=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    run = subprocess.run([str(out / "host_logic_check")], capture_output=True, text=True, timeout=900, env=env)
    assert run.returncode == 0 and "host logic ok" in run.stdout, run.stdout[-1000:] + run.stderr[-4000:]
    assert "pimemb:" not in run.stderr and "Sanitizer" not in run.stderr, run.stderr[-4000:]


def test_peer_group_mapping_watchdog_ends_the_process(tmp_path):
    """hipIpcOpenMemHandle has no deadline and has been seen never to return on this runtime (allocations above 2 GiB).  The
    stub makes it hang for good: the watchdog of emb_peer_create must end the PROCESS with a non-zero status and a message
    that names the rank, the peer and the chunk -- within the group's timeout, leaving no shared-memory segment behind."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "obj"
    build = subprocess.run(["bash", os.path.join(ROOT, "tests", "cpp", "build_host_logic_check.sh"), "address,undefined", str(out)],
                           capture_output=True, text=True, timeout=900)
    if build.returncode != 0 and "libclang_rt" in build.stderr and "No such file" in build.stderr:
        pytest.skip("sanitizer runtime not installed: " + build.stderr[-200:])
    assert build.returncode == 0, build.stderr[-3000:]
    before = set(os.listdir("/dev/shm")) if os.path.isdir("/dev/shm") else set()
    run = subprocess.run([str(out / "host_logic_check"), "ipc-hang"], capture_output=True, text=True, timeout=120)
    assert run.returncode == 70, (run.returncode, run.stderr[-2000:])
    assert "hipIpcOpenMemHandle of rank" in run.stderr and "chunk 0" in run.stderr and "has not returned within" in run.stderr
    assert "returned although" not in run.stderr
    left = {n for n in (set(os.listdir("/dev/shm")) - before if os.path.isdir("/dev/shm") else set()) if n.startswith("pimemb-hostcheck-hang")}
    assert not left, left
