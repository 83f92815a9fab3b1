// TEST INFRASTRUCTURE -- never linked into libpimemb.so, never shipped.
//
// A stand-in for the HIP runtime so the HOST side of the library (csrc/pimemb_engine.cpp, pimemb_shard.cpp,
// pimemb_compat.cpp and the host half of pimemb_kernels.hip, all compiled with `--cuda-host-only`) can run on a CPU-only
// box under ThreadSanitizer / AddressSanitizer (tests/test_host_side_sanitizers.py).  What it provides:
//   * "device" and pinned memory = calloc'd host memory (so ASan sees every overrun of a launch image or a staging buffer),
//     copies = memcpy, streams / events = small heap objects, everything synchronous;
//   * kernel launches are NO-OPS -- no lookup is computed here, nothing a test could mistake for a result -- except the
//     few SIGNALLING kernels whose words the host code waits for: store_word, publish_words, zero_words, validate (counts
//     out-of-range indices: its verdict steers host control flow), and the routers' COUNTS (they size the
//     sharded step's transfers; the request pieces themselves stay zero), the two mailbox kernels of the peer-store mode, and the served-bag
//     COUNTERS of a checked shard's counted ranged launch with the kernel that publishes them.  They run synchronously inside hipLaunchKernel,
//     found by the name the compiler registers for them.
// What the host-logic check asserts is therefore only return codes, tickets, ordering and that the sanitizers stay silent.
#include <hip/hip_runtime.h>
#include <sanitizer/common_interface_defs.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <unistd.h>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "pimemb_internal.h"
#include "pimemb_peer.h"

namespace {

std::mutex &reg_mu() { static std::mutex m; return m; }
std::map<const void *, std::string> &registry() { static std::map<const void *, std::string> r; return r; }

struct CallCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<CallCfg> t_cfg;

struct FakeStream { int device; };
struct FakeEvent { int recorded; int device; };

// ---- devices: the stub plays a node with several GPUs ----------------------------------------------------------------------
// The current device is per THREAD and starts at 0, as in the real runtime -- a helper thread the library starts must select
// its engine's device itself.  pimemb_stub_expect_device(d) arms a check: from then on every HIP call (but the three that
// query / select devices and the error getters) must be made with device d current, on streams and events created on device d;
// a call that is not counts as a violation and is named on stderr.  An engine / queue / shard / peer group created on device 1
// by a thread whose current device stays 0 must come through its whole life without one (tests/cpp/host_logic_check.cpp).
std::atomic<int> g_device_count{1}, g_expect_device{-1};
std::atomic<long> g_violations{0}, g_tracked_calls{0};
thread_local int t_device = 0;

void track(const char *what, int object_device = -1) {
    const int want = g_expect_device.load(std::memory_order_relaxed);
    if (want < 0) return;
    if (object_device == -2) object_device = -1;           // (a stale stream handle: refused by the call itself)
    g_tracked_calls.fetch_add(1, std::memory_order_relaxed);
    if (t_device != want || (object_device >= 0 && object_device != want)) {
        if (g_violations.fetch_add(1) < 20 && getenv("PIMEMB_STUB_STACKS")) __sanitizer_print_stack_trace();     // (where from?)
        if (g_violations.load() <= 20)
            fprintf(stderr, "[hip stub] %s: device %d current%s, the engine under test lives on device %d\n", what, t_device,
                    object_device >= 0 && object_device != want ? " (and the stream / event belongs to another device)" : "", want);
    }
}
// live streams (a handle the caller destroyed is refused like the real runtime refuses it -- never dereferenced)
std::mutex &stream_mu() { static std::mutex m; return m; }
std::map<hipStream_t, int> &streams() { static std::map<hipStream_t, int> m; return m; }
int stream_device(hipStream_t s) {
    if (!s) return -1;
    std::lock_guard<std::mutex> lk(stream_mu());
    auto it = streams().find(s);
    return it == streams().end() ? -2 : it->second;
}

template <typename T> T arg(void **args, int i) { return *static_cast<T *>(args[i]); }

struct PublishSrcMirror { const uint32_t *p[3]; uint32_t n[3]; };     // pimemb_kernels.hip: PublishSrc

template <typename IdxT>
void emulate_validate(void **args, dim3 grid) {
    using pimemb::DevDesc;
    (void)grid;
    DevDesc *descs = arg<DevDesc *>(args, 0);
    const uint32_t n_descs = arg<uint32_t>(args, 1);
    // (args[2], the call's ValidateCtl: the kernel's workgroups meet there; here the whole grid is this one function)
    unsigned long long *result = const_cast<unsigned long long *>(arg<volatile unsigned long long *>(args, 3));
    const unsigned long long seq = arg<unsigned long long>(args, 4);
    const int poison = arg<int>(args, 5);
    unsigned long long bad = 0;
    for (uint32_t d = 0; d < n_descs; d++) {
        const DevDesc &dd = descs[d];
        const IdxT *idx = static_cast<const IdxT *>(dd.indices), *off = static_cast<const IdxT *>(dd.offsets);
        for (uint64_t i = 0; i < dd.n_idx; i++)
            if (idx[i] < 0 || (uint64_t)idx[i] >= dd.nr_rows) bad++;
        if (off) {
            for (uint64_t b = 0; b < dd.n_bags; b++) {
                const uint64_t nxt = (b + 1 < dd.n_bags) ? (uint64_t)off[b + 1] : dd.n_idx;
                if (off[b] < 0 || (uint64_t)off[b] > nxt || nxt > dd.n_idx) bad++;
            }
        } else if ((uint64_t)dd.fixed_pooling * dd.n_bags != dd.n_idx) {
            bad++;
        }
    }
    if (bad && poison)
        for (uint32_t d = 0; d < n_descs; d++) descs[d].n_tiles = 0;
    __atomic_store_n(result, pimemb::validate_word(seq, bad), __ATOMIC_RELEASE);
}

// The routers' COUNTS (nothing else of their output: request pieces, slots and the rest of `meta` stay zero -- no lookup is
// computed here): how many sub-bags and indices of every row-split table this rank asks every shard for, and the two peaks.
// They size every transfer of the sharded step, so with them the host's offset arithmetic runs with real, ragged numbers
// on every rank (tests/cpp/host_logic_check.cpp: ranks as threads; a receive insists on the byte count its peer sent).
struct RouteBagTableMirror { const void *indices, *offsets; uint64_t n_indices; uint32_t fixed_pooling, rows_per_shard; };   // pimemb_kernels.hip
struct RouteBagParamsMirror { RouteBagTableMirror t[pimemb::kRouteBagMaxTables]; uint32_t idx64; };

void emulate_route_counts(const RouteBagParamsMirror &rp, uint64_t n_bags, uint32_t N, uint32_t K, uint32_t *meta) {
    auto pad4 = [](uint32_t v) { return (v + 3u) & ~3u; };
    std::vector<uint32_t> words(N, 0), rows(N, 0);
    for (uint32_t k = 0; k < K; k++) {
        const RouteBagTableMirror &t = rp.t[k];
        auto word = [&](const void *a, uint64_t i) -> uint64_t {        // (route_word: uint32 or int64 arrays, a negative id = huge)
            return rp.idx64 ? (uint64_t)static_cast<const int64_t *>(a)[i] : (uint64_t)static_cast<const uint32_t *>(a)[i];
        };
        std::vector<uint32_t> n_sub(N, 0), n_idx(N, 0);
        for (uint64_t b = 0; b < n_bags; b++) {
            uint64_t p = t.offsets ? word(t.offsets, b) : b * t.fixed_pooling;
            uint64_t e = t.offsets ? (b + 1 < n_bags ? word(t.offsets, b + 1) : t.n_indices) : p + t.fixed_pooling;
            if (e > t.n_indices) e = t.n_indices;
            if (p > e) p = e;
            std::vector<uint32_t> here(N, 0);
            for (; p < e; p++) {
                uint64_t d = word(t.indices, p) / t.rows_per_shard;
                if (d >= N) d = N - 1;                         // the last shard takes anything beyond
                here[d]++;
            }
            for (uint32_t d = 0; d < N; d++) {
                n_idx[d] += here[d];
                n_sub[d] += here[d] ? 1u : 0u;
            }
        }
        for (uint32_t d = 0; d < N; d++) {
            meta[(d * (K + 1) + k) * 2 + 0] = n_sub[d];
            meta[(d * (K + 1) + k) * 2 + 1] = n_idx[d];
            words[d] += pad4(n_sub[d]) + pad4(n_idx[d]);
            rows[d] += n_sub[d];
        }
    }
    // ... and where destination d's piece starts (words) and where shard d's partial rows of table k will sit (rows):
    // meta = counts[N][K+1][2] | base[N][K][2] | piece[N+1] | ret_row0[N][K] | mode   (pimemb_kernels.hip: meta_layout)
    const uint32_t base = 2 * N * (K + 1), piece = base + 2 * N * K, row0 = piece + N + 1;
    uint32_t w_at = 0, r_at = 0;
    for (uint32_t d = 0; d < N; d++) {
        meta[piece + d] = w_at;
        w_at += words[d];
        uint32_t r = r_at;
        for (uint32_t k = 0; k < K; k++) {
            meta[row0 + d * K + k] = r;
            r += meta[(d * (K + 1) + k) * 2 + 0];
        }
        r_at += rows[d];
    }
    meta[piece + N] = w_at;
    uint32_t peak_w = 0, peak_r = 0;
    for (uint32_t d = 0; d < N; d++) {
        if (words[d] > peak_w) peak_w = words[d];
        if (rows[d] > peak_r) peak_r = rows[d];
    }
    for (uint32_t d = 0; d < N; d++) {
        meta[(d * (K + 1) + K) * 2 + 0] = peak_w;
        meta[(d * (K + 1) + K) * 2 + 1] = peak_r;
    }
}

// The COUNTED ranged lookup of a checked shard's direct path: no row is gathered here, but the per-descriptor counters of
// served bags steer host control flow (the requester compares their sum with its bag count), so they are produced: the
// number of indices inside [row_lo, row_lo + nr_rows), added to the counter the descriptor names in pad_[1].
void emulate_ranged_counts(void **args, dim3 grid, bool idx64) {
    using pimemb::DevDesc;
    const DevDesc *descs = arg<const DevDesc *>(args, 0);
    const uint32_t chunks_arg = arg<uint32_t>(args, 1);
    const uint32_t *xmap = arg<const uint32_t *>(args, 2);
    std::vector<uint32_t> ids;
    if (!xmap) {
        for (uint32_t d = 0; d < grid.y; d++) ids.push_back(d);
    } else if (chunks_arg & pimemb::kXmapDirect) {
        for (uint32_t b = 0; b < grid.x; b++)
            if (xmap[2 * b] != 0xffffffffu) ids.push_back(xmap[2 * b]);
    } else {
        uint32_t n_seg = 0;
        for (uint32_t c = 0; c < 8; c++) n_seg += xmap[c];
        for (uint32_t i = 0; i < n_seg; i++) ids.push_back(xmap[16 + 4 * i]);
    }
    std::sort(ids.begin(), ids.end());
    ids.erase(std::unique(ids.begin(), ids.end()), ids.end());
    for (uint32_t d : ids) {
        const DevDesc &dd = descs[d];
        uint32_t *ctr = reinterpret_cast<uint32_t *>(dd.pad_[1]);
        if (!ctr || dd.n_tiles == 0) continue;
        const uint64_t row_lo = dd.pad_[0] & ~pimemb::kRangeOpenEnd;      // (the open end's bags are zeroed, never counted)
        uint32_t n = 0;
        for (uint64_t b = 0; b < dd.n_bags; b++) {
            const uint64_t id = idx64 ? (uint64_t)static_cast<const int64_t *>(dd.indices)[b] : (uint64_t)static_cast<const uint32_t *>(dd.indices)[b];
            if (id - row_lo < dd.nr_rows) n++;
        }
        __atomic_fetch_add(ctr + (size_t)(d % EMB_SERVED_LANES) * (EMB_SERVED_STRIDE / 4), n, __ATOMIC_RELAXED);      // (any lane of the counter)
    }
}

void emulate(const std::string &name, void **args, dim3 grid) {
    if (name.find("bag_sum_wavebatch_kernel") != std::string::npos && name.find("EELb1EEEvPK") != std::string::npos) {
        emulate_ranged_counts(args, grid, name.find("bag_sum_wavebatch_kernelIl") != std::string::npos);
    } else if (name.find("served_counts_kernel") != std::string::npos) {
        const pimemb::ServedArgs a = arg<pimemb::ServedArgs>(args, 0);
        for (uint32_t j = 0; j < a.n_flag; j++) __atomic_store_n(reinterpret_cast<unsigned long long *>(a.flag[j]), a.value[j], __ATOMIC_RELEASE);
        for (uint32_t i = 0; i < a.n_seg; i++) {
            for (uint32_t j = 0; j < a.seg[i].n_counted + a.seg[i].n_fill; j++) {     // an entry = (tag << 32) | count; a counter = EMB_SERVED_LANES words
                uint32_t sum = 0xffffffffu;
                if (j < a.seg[i].n_counted) {
                    sum = 0;
                    for (uint32_t l = 0; l < EMB_SERVED_LANES; l++)
                        sum += __atomic_exchange_n(a.seg[i].ctr + (size_t)j * (EMB_SERVED_BYTES / 4) + (size_t)l * (EMB_SERVED_STRIDE / 4), 0u, __ATOMIC_RELAXED);
                }
                __atomic_store_n(a.seg[i].dst + j, ((unsigned long long)a.seg[i].tag << 32) | sum, __ATOMIC_RELEASE);
            }
        }
    } else if (name.find("store_word_kernel") != std::string::npos) {
        unsigned long long *w = const_cast<unsigned long long *>(arg<volatile unsigned long long *>(args, 0));
        __atomic_store_n(w, arg<unsigned long long>(args, 1), __ATOMIC_RELEASE);
    } else if (name.find("zero_words_kernel") != std::string::npos) {
        memset(arg<uint32_t *>(args, 0), 0, (size_t)arg<uint32_t>(args, 1) * 4);
    } else if (name.find("publish_words_kernel") != std::string::npos) {
        const PublishSrcMirror src = arg<PublishSrcMirror>(args, 0);
        uint32_t *dst = arg<uint32_t *>(args, 1);
        uint32_t at = 0;
        for (int j = 0; j < 3; j++) {
            if (src.p[j]) memcpy(dst + at, src.p[j], (size_t)src.n[j] * 4);
            at += src.n[j];
        }
        unsigned long long *flag = const_cast<unsigned long long *>(arg<volatile unsigned long long *>(args, 2));
        __atomic_store_n(flag, arg<unsigned long long>(args, 3), __ATOMIC_RELEASE);
    } else if (name.find("route_bags_place_kernel") != std::string::npos) {          // (rp, n_bags, n_shards, n_tables, work, slots, meta, send)
        emulate_route_counts(arg<RouteBagParamsMirror>(args, 0), arg<uint64_t>(args, 1), arg<uint32_t>(args, 2), arg<uint32_t>(args, 3),
                             arg<uint32_t *>(args, 6));
    } else if (name.find("route_onehot_place_fused_kernel") != std::string::npos || name.find("route_onehot_place_kernel") != std::string::npos) {
        // (rp, n_bags, n_shards, n_tables, n_blocks, slots, work, meta, send)
        emulate_route_counts(arg<RouteBagParamsMirror>(args, 0), arg<uint64_t>(args, 1), arg<uint32_t>(args, 2), arg<uint32_t>(args, 3),
                             arg<uint32_t *>(args, 7));
    } else if (name.find("peer_post_kernel") != std::string::npos) {          // (meta, n_row_tables, n_shards, PeerPostArgs, value): the mailbox message
        const uint32_t *meta = arg<const uint32_t *>(args, 0);
        const uint32_t Kr = arg<uint32_t>(args, 1), N = arg<uint32_t>(args, 2);
        const pimemb::PeerPostArgs a = arg<pimemb::PeerPostArgs>(args, 3);
        const unsigned long long value = arg<unsigned long long>(args, 4);
        const uint32_t K1 = Kr ? Kr : 1, piece = 2 * N * (K1 + 1) + 2 * N * K1, row0 = piece + N + 1;
        for (uint32_t p = 0; p < N; p++) {
            if (a.box[p] == 0) continue;
            pimemb::PeerMsg *box = reinterpret_cast<pimemb::PeerMsg *>(a.box[p]);
            uint32_t at = 0;
            if (Kr) {
                const uint32_t n = 2 * (Kr + 1);
                for (uint32_t i = 0; i < n; i++) box->words[i] = meta ? meta[(p * (Kr + 1)) * 2 + i] : 0u;
                box->words[n] = meta ? meta[piece + p] : 0u;
                box->words[n + 1] = meta ? meta[row0 + p * Kr] : 0u;
                at = n + 2;
            }
            for (uint32_t i = 0; i < a.n_consts[p]; i++) box->words[at + i] = a.consts[p][i];
        }
        for (uint32_t p = 0; p < N; p++)
            if (a.box[p] != 0)
                __atomic_store_n(const_cast<unsigned long long *>(&reinterpret_cast<pimemb::PeerMsg *>(a.box[p])->posted), value, __ATOMIC_RELEASE);
    } else if (name.find("peer_done_kernel") != std::string::npos) {          // (PeerDoneArgs, value)
        const pimemb::PeerDoneArgs a = arg<pimemb::PeerDoneArgs>(args, 0);
        for (uint32_t i = 0; i < a.n; i++)
            __atomic_store_n(const_cast<unsigned long long *>(&reinterpret_cast<pimemb::PeerMsg *>(a.box[i])->served), arg<unsigned long long>(args, 1), __ATOMIC_RELEASE);
    } else if (name.find("validate_kernelIjE") != std::string::npos) {
        emulate_validate<uint32_t>(args, grid);
    } else if (name.find("validate_kernelIlE") != std::string::npos) {
        emulate_validate<int64_t>(args, grid);
    }
    // every other kernel (the lookups, the router, the un-router, the column scatter, the peer mailbox kernels): nothing
}

}  // namespace

extern "C" {

unsigned char pimemb_stub_fatbin[16] = {0};      // what the kernels object's __hip_fatbin_<hash> is pointed at when linking

void **__hipRegisterFatBinary(const void *) {
    static void *handle = nullptr;
    return &handle;
}
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *host_fn, char *, const char *device_name, unsigned int, void *, void *, void *,
                           void *, int *) {
    std::lock_guard<std::mutex> lk(reg_mu());
    registry()[host_fn] = device_name ? device_name : "";
}
void __hipRegisterVar(void **, void *, char *, const char *, int, size_t, int, int) {}

hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
    t_cfg.push_back(CallCfg{grid, block, shmem, stream});
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *stream) {
    if (t_cfg.empty()) return hipErrorInvalidValue;
    const CallCfg c = t_cfg.back();
    t_cfg.pop_back();
    *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
    return hipSuccess;
}

hipError_t hipLaunchKernel(const void *fn, dim3 grid, dim3 block, void **args, size_t, hipStream_t stream) {
    track("hipLaunchKernel", stream_device(stream));
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x == 0 || block.x * block.y * block.z > 1024u) return hipErrorInvalidConfiguration;
    std::string name;
    {
        std::lock_guard<std::mutex> lk(reg_mu());
        auto it = registry().find(fn);
        if (it == registry().end()) return hipErrorInvalidDeviceFunction;
        name = it->second;
    }
    emulate(name, args, grid);
    return hipSuccess;
}

hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
hipError_t hipGetDeviceCount(int *n) { *n = g_device_count.load(); return hipSuccess; }
hipError_t hipGetDevice(int *d) { *d = t_device; return hipSuccess; }
hipError_t hipSetDevice(int d) {
    if (d < 0 || d >= g_device_count.load()) return hipErrorInvalidDevice;
    t_device = d;
    return hipSuccess;
}
std::atomic<long> g_device_syncs{0};
long pimemb_stub_device_syncs(void) { return g_device_syncs.load(); }
hipError_t hipDeviceSynchronize(void) { track("hipDeviceSynchronize"); g_device_syncs.fetch_add(1); return hipSuccess; }

hipError_t hipMalloc(void **p, size_t n) {
    track("hipMalloc");
    *p = calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipFree(void *p) { track("hipFree"); free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) {
    track("hipHostMalloc");
    *p = calloc(n ? n : 1, 1);
    return *p ? hipSuccess : hipErrorOutOfMemory;
}
hipError_t hipHostFree(void *p) { track("hipHostFree"); free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { track("hipMemcpy"); if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t st) { track("hipMemcpyAsync", stream_device(st)); if (n) memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { track("hipMemset"); if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { track("hipMemsetAsync", stream_device(st)); if (n) memset(d, v, n); return hipSuccess; }

hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) {
    track("hipStreamCreateWithFlags");
    *s = reinterpret_cast<hipStream_t>(new FakeStream{t_device});
    std::lock_guard<std::mutex> lk(stream_mu());
    streams()[*s] = t_device;
    return hipSuccess;
}
hipError_t hipStreamDestroy(hipStream_t s) {
    const int dev = stream_device(s);
    if (dev == -2) return hipErrorInvalidHandle;
    track("hipStreamDestroy", dev);
    {
        std::lock_guard<std::mutex> lk(stream_mu());
        streams().erase(s);
    }
    delete reinterpret_cast<FakeStream *>(s);
    return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { track("hipStreamSynchronize", stream_device(s)); return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t s) { track("hipStreamQuery", stream_device(s)); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t, unsigned int) { track("hipStreamWaitEvent", stream_device(s)); return hipSuccess; }

hipError_t hipEventCreate(hipEvent_t *e) { track("hipEventCreate"); *e = reinterpret_cast<hipEvent_t>(new FakeEvent{0, t_device}); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { track("hipEventDestroy", reinterpret_cast<FakeEvent *>(e)->device); delete reinterpret_cast<FakeEvent *>(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) {
    const int sdev = stream_device(s);
    if (sdev == -2) return hipErrorInvalidHandle;          // a destroyed stream
    track("hipEventRecord", reinterpret_cast<FakeEvent *>(e)->device);
    // (the real runtime refuses an event of one device on a stream of another)
    if (s && sdev != reinterpret_cast<FakeEvent *>(e)->device) return hipErrorInvalidHandle;
    __atomic_store_n(&reinterpret_cast<FakeEvent *>(e)->recorded, 1, __ATOMIC_RELAXED);
    return hipSuccess;
}
hipError_t hipEventSynchronize(hipEvent_t e) { track("hipEventSynchronize", reinterpret_cast<FakeEvent *>(e)->device); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t e) { track("hipEventQuery", reinterpret_cast<FakeEvent *>(e)->device); return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { track("hipEventElapsedTime"); *ms = 0.001f; return hipSuccess; }

// ---- test controls (tests/cpp/host_logic_check.cpp) ----
void pimemb_stub_set_device_count(int n) { g_device_count.store(n); }
void pimemb_stub_expect_device(int d) { g_expect_device.store(d); }
long pimemb_stub_violations(void) { return g_violations.load(); }
long pimemb_stub_tracked_calls(void) { return g_tracked_calls.load(); }

}  // extern "C"

// ---- what csrc/pimemb_peer.cpp needs for a group of ONE process (an IPC handle is the pointer itself) ------------------------
extern "C" {
// two behaviours of real runtimes the peer group must survive (pimemb_peer.cpp): no fine-grained device memory, and an IPC
// mapping call that never returns
std::atomic<int> g_no_finegrained{0}, g_ipc_open_hangs{0};
void pimemb_stub_no_finegrained(int on) { g_no_finegrained.store(on); }
void pimemb_stub_ipc_open_hangs(int on) { g_ipc_open_hangs.store(on); }
hipError_t hipExtMallocWithFlags(void **p, size_t n, unsigned int) {
    if (g_no_finegrained.load()) { track("hipExtMallocWithFlags"); return hipErrorNotSupported; }
    return hipMalloc(p, n);
}
hipError_t hipHostRegister(void *, size_t, unsigned int) { track("hipHostRegister"); return hipSuccess; }
hipError_t hipHostUnregister(void *) { track("hipHostUnregister"); return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **dev, void *host, unsigned int) { track("hipHostGetDevicePointer"); *dev = host; return hipSuccess; }
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) {
    track("hipIpcGetMemHandle");
    memset(h, 0, sizeof(*h));
    memcpy(h, &p, sizeof(p));
    return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void **p, hipIpcMemHandle_t h, unsigned int) {
    track("hipIpcOpenMemHandle");
    while (g_ipc_open_hangs.load()) usleep(1000);       // "never returns" (until the process ends)
    memcpy(p, &h, sizeof(*p));
    return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void *) { track("hipIpcCloseMemHandle"); return hipSuccess; }
}
