// pimemb_peer.cpp -- a group of ranks (one process per GPU) that can read and write each other's HBM directly: the
// substrate of the sharded lookup's collective-free exchange (EMB_SHARD_PEER_STORES, pimemb_shard.cpp).
//
// xGMI is point to point and a lookup descriptor is just pointers: with every rank's index and output buffers mapped into
// its peers (HIP IPC), the rank that owns a table gathers a requester's indices in place and stores the pooled rows straight
// into the requester's HBM -- the reference's result pull lands directly in the caller's final_results too
// (upmem/include/emb_host.h:312-321).  No collectives library is involved, and none is needed to set the group up:
//
//   * one POSIX shared-memory segment per job (`/pimemb-<tag>`) holds, per rank, the IPC handle of its ARENA -- one device
//     allocation everything a peer may touch is carved from -- and a grid of small MAILBOXES [dst][src][slot] for the
//     per-batch handshake (what a rank asks a peer for, where its buffers are, "your rows are in place");
//   * every rank maps the segment into its GPU's address space (hipHostRegister), so mailbox words are written by tiny kernels
//     in stream order behind the data they announce, and read by the host with plain loads (no event, no copy engine).
//
// STATUS: exercised with 2-4 processes sharing the development box's one GPU (peer mappings of the same device); never
// over xGMI.  Visibility of peer stores across real links rests on the arena being fine-grained memory and on kernel
// boundaries; link rates are unmeasured.
#include <signal.h>
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "pimemb_peer.h"

namespace {

using pimemb::fail;

double now_s() {
    using namespace std::chrono;
    return duration<double>(steady_clock::now().time_since_epoch()).count();
}

constexpr uint64_t kMagic = 0x70696d656d623034ull;     // "pimemb04"

// PIMEMB_PEER_TRACE=1: a line per set-up step on stderr (which call is a group stuck in?)
void trace(int rank, const char *what) {
    static const bool on = getenv("PIMEMB_PEER_TRACE") != nullptr;
    if (on) {
        fprintf(stderr, "[pimemb peer %d] %.3f %s\n", rank, now_s(), what);
        fflush(stderr);
    }
}

// The arena is a set of CHUNKS, each a device allocation of its own with its own IPC handle: hipIpcOpenMemHandle of an
// allocation above 2 GiB never returns on this runtime (ROCm 7.0 / 7.2, same-device peers; tools/peer_arena_probe.py:
// 1.9 GiB maps in milliseconds, 2.1 GiB hangs, fine- or coarse-grained alike).  An arena OFFSET is chunk * kChunkBytes +
// the byte inside the chunk; an allocation never straddles two chunks.
constexpr uint64_t kChunkShift = 30, kChunkBytes = 1ull << kChunkShift;
constexpr uint32_t kMaxChunks = 192;   // 192 GiB of peer-visible memory per rank at most

struct RankInfo {
    std::atomic<uint64_t> ready;       // 1: handles and sizes below are valid; 2: this rank has mapped every peer
    std::atomic<uint64_t> barrier;     // generation counter of emb_peer_barrier
    uint64_t arena_bytes;
    int32_t device, pid;
    uint32_t n_chunks, pad0;
    hipIpcMemHandle_t handle[kMaxChunks];
};

struct ShmHeader {
    std::atomic<uint64_t> magic;
    uint32_t world, msg_bytes, slots, pad;
};

size_t header_bytes(int world) { return 4096 + (size_t)world * sizeof(RankInfo); }

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() {
        if (switched) (void)hipSetDevice(prev);
    }
};

// hipIpcOpenMemHandle has no deadline of its own, and on this runtime it has been seen never to return (an allocation above
// 2 GiB: see the chunk note above; what it does with a different threshold across real devices is unknown).  A rank stuck in
// it cannot report an error -- the call never comes back -- so a watchdog thread ENDS THE PROCESS with a message naming the
// rank, the peer and the chunk once a mapping call has been out for longer than the group's timeout: the launcher sees a
// non-zero exit instead of a job that hangs until its own limit.  (A fresh failure: nothing is re-executed.)
struct MapWatchdog {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false;
    int rank = 0, peer = -1;
    unsigned chunk = 0;
    double call_t0 = 0.0;       // 0: no mapping call is out
    double limit_s;
    std::string unlink_name;    // the job's segment (whoever ends the job takes the name with it: nothing is left behind in /dev/shm)
    std::thread th;
    MapWatchdog(int rank_, double limit, const std::string &unlink) : rank(rank_), limit_s(limit), unlink_name(unlink) {
        th = std::thread([this] {
            std::unique_lock<std::mutex> lk(mu);
            while (!done) {
                cv.wait_for(lk, std::chrono::milliseconds(50));
                if (!done && call_t0 > 0.0 && now_s() - call_t0 > limit_s) {
                    fprintf(stderr, "emb_peer_create: rank %d: hipIpcOpenMemHandle of rank %d's arena (chunk %u) has not returned within %.0f s -- "
                            "the runtime cannot map that allocation (seen for IPC mappings above 2 GiB on ROCm 7.0 / 7.2); this process "
                            "is ended so the job fails instead of hanging (PIMEMB_SHARD_TIMEOUT_S sets the limit)\n", rank, peer, chunk, limit_s);
                    fflush(stderr);
                    if (!unlink_name.empty()) (void)shm_unlink(unlink_name.c_str());
                    // PIMEMB_PEER_WATCHDOG=report: a caller with a deadline of its own (bench.py's peer-store leg) ends the
                    // process itself -- with the numbers it already has
                    const char *mode = getenv("PIMEMB_PEER_WATCHDOG");
                    if (mode && mode[0] == 'r') return;
                    _exit(70);
                }
            }
        });
    }
    void enter(int p, unsigned c) {
        std::lock_guard<std::mutex> lk(mu);
        peer = p;
        chunk = c;
        call_t0 = now_s();
    }
    void leave() {
        std::lock_guard<std::mutex> lk(mu);
        call_t0 = 0.0;
    }
    ~MapWatchdog() {
        {
            std::lock_guard<std::mutex> lk(mu);
            done = true;
        }
        cv.notify_all();
        th.join();
    }
};

}  // namespace

struct emb_peer {
    emb_engine *e = nullptr;
    int device = 0, rank = 0, world = 1;
    std::string shm_name;
    int fd = -1;
    char *shm = nullptr;           // host mapping of the segment
    char *shm_dev = nullptr;       // the same bytes as the GPU addresses them
    size_t shm_bytes = 0;
    bool registered = false, creator = false, unlinked = false;
    std::vector<char *> chunks;    // this rank's arena
    uint64_t arena_bytes = 0, arena_used = 0;
    bool fine_grained = false;
    std::vector<std::vector<char *>> base;   // base[p][c]: chunk c of rank p's arena as THIS process addresses it
    std::vector<uint64_t> peer_bytes;
    uint64_t barrier_gen = 0, epochs = 0;
    double timeout_s = 60.0;
};

namespace pimemb {

ShmHeader *peer_header(emb_peer *p) { return reinterpret_cast<ShmHeader *>(p->shm); }
RankInfo *peer_rank_info(emb_peer *p, int r) { return reinterpret_cast<RankInfo *>(p->shm + 4096) + r; }

PeerMsg *peer_box(emb_peer *p, int dst, int src, uint32_t slot) {
    const size_t at = header_bytes(p->world) + (((size_t)dst * p->world + src) * kPeerSlots + slot) * kPeerMsgBytes;
    return reinterpret_cast<PeerMsg *>(p->shm + at);
}
PeerMsg *peer_box_dev(emb_peer *p, int dst, int src, uint32_t slot) {
    const size_t at = header_bytes(p->world) + (((size_t)dst * p->world + src) * kPeerSlots + slot) * kPeerMsgBytes;
    return reinterpret_cast<PeerMsg *>(p->shm_dev + at);
}
int peer_rank(const emb_peer *p) { return p->rank; }
int peer_world(const emb_peer *p) { return p->world; }
uint64_t peer_arena_bytes(const emb_peer *p, int r) { return p->peer_bytes[(size_t)r]; }
// `bytes` at arena offset `off` of rank r, as this process addresses them; nullptr when the range leaves the arena or
// straddles two chunks (no allocation does)
char *peer_ptr(const emb_peer *p, int r, uint64_t off, uint64_t bytes) {
    const uint64_t c = off >> kChunkShift, last = (off + (bytes ? bytes - 1 : 0)) >> kChunkShift;
    if (c != last || off + bytes > p->peer_bytes[(size_t)r] || c >= p->base[(size_t)r].size()) return nullptr;
    return p->base[(size_t)r][c] + (off & (kChunkBytes - 1));
}
// arena offset of a pointer into THIS rank's arena (~0 when it is not inside one chunk of it)
uint64_t peer_offset(const emb_peer *p, const void *ptr, uint64_t bytes) {
    const char *c = static_cast<const char *>(ptr);
    for (size_t k = 0; k < p->chunks.size(); k++) {
        const uint64_t len = std::min<uint64_t>(kChunkBytes, p->arena_bytes - k * kChunkBytes);
        if (c >= p->chunks[k] && c + bytes <= p->chunks[k] + len) return k * kChunkBytes + (uint64_t)(c - p->chunks[k]);
    }
    return ~0ull;
}
double peer_timeout_s(const emb_peer *p) { return p->timeout_s; }
uint64_t peer_next_epoch(emb_peer *p) { return ++p->epochs; }
bool peer_owns(const emb_peer *p, const void *ptr, uint64_t bytes) { return peer_offset(p, ptr, bytes) != ~0ull; }

}  // namespace pimemb

extern "C" {

int emb_peer_create(emb_engine *e, const char *job_tag, int32_t rank, int32_t world, uint64_t arena_bytes, emb_peer **out) {
    if (!e || !job_tag || !out) return fail(EMB_ERR_INVALID, "emb_peer_create: NULL argument");
    *out = nullptr;
    if (world < 1 || world > 64 || rank < 0 || rank >= world) return fail(EMB_ERR_INVALID, "emb_peer_create: rank %d of %d (1..64 ranks)", rank, world);
    if (arena_bytes < (1u << 20)) arena_bytes = 1u << 20;
    emb_peer *p = new (std::nothrow) emb_peer();
    if (!p) return fail(EMB_ERR_NOMEM, "out of host memory");
    p->e = e;
    p->rank = rank;
    p->world = world;
    if (const char *t = getenv("PIMEMB_SHARD_TIMEOUT_S")) p->timeout_s = atof(t) > 0 ? atof(t) : p->timeout_s;
    int32_t dev = 0;
    int rc = emb_device_of(e, &dev);
    if (rc) {
        delete p;
        return rc;
    }
    p->device = dev;
    DeviceGuard g(dev);
    auto bail = [&](int code) {
        (void)emb_peer_destroy(p);
        return code;
    };
    // ---- the shared segment: whoever comes first creates and sizes it (ftruncate zero-fills), everyone maps it
    p->shm_name = std::string("/pimemb-") + job_tag;
    p->shm_bytes = header_bytes(world) + (size_t)world * world * pimemb::kPeerSlots * pimemb::kPeerMsgBytes;
    p->shm_bytes = (p->shm_bytes + 4095) / 4096 * 4096;
    p->fd = shm_open(p->shm_name.c_str(), O_CREAT | O_EXCL | O_RDWR, 0600);
    if (p->fd >= 0) {
        p->creator = true;
        if (ftruncate(p->fd, (off_t)p->shm_bytes) != 0) return bail(fail(EMB_ERR_NOMEM, "emb_peer_create: ftruncate(%s, %zu) failed", p->shm_name.c_str(), p->shm_bytes));
    } else {
        const double t0 = now_s();
        struct stat st {};
        for (;;) {       // the creator may not have sized it yet
            if (p->fd < 0) p->fd = shm_open(p->shm_name.c_str(), O_RDWR, 0600);
            if (p->fd >= 0 && fstat(p->fd, &st) == 0 && (size_t)st.st_size >= p->shm_bytes) break;
            if (now_s() - t0 > p->timeout_s) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: %s did not appear within %.0f s", p->shm_name.c_str(), p->timeout_s));
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
    }
    void *m = mmap(nullptr, p->shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, p->fd, 0);
    if (m == MAP_FAILED) return bail(fail(EMB_ERR_NOMEM, "emb_peer_create: mmap of %s failed", p->shm_name.c_str()));
    p->shm = static_cast<char *>(m);
    ShmHeader *h = pimemb::peer_header(p);
    if (p->creator) {
        h->world = (uint32_t)world;
        h->msg_bytes = pimemb::kPeerMsgBytes;
        h->slots = pimemb::kPeerSlots;
        h->magic.store(kMagic, std::memory_order_release);
    } else {
        const double t0 = now_s();
        while (h->magic.load(std::memory_order_acquire) != kMagic) {
            if (now_s() - t0 > p->timeout_s) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: %s was never initialised (a stale segment of another job?)", p->shm_name.c_str()));
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if ((int)h->world != world || h->msg_bytes != pimemb::kPeerMsgBytes)
            return bail(fail(EMB_ERR_INVALID, "emb_peer_create: %s belongs to a group of %u ranks (this one: %d) -- use a fresh job tag", p->shm_name.c_str(), h->world, world));
    }
    trace(rank, "segment mapped; hipHostRegister");
    // the GPU writes mailbox words: map the segment into the device's address space
    hipError_t err = hipHostRegister(p->shm, p->shm_bytes, hipHostRegisterMapped);
    if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipHostRegister of the shared segment: %s", hipGetErrorString(err)));
    p->registered = true;
    void *dptr = nullptr;
    err = hipHostGetDevicePointer(&dptr, p->shm, 0);
    if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipHostGetDevicePointer: %s", hipGetErrorString(err)));
    p->shm_dev = static_cast<char *>(dptr);

    // ---- the arena: everything a peer may read or write lives here, in chunks of at most 1 GiB (see RankInfo).  Fine-grained
    // by default (stores of one agent are visible to another without waiting for a cache write-back);
    // PIMEMB_PEER_ARENA=coarse takes ordinary device memory
    const char *mode = getenv("PIMEMB_PEER_ARENA");
    const uint32_t n_chunks = (uint32_t)((arena_bytes + kChunkBytes - 1) >> kChunkShift);
    if (n_chunks > kMaxChunks) return bail(fail(EMB_ERR_UNSUPPORTED, "emb_peer_create: an arena of %llu bytes needs %u chunks (limit %u)", (unsigned long long)arena_bytes, n_chunks, kMaxChunks));
    RankInfo *me = pimemb::peer_rank_info(p, rank);
    p->fine_grained = !(mode && mode[0] == 'c');
    for (int attempt = 0; attempt < 2 && p->chunks.empty(); attempt++) {
        bool ok = true;
        for (uint32_t c = 0; c < n_chunks && ok; c++) {
            const uint64_t len = std::min<uint64_t>(kChunkBytes, arena_bytes - (uint64_t)c * kChunkBytes);
            void *a = nullptr;
            err = p->fine_grained ? hipExtMallocWithFlags(&a, len, hipDeviceMallocFinegrained) : hipMalloc(&a, len);
            if (err == hipSuccess) {
                p->chunks.push_back(static_cast<char *>(a));
                err = hipIpcGetMemHandle(&me->handle[c], a);
            }
            ok = err == hipSuccess;
        }
        if (!ok) {
            (void)hipGetLastError();
            for (char *c : p->chunks) (void)hipFree(c);
            p->chunks.clear();
            if (!p->fine_grained)
                return bail(fail(err == hipErrorOutOfMemory ? EMB_ERR_NOMEM : EMB_ERR_DEVICE, "emb_peer_create: %llu bytes of arena: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)",
                                 (unsigned long long)arena_bytes, hipGetErrorString(err)));
            // A runtime that cannot allocate / export fine-grained memory.  Ordinary (coarse-grained) memory is a stand-in only
            // where no second agent stores into it: with peers, rows another GPU stored into this rank's arena are not
            // guaranteed visible to this GPU's kernels at a kernel boundary -- so the fallback is taken silently for a group of
            // ONE rank only; with peers it must be asked for (PIMEMB_PEER_ARENA=coarse: same-device rehearsals), else refused.
            if (world > 1)
                return bail(fail(EMB_ERR_UNSUPPORTED, "emb_peer_create: this runtime cannot allocate / export fine-grained device memory (%s); a peer group of %d ranks "
                                 "will not fall back to ordinary memory by itself -- stores of another GPU into it are not guaranteed visible. Use the RCCL "
                                 "exchange, or set PIMEMB_PEER_ARENA=coarse to take ordinary memory knowingly (ranks sharing one device)",
                                 hipGetErrorString(err), world));
            p->fine_grained = false;
        }
    }
    p->arena_bytes = arena_bytes;
    trace(rank, p->fine_grained ? "arena allocated (fine-grained), handles taken" : "arena allocated (ordinary), handles taken");
    me->arena_bytes = arena_bytes;
    me->n_chunks = n_chunks;
    me->device = dev;
    me->pid = (int32_t)getpid();
    me->ready.store(1, std::memory_order_release);

    // ---- map every peer's arena
    p->base.assign((size_t)world, {});
    p->peer_bytes.assign((size_t)world, 0);
    p->base[(size_t)rank] = p->chunks;
    p->peer_bytes[(size_t)rank] = arena_bytes;
    const double t0 = now_s();
    std::unique_ptr<MapWatchdog> dog(world > 1 ? new MapWatchdog(rank, p->timeout_s, p->shm_name) : nullptr);
    for (int r = 0; r < world; r++) {
        if (r == rank) continue;
        RankInfo *ri = pimemb::peer_rank_info(p, r);
        while (ri->ready.load(std::memory_order_acquire) < 1) {
            if (now_s() - t0 > p->timeout_s) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: rank %d did not join %s within %.0f s", r, p->shm_name.c_str(), p->timeout_s));
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        for (uint32_t c = 0; c < ri->n_chunks; c++) {
            void *mapped = nullptr;
            dog->enter(r, c);
            err = hipIpcOpenMemHandle(&mapped, ri->handle[c], hipIpcMemLazyEnablePeerAccess);
            dog->leave();
            if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipIpcOpenMemHandle of rank %d's arena (chunk %u): %s", r, c, hipGetErrorString(err)));
            p->base[(size_t)r].push_back(static_cast<char *>(mapped));
        }
        p->peer_bytes[(size_t)r] = ri->arena_bytes;
    }
    trace(rank, "peers mapped; barrier");
    me->ready.store(2, std::memory_order_release);
    *out = p;
    rc = emb_peer_barrier(p);       // nobody proceeds (or tears the segment down) before everyone has mapped everyone
    if (rc) {
        *out = nullptr;
        return bail(rc);
    }
    if (p->creator) {       // every rank holds its mapping: the name can go (nothing is left behind)
        (void)shm_unlink(p->shm_name.c_str());
        p->unlinked = true;
    }
    return EMB_OK;
}

int emb_peer_alloc(emb_peer *p, uint64_t bytes, void **ptr) {
    if (!p || !ptr) return fail(EMB_ERR_INVALID, "emb_peer_alloc: NULL argument");
    uint64_t at = (p->arena_used + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    if (bytes > kChunkBytes)
        return fail(EMB_ERR_UNSUPPORTED, "emb_peer_alloc: one allocation of %llu bytes -- the arena is made of 1-GiB chunks (IPC mappings above 2 GiB "
                    "hang on this runtime): allocate per table", (unsigned long long)bytes);
    if ((at >> kChunkShift) != ((at + bytes - 1) >> kChunkShift)) at = ((at >> kChunkShift) + 1) << kChunkShift;      // never across two chunks
    if (at + bytes > p->arena_bytes)
        return fail(EMB_ERR_NOMEM, "emb_peer_alloc: %llu bytes do not fit the arena (%llu of %llu used): create the group with a larger one",
                    (unsigned long long)bytes, (unsigned long long)p->arena_used, (unsigned long long)p->arena_bytes);
    *ptr = p->chunks[(size_t)(at >> kChunkShift)] + (at & (kChunkBytes - 1));
    p->arena_used = at + bytes;
    return EMB_OK;
}

int emb_peer_info(emb_peer *p, int32_t *rank, int32_t *world, void **arena, uint64_t *arena_bytes, uint64_t *used, int32_t *fine_grained) {
    if (!p) return fail(EMB_ERR_INVALID, "emb_peer_info: group is NULL");
    if (rank) *rank = p->rank;
    if (world) *world = p->world;
    if (arena) *arena = p->chunks.empty() ? nullptr : p->chunks[0];
    if (arena_bytes) *arena_bytes = p->arena_bytes;
    if (used) *used = p->arena_used;
    if (fine_grained) *fine_grained = p->fine_grained ? 1 : 0;
    return EMB_OK;
}

int emb_peer_barrier(emb_peer *p) {
    if (!p) return fail(EMB_ERR_INVALID, "emb_peer_barrier: group is NULL");
    const uint64_t gen = ++p->barrier_gen;
    pimemb::peer_rank_info(p, p->rank)->barrier.store(gen, std::memory_order_release);
    const double t0 = now_s();
    for (int r = 0; r < p->world; r++)
        while (pimemb::peer_rank_info(p, r)->barrier.load(std::memory_order_acquire) < gen) {
            if (now_s() - t0 > p->timeout_s) return fail(EMB_ERR_DEVICE, "emb_peer_barrier: rank %d did not arrive within %.0f s", r, p->timeout_s);
            std::this_thread::yield();
        }
    return EMB_OK;
}

int emb_peer_destroy(emb_peer *p) {
    if (!p) return EMB_OK;
    DeviceGuard g(p->device);
    (void)hipDeviceSynchronize();
    for (int r = 0; r < (int)p->base.size(); r++)
        if (r != p->rank)
            for (char *c : p->base[(size_t)r]) (void)hipIpcCloseMemHandle(c);
    for (char *c : p->chunks) (void)hipFree(c);
    if (p->registered) (void)hipHostUnregister(p->shm);
    if (p->shm) (void)munmap(p->shm, p->shm_bytes);
    if (p->fd >= 0) (void)close(p->fd);
    if (p->creator && !p->unlinked && !p->shm_name.empty()) (void)shm_unlink(p->shm_name.c_str());   // a set-up that failed half-way leaves no segment behind
    (void)hipGetLastError();
    delete p;
    return EMB_OK;
}

// ---- last words (see pimemb.h) -----------------------------------------------------------------------------------------
namespace {
char *g_last_words = nullptr;                  // malloc'd, never freed while the handlers are in (a handler may be reading it)
size_t g_last_words_len = 0;
int g_last_words_fd = 1, g_last_words_status = 0;
bool g_last_words_in = false;
struct sigaction g_last_words_old[4];
// (SIGTERM: what a launcher sends the surviving ranks when ANOTHER rank died -- the line belongs on stdout all the same)
const int kLastWordSignals[4] = {SIGABRT, SIGSEGV, SIGBUS, SIGTERM};

void last_words_handler(int) {
    size_t off = 0;
    while (off < g_last_words_len) {
        const ssize_t n = write(g_last_words_fd, g_last_words + off, g_last_words_len - off);
        if (n <= 0) break;
        off += (size_t)n;
    }
    _exit(g_last_words_status);
}
}  // namespace

int emb_peer_last_words(const char *line, int fd, int status) {
    static std::mutex mu;
    std::lock_guard<std::mutex> lk(mu);
    if (g_last_words_in) {                     // take the handlers out first: nothing reads the old text after this
        for (int i = 0; i < 4; i++) (void)sigaction(kLastWordSignals[i], &g_last_words_old[i], nullptr);
        g_last_words_in = false;
    }
    free(g_last_words);
    g_last_words = nullptr;
    g_last_words_len = 0;
    if (!line) return EMB_OK;
    const size_t n = strlen(line);
    g_last_words = static_cast<char *>(malloc(n + 2));
    if (!g_last_words) return fail(EMB_ERR_NOMEM, "emb_peer_last_words: %zu bytes", n + 2);
    memcpy(g_last_words, line, n);
    g_last_words_len = n;
    g_last_words_fd = fd;
    g_last_words_status = status;
    if (n && line[n - 1] != '\n') g_last_words[g_last_words_len++] = '\n';
    struct sigaction sa;
    memset(&sa, 0, sizeof sa);
    sa.sa_handler = last_words_handler;
    sigemptyset(&sa.sa_mask);
    for (int i = 0; i < 4; i++)
        if (sigaction(kLastWordSignals[i], &sa, &g_last_words_old[i]) != 0) return fail(EMB_ERR_DEVICE, "emb_peer_last_words: sigaction failed");
    g_last_words_in = true;
    return EMB_OK;
}

}  // extern "C"
