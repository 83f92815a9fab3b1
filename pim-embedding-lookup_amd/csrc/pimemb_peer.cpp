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
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
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

struct RankInfo {
    std::atomic<uint64_t> ready;       // 1: handle and sizes below are valid; 2: this rank has mapped every peer
    std::atomic<uint64_t> barrier;     // generation counter of emb_peer_barrier
    uint64_t arena_bytes;
    int32_t device, pid;
    hipIpcMemHandle_t handle;
    char pad[128 - 32 - sizeof(hipIpcMemHandle_t) % 128];
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

}  // namespace

struct emb_peer {
    emb_engine *e = nullptr;
    int device = 0, rank = 0, world = 1;
    std::string shm_name;
    int fd = -1;
    char *shm = nullptr;           // host mapping of the segment
    char *shm_dev = nullptr;       // the same bytes as the GPU addresses them
    size_t shm_bytes = 0;
    bool registered = false, creator = false;
    char *arena = nullptr;
    uint64_t arena_bytes = 0, arena_used = 0;
    bool fine_grained = false;
    std::vector<char *> base;      // base[p]: rank p's arena as THIS process addresses it (base[rank] == arena)
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
char *peer_base(const emb_peer *p, int r) { return p->base[(size_t)r]; }
uint64_t peer_arena_bytes(const emb_peer *p, int r) { return p->peer_bytes[(size_t)r]; }
double peer_timeout_s(const emb_peer *p) { return p->timeout_s; }
uint64_t peer_next_epoch(emb_peer *p) { return ++p->epochs; }
bool peer_owns(const emb_peer *p, const void *ptr, uint64_t bytes) {
    const char *c = static_cast<const char *>(ptr);
    return c >= p->arena && c + bytes <= p->arena + p->arena_bytes;
}

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
    // the GPU writes mailbox words: map the segment into the device's address space
    hipError_t err = hipHostRegister(p->shm, p->shm_bytes, hipHostRegisterMapped);
    if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipHostRegister of the shared segment: %s", hipGetErrorString(err)));
    p->registered = true;
    void *dptr = nullptr;
    err = hipHostGetDevicePointer(&dptr, p->shm, 0);
    if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipHostGetDevicePointer: %s", hipGetErrorString(err)));
    p->shm_dev = static_cast<char *>(dptr);

    // ---- the arena: everything a peer may read or write lives here.  Fine-grained by default (stores of one agent are
    // visible to another without waiting for a cache write-back); PIMEMB_PEER_ARENA=coarse takes ordinary device memory
    const char *mode = getenv("PIMEMB_PEER_ARENA");
    void *a = nullptr;
    if (!(mode && mode[0] == 'c')) {
        err = hipExtMallocWithFlags(&a, arena_bytes, hipDeviceMallocFinegrained);
        if (err == hipSuccess) p->fine_grained = true;
        else (void)hipGetLastError();
    }
    if (!a) {
        err = hipMalloc(&a, arena_bytes);
        if (err != hipSuccess) return bail(fail(EMB_ERR_NOMEM, "emb_peer_create: %llu bytes of arena: %s", (unsigned long long)arena_bytes, hipGetErrorString(err)));
    }
    p->arena = static_cast<char *>(a);
    p->arena_bytes = arena_bytes;
    RankInfo *me = pimemb::peer_rank_info(p, rank);
    err = hipIpcGetMemHandle(&me->handle, p->arena);
    if (err != hipSuccess && p->fine_grained) {      // (a runtime that cannot export fine-grained memory: ordinary memory instead)
        (void)hipGetLastError();
        (void)hipFree(p->arena);
        p->arena = nullptr;
        p->fine_grained = false;
        err = hipMalloc(&a, arena_bytes);
        if (err == hipSuccess) {
            p->arena = static_cast<char *>(a);
            err = hipIpcGetMemHandle(&me->handle, p->arena);
        }
    }
    if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipIpcGetMemHandle: %s (HSA_ENABLE_IPC_MODE_LEGACY=0 set?)", hipGetErrorString(err)));
    me->arena_bytes = arena_bytes;
    me->device = dev;
    me->pid = (int32_t)getpid();
    me->ready.store(1, std::memory_order_release);

    // ---- map every peer's arena
    p->base.assign((size_t)world, nullptr);
    p->peer_bytes.assign((size_t)world, 0);
    p->base[(size_t)rank] = p->arena;
    p->peer_bytes[(size_t)rank] = arena_bytes;
    const double t0 = now_s();
    for (int r = 0; r < world; r++) {
        if (r == rank) continue;
        RankInfo *ri = pimemb::peer_rank_info(p, r);
        while (ri->ready.load(std::memory_order_acquire) < 1) {
            if (now_s() - t0 > p->timeout_s) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: rank %d did not join %s within %.0f s", r, p->shm_name.c_str(), p->timeout_s));
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        void *mapped = nullptr;
        err = hipIpcOpenMemHandle(&mapped, ri->handle, hipIpcMemLazyEnablePeerAccess);
        if (err != hipSuccess) return bail(fail(EMB_ERR_DEVICE, "emb_peer_create: hipIpcOpenMemHandle of rank %d's arena: %s", r, hipGetErrorString(err)));
        p->base[(size_t)r] = static_cast<char *>(mapped);
        p->peer_bytes[(size_t)r] = ri->arena_bytes;
    }
    me->ready.store(2, std::memory_order_release);
    *out = p;
    rc = emb_peer_barrier(p);       // nobody proceeds (or tears the segment down) before everyone has mapped everyone
    if (rc) {
        *out = nullptr;
        return bail(rc);
    }
    if (p->creator) (void)shm_unlink(p->shm_name.c_str());      // every rank holds its mapping: the name can go (nothing is left behind)
    return EMB_OK;
}

int emb_peer_alloc(emb_peer *p, uint64_t bytes, void **ptr) {
    if (!p || !ptr) return fail(EMB_ERR_INVALID, "emb_peer_alloc: NULL argument");
    const uint64_t at = (p->arena_used + 255) / 256 * 256;
    if (bytes == 0) bytes = 256;
    if (at + bytes > p->arena_bytes)
        return fail(EMB_ERR_NOMEM, "emb_peer_alloc: %llu bytes do not fit the arena (%llu of %llu used): create the group with a larger one",
                    (unsigned long long)bytes, (unsigned long long)p->arena_used, (unsigned long long)p->arena_bytes);
    *ptr = p->arena + at;
    p->arena_used = at + bytes;
    return EMB_OK;
}

int emb_peer_info(emb_peer *p, int32_t *rank, int32_t *world, void **arena, uint64_t *arena_bytes, uint64_t *used, int32_t *fine_grained) {
    if (!p) return fail(EMB_ERR_INVALID, "emb_peer_info: group is NULL");
    if (rank) *rank = p->rank;
    if (world) *world = p->world;
    if (arena) *arena = p->arena;
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
        if (r != p->rank && p->base[(size_t)r]) (void)hipIpcCloseMemHandle(p->base[(size_t)r]);
    if (p->arena) (void)hipFree(p->arena);
    if (p->registered) (void)hipHostUnregister(p->shm);
    if (p->shm) (void)munmap(p->shm, p->shm_bytes);
    if (p->fd >= 0) (void)close(p->fd);
    (void)hipGetLastError();
    delete p;
    return EMB_OK;
}

}  // extern "C"
