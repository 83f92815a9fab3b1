// pimemb_comm.cpp -- optional native exchange step for the sharded lookup: an all-to-all of byte
// ranges issued straight to RCCL (grouped ncclSend / ncclRecv) on the caller's HIP stream.
//
// Why: torch.distributed.all_to_all_single costs 25-45 us of CPU per call (tools/a2a_probe.py) and its
// completion is a separate work object; the sharded lookup step is host-bound on that.  Issued from
// here the exchange is ordinary stream work between two kernel launches.
//
// RCCL is NOT a link-time dependency: the symbols are looked up at run time, first in the process
// (a torch process has torch's bundled librccl.so mapped -- one RCCL per process, like the one HIP
// runtime), then from librccl.so on the loader path.  The reference has no counterpart: its "exchange"
// is dpu_push_xfer to every DPU (emb_host.h:258-287) and a pull of the results (:321).
//
// STATUS: exercised with one rank (self send/recv) and with three and four ranks sharing the development box's one GPU
// (every rank claiming a host of its own, so RCCL connects them through its socket transport:
// tests/test_gpu_sharding.py::test_rccl_several_ranks_on_one_gpu[native-*], all tables verified on every rank); never
// over xGMI.  It is the only transport of the RCCL-mode sharded step (emb_shard_* -> emb_comm_exchange); bench.py's N > 1
// legs call emb_shard_*, torch.distributed carries the bootstrap, the barriers and the job clock only.
#include <dlfcn.h>

#include <cstdio>
#include <cstring>
#include <mutex>
#include <vector>

#include "pimemb_internal.h"

namespace {

using pimemb::fail;

typedef int ncclResult_t;              // ncclSuccess == 0
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;   // NCCL_UNIQUE_ID_BYTES (rccl.h:40)
constexpr int kNcclUint8 = 1;          // ncclDataType_t ncclUint8 (rccl.h:460)

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_rccl_once;

void load_rccl() {
    void *h = dlopen("librccl.so", RTLD_NOW | RTLD_NOLOAD);        // already mapped by torch?
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) return;
#define SYM(field, name) g_rccl.field = reinterpret_cast<decltype(g_rccl.field)>(dlsym(h, name))
    SYM(GetUniqueId, "ncclGetUniqueId");
    SYM(CommInitRank, "ncclCommInitRank");
    SYM(CommDestroy, "ncclCommDestroy");
    SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd");
    SYM(Send, "ncclSend");
    SYM(Recv, "ncclRecv");
    SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.GroupStart &&
                g_rccl.GroupEnd && g_rccl.Send && g_rccl.Recv;
}

const Rccl *rccl() {
    std::call_once(g_rccl_once, load_rccl);
    return g_rccl.ok ? &g_rccl : nullptr;
}

int nccl_fail(const char *what, ncclResult_t r) {
    return fail(EMB_ERR_DEVICE, "%s failed: %s", what,
                g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
}

}  // namespace

struct emb_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 0, device = 0;
};

extern "C" {

int emb_comm_unique_id(void *id128) {
    if (!id128) return fail(EMB_ERR_INVALID, "emb_comm_unique_id: id128 is NULL");
    const Rccl *r = rccl();
    if (!r) return fail(EMB_ERR_UNSUPPORTED, "librccl.so is not available in this process");
    ncclUniqueId id;
    ncclResult_t rc = r->GetUniqueId(&id);
    if (rc) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id128, id.internal, sizeof id.internal);
    return EMB_OK;
}

int emb_comm_create(emb_engine *e, const void *id128, int32_t rank, int32_t world, emb_comm **out) {
    if (!e || !id128 || !out) return fail(EMB_ERR_INVALID, "emb_comm_create: NULL argument");
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) return fail(EMB_ERR_INVALID, "emb_comm_create: rank %d of %d", rank, world);
    const Rccl *r = rccl();
    if (!r) return fail(EMB_ERR_UNSUPPORTED, "librccl.so is not available in this process");
    int32_t dev = 0;
    int rc0 = emb_device_of(e, &dev);
    if (rc0) return rc0;
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != dev && hipSetDevice(dev) != hipSuccess) return fail(EMB_ERR_DEVICE, "emb_comm_create: hipSetDevice(%d)", dev);
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof id.internal);
    emb_comm *c = new (std::nothrow) emb_comm();
    if (!c) return fail(EMB_ERR_NOMEM, "out of host memory");
    ncclResult_t rc = r->CommInitRank(&c->comm, world, id, rank);
    if (prev >= 0 && prev != dev) (void)hipSetDevice(prev);
    if (rc) {
        delete c;
        return nccl_fail("ncclCommInitRank", rc);
    }
    c->rank = rank;
    c->world = world;
    c->device = dev;
    *out = c;
    return EMB_OK;
}

// One send / receive moves at most kChunk bytes: RCCL 2.26.6 (the one torch 2.10 bundles) delivers only the first half
// of a transfer above 1 GiB (tools/a2a_size_probe.py: intact up to 1.0 GiB, corrupt from 1.1 GiB, any element type).
// Both ends of a pair know the pair's size, so both cut it into the same pieces; pieces to one peer match in order.
constexpr uint64_t kChunk = 512ull << 20;

int emb_comm_exchange(emb_comm *c, const emb_comm_op *ops, uint32_t n_ops, void *stream) {
    if (!c || (n_ops && !ops)) return fail(EMB_ERR_INVALID, "emb_comm_exchange: NULL argument");
    for (uint32_t i = 0; i < n_ops; i++) {
        if (ops[i].peer < 0 || ops[i].peer >= c->world) return fail(EMB_ERR_INVALID, "emb_comm_exchange: op %u: peer %d of %d", i, ops[i].peer, c->world);
        if (ops[i].bytes && !ops[i].ptr) return fail(EMB_ERR_INVALID, "emb_comm_exchange: op %u: NULL buffer", i);
    }
    const Rccl *r = rccl();
    hipStream_t s = static_cast<hipStream_t>(stream);
    int prev = -1;
    (void)hipGetDevice(&prev);
    if (prev != c->device) (void)hipSetDevice(c->device);
    ncclResult_t rc = r->GroupStart();
    for (uint32_t i = 0; rc == 0 && i < n_ops; i++) {
        const emb_comm_op &op = ops[i];
        for (uint64_t o = 0; rc == 0 && o < op.bytes; o += kChunk) {
            const uint64_t n = op.bytes - o < kChunk ? op.bytes - o : kChunk;
            rc = op.is_recv ? r->Recv(static_cast<char *>(op.ptr) + o, n, kNcclUint8, op.peer, c->comm, s)
                            : r->Send(static_cast<const char *>(op.ptr) + o, n, kNcclUint8, op.peer, c->comm, s);
        }
    }
    ncclResult_t rc2 = r->GroupEnd();
    if (prev >= 0 && prev != c->device) (void)hipSetDevice(prev);
    if (rc) return nccl_fail("ncclSend/ncclRecv", rc);
    if (rc2) return nccl_fail("ncclGroupEnd", rc2);
    return EMB_OK;
}

int emb_comm_rank(const emb_comm *c, int32_t *rank, int32_t *world) {
    if (!c) return fail(EMB_ERR_INVALID, "emb_comm_rank: comm is NULL");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    return EMB_OK;
}

int emb_comm_all_to_all(emb_comm *c, const void *send, const uint64_t *send_off, void *recv,
                        const uint64_t *recv_off, void *stream) {
    if (!c || !send || !recv || !send_off || !recv_off) return fail(EMB_ERR_INVALID, "emb_comm_all_to_all: NULL argument");
    std::vector<emb_comm_op> ops;
    ops.reserve(2 * (size_t)c->world);
    for (int p = 0; p < c->world; p++) {
        ops.push_back(emb_comm_op{p, 0, const_cast<char *>(static_cast<const char *>(send)) + send_off[p], send_off[p + 1] - send_off[p]});
        ops.push_back(emb_comm_op{p, 1, static_cast<char *>(recv) + recv_off[p], recv_off[p + 1] - recv_off[p]});
    }
    return emb_comm_exchange(c, ops.data(), (uint32_t)ops.size(), stream);
}

int emb_comm_destroy(emb_comm *c) {
    if (!c) return EMB_OK;
    const Rccl *r = rccl();
    if (r && c->comm) (void)r->CommDestroy(c->comm);
    delete c;
    return EMB_OK;
}

}  // extern "C"
