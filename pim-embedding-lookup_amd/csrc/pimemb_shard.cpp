// pimemb_shard.cpp -- the sharded lookup as ONE library call per batch (emb_shard_* in include/pimemb.h).
//
// Reference counterpart: lookup() serves every device from one call -- push the indices to all DPUs
// (upmem/include/emb_host.h:258-270), launch (:297), pull every result into the caller's final_results (:312-321).
// Here the devices are the GPUs of one node (one process each), tables are placed per SURVEY.md section 8 row E
// (replicated / whole on an owner / split by row range), and a batch goes through
//
//     R route + counts out, L local lookup   ->   Q counts in (host), requests travel   ->   S fused lookup over what
//     arrived, T pooled rows return, U partial rows added in shard order
//
// software-pipelined over consecutive batches (depth 0..3).  Everything between two kernels is an RCCL group issued from
// here (emb_comm_exchange) or nothing at all: pieces a rank addresses to itself are served in place, whole tables travel
// straight out of / into the caller's buffers.  The only host wait of a batch is for the counts, which a small kernel
// drops into pinned memory behind a flag word (polled; no event, no copy engine).
//
// Streams.  Every kernel (R, L, S, U) is enqueued on the CALLER's stream, every transfer on ONE internal stream; an event
// crosses between the two only where a transfer really happens (counts out, requests in, rows out, rows back: four
// records + four waits per batch with peers, none at all with one rank).  The first version ran R / S+L / U on three
// internal streams with an input event per batch: 16 event operations and ~60 us of host time per step on this runtime
// (an event record or wait costs ~3 us, the first kernel behind a cross-stream wait several times that).
//
// No lookup is computed here: S and L are emb_lookup_batched launches (pimemb_engine.cpp), R and U the routing kernels
// (pimemb_kernels.hip).  This file is streams, events, byte offsets and the order in which all ranks issue transfers.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "pimemb_internal.h"
#include "pimemb_peer.h"

namespace {

using pimemb::fail;

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t _e = (expr);                                                                         \
        if (_e != hipSuccess)                                                                           \
            return fail(_e == hipErrorOutOfMemory ? EMB_ERR_NOMEM : EMB_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, \
                        hipGetErrorString(_e), __FILE__, __LINE__);                                     \
    } while (0)
#define EMB_TRY(expr)              \
    do {                           \
        int _rc = (expr);          \
        if (_rc != EMB_OK) return _rc; \
    } while (0)

double now_us() {
    using namespace std::chrono;
    return duration<double, std::micro>(steady_clock::now().time_since_epoch()).count();
}

// PIMEMB_SHARD_PROFILE=1: where the host time of a step goes, printed by emb_shard_destroy (developer aid).
struct HostProf {
    bool on = getenv("PIMEMB_SHARD_PROFILE") != nullptr;
    double acc[16] = {};
    const char *name[16] = {"", "route kernels", "whole counts", "ev_routed", "counts exchange", "publish", "local lookup",
                            "wait counts", "sizes+ensure", "request exchange", "serve descs", "serve lookup", "return exchange",
                            "unroute", "", ""};
    double t = 0;
    void start() { if (on) t = now_us(); }
    void lap(int k) { if (on) { const double n = now_us(); acc[k] += n - t; t = n; } }
};
HostProf g_hp;

constexpr int kRing = 6;            // batches in flight at most: depth 3 keeps four live, one slot is being filled, one to spare
// words of a whole-table count entry: {bags, indices, fixed pooling (0 = offsets travel), 0} and -- for the collective-free
// exchange only -- where the requester's arrays sit in its arena: {indices, offsets, pooled} as 64-bit byte offsets, 2 spare
constexpr uint32_t kWholeWords = 12;
constexpr uint32_t kPeerConstHead = 8;   // host-written words in front of a destination's whole entries: req_send offset (2), ret_recv offset (2), bags, whole entries, 2 spare

inline uint64_t pad4(uint64_t v) { return (v + 3u) & ~(uint64_t)3u; }
// index / offset arrays are uint32 (the reference's width) or int64 (torch's): bytes per entry, and the 16-byte aligned uint32
// words an array of n entries takes in a staging buffer
inline uint32_t isz_of(uint32_t itype) { return itype == EMB_IDX_I64 ? 8u : 4u; }
inline uint64_t idx_words(uint64_t n, uint32_t itype) { return pad4(n * (isz_of(itype) / 4u)); }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    bool arena = false;      // carved from the peer group's arena (peers address it): never freed, outgrown pieces are abandoned
};

// what a peer told this rank about its side of a batch (collective-free exchange)
struct PeerFrom {
    uint64_t req_send_off = 0, ret_recv_off = 0;   // byte offsets inside the peer's arena
    uint32_t piece_word = 0, ret_row0 = 0;
    // the source handed its one-index-per-bag row-split tables over DIRECTLY: raw index array + output buffer per table
    bool direct = false;
    uint32_t itype = EMB_IDX_U32;                  // width of the source's raw index arrays (direct path)
    uint64_t n_bags = 0;
    std::vector<uint64_t> idx_off, out_off;        // per row-split table (peers: arena offsets; this rank itself: unused)
};

enum Via : int { SELF = 0, COMM = 1, PEER = 2 };

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

enum Stage : int { FREE = 0, ROUTED = 1, REQUESTED = 2, SERVED = 3, DONE = 4 };

struct Batch {
    uint64_t seq = ~0ull;
    Stage stage = FREE;
    uint64_t n_bags = 0;
    uint32_t itype = EMB_IDX_U32;       // width of this batch's index / offset arrays (emb_shard_input.index_type, every table alike)
    std::vector<emb_shard_input> in;
    // HBM, grow-only, owned by the slot
    DevBuf req_send, meta, slotmap, counts_in, wc_send, req_recv, ret_send, ret_recv;
    DevBuf wide;                        // an int64 job's routed step: the request pieces it serves, widened to int64 words (widen_pieces)
    // pinned host
    uint32_t *wc_host = nullptr;        // whole-table counts this rank sends: [sum_p |whole_of[p]|][kWholeWords]
    uint32_t *pc_host = nullptr;        // collective-free exchange: per destination, the constants its mailbox message ends with
    std::vector<PeerFrom> from;         // ... and what every source told this rank
    uint32_t *counts_host = nullptr;    // [sent row counts N(Kr+1)2 | received row counts N(Kr+1)2 | received whole counts N*M*4]
    unsigned long long *flag = nullptr; // raised (= seq + 1) behind counts_host by publish_words_kernel
    hipEvent_t ev_routed = nullptr, ev_req = nullptr, ev_served = nullptr, ev_ret = nullptr, ev_out = nullptr;
    bool req_recorded = false, ret_recorded = false, out_recorded = false, counts_posted = false;
    // per-peer sizes of this batch (row-split path), from the counts
    std::vector<uint64_t> out_words, in_words, rows_back, rows_served;
    int deferred_rc = EMB_OK;
    bool reported = false;              // deferred_rc has been handed to the caller (a batch's finding is returned ONCE)
    bool check_pending = false;         // EMB_SHARD_DEFER_REPORT: the served counts of this (completed) batch have not been compared yet
    bool direct = false;                // this batch's row-split tables skip router and un-router (see stage_route)
    // checked shards: what the counted ranged launches served (see stage_serve / stage_unroute)
    DevBuf chk_ctr;                     // HBM counters [N][Kr + M] (source p: its row-split tables, then the whole tables owned here) + [R] replicated
    unsigned long long *chk_host = nullptr;    // pinned, one self-describing word (tag << 32 | count) per entry: [R] replicated | [Kr + M] what this rank served ITSELF
    bool rep_counted = false, self_published = false;
    bool lazy_unpublished = false;      // deferred report: some of this batch's counters wait in emb_shard::lazy_segs for the next shared publish kernel
    // optional kernel timing (emb_shard_set_kernel_timing): start / stop around R, L, S, U
    hipEvent_t tev[10] = {};
    bool timed[5] = {false, false, false, false, false}, harvest = false;
};

}  // namespace

struct emb_shard {
    emb_engine *e = nullptr;
    emb_comm *comm = nullptr;
    emb_peer *peer = nullptr;
    bool peer_mode = false;
    uint64_t peer_tag = 0;                           // (epoch of this object among the group's users) << 40: added to every flag value
    int device = 0;
    int rank = 0, N = 1;
    uint32_t T = 0, dim = 0, depth = 2, flags = 0;
    std::vector<emb_shard_table> tabs;
    std::vector<uint32_t> rep, rows;                 // table ids: replicated / row-split (Kr = rows.size())
    std::vector<std::vector<uint32_t>> whole_of;     // per owner rank: its whole tables, in table order
    std::vector<uint32_t> wc_off;                    // first whole-count entry addressed to peer p (N + 1 entries)
    std::vector<uint32_t> elem_bytes;                // per table: element size of what this rank holds (0: nothing held)
    uint32_t Kr = 0, M = 0, Wtot = 0;                // row-split tables, whole tables owned here, whole tables in all
    bool self_via_comm = false, check_served = false;
    bool defer_report = false;                       // EMB_SHARD_DEFER_REPORT: requester-side findings surface at the next call
    hipStream_t s_comm = nullptr;                    // every transfer (RCCL group) of every batch, in one order on all ranks
    hipStream_t cs = nullptr;                        // the caller's stream: every kernel runs there
    bool cs_known = false;
    hipEvent_t ev_switch = nullptr;                  // the caller changed streams between two submits
    DevBuf work;                                     // scratch of one route call (ordered on the caller's stream)
    struct CachedPlan {
        std::vector<emb_lookup_desc> key;
        std::vector<uint64_t> key_lo;                // row ranges of a ranged launch (empty: an ordinary one)
        std::vector<uint32_t *> key_ctr;             // ... and its served counters (a checked shard's counted launch)
        uint32_t itype = EMB_IDX_U32;                // width of the index arrays its descriptors point at
        emb_plan *plan = nullptr;
        uint64_t last_use = 0, seen = 0;
    };
    // recurring lookups (same buffers, same lengths): one emb_plan_launch.  A pipelined loop over R rotating batch slots has R
    // steady-state launches (L(n) next to S(n - 2)) and as many again each time the pipeline fills and drains (L alone, S
    // alone): room for all of them, or the first call after a flush rebuilds -- and, evicting, waits for the device (seen:
    // 0.5 ms in the first submit of a 20-step timed region with 16 entries and 8 slots).
    static constexpr size_t kPlanCache = 64;
    std::vector<CachedPlan> plans;
    std::vector<emb_plan *> retired;                 // evicted plans: destroyed with the shard (emb_plan_destroy waits for the device)
    uint64_t plan_clock = 0;
    Batch ring[kRing];
    uint64_t next_seq = 0;
    double timeout_s = 60.0;
    emb_shard_stats st{};
    std::vector<emb_comm_op> ops;                    // scratch
    std::vector<emb_lookup_desc> descs;              // scratch
    std::vector<emb_lookup_desc> local;              // L(n) of the batch being submitted: launched together with the S of an older
                                                     // batch that is served in the same call (one launch instead of two), else alone
    Batch *local_of = nullptr;
    bool kernel_timing = false;
    bool allow_direct = true;                        // PIMEMB_SHARD_DIRECT=0 switches the direct one-hot path off (A/B, rehearsal)
    bool merge_direct = true;                        // PIMEMB_SHARD_DIRECT_MERGE=0: the direct path's lookup stays a launch of its own (A/B)
    std::vector<uint64_t> row_lo;                    // scratch
    std::vector<emb_lookup_desc> rdescs;             // scratch: the direct path's descriptors (one per source and row-split table)
    // checked shards count what their ranged launches serve: one counter per descriptor, parallel to descs / local / rdescs
    std::vector<uint32_t *> desc_ctr, local_ctr, rdesc_ctr;
    // ... and the width of each descriptor's index array (routed pieces are uint32 local row ids whatever the caller's width)
    std::vector<uint32_t> desc_it, local_it, rdesc_it;
    struct PieceRef { uint32_t desc, src; uint64_t off_word, idx_word; };     // a routed piece's descriptor: whose piece, where in it
    std::vector<PieceRef> piece_refs;
    bool widen = true;                                 // PIMEMB_SHARD_WIDEN=0: A/B (two launches, as before)
    // EMB_SHARD_DEFER_REPORT with no peer-store peer: the counters of a counted launch go to this rank's own pinned words only, and
    // nobody waits for them inside the call -- so they are not published by a kernel of their own behind every lookup (+4.7 us on a
    // 62-us step) but collected here and carried by ONE kernel every second submit (and before anything that waits for them)
    std::vector<pimemb::ServedSeg> lazy_segs;
    uint32_t lazy_age = 0;
    uint32_t refused = 0;                            // bit w: a validating launch over descriptors of index width w found something and
                                                     // gathered NOTHING (fused_lookup_one; its outputs were zeroed) -- the caller says whose batch that was
    uint32_t max_whole = 0;                          // most whole tables on one owner: the served counts' tail of a mailbox is Kr + max_whole words
    char range_msg[200] = {0};                       // what a requester-side count mismatch said (emb_last_error of the deferred EMB_ERR_RANGE)
};

namespace {

// Grow-only device buffer of a slot.  The slot's previous occupant went through all its stages at least two submits ago,
// but its kernels may still be queued: growing (rare: 25 % headroom) waits for the caller's stream and the transfer stream.
int ensure(emb_shard *s, DevBuf &buf, size_t bytes) {
    if (buf.cap >= bytes && buf.p) return EMB_OK;
    if (buf.arena) {             // peers hold addresses inside the arena: a larger piece is carved, the old one abandoned
        const size_t cap = 2 * bytes + 256;
        void *p = nullptr;
        EMB_TRY(emb_peer_alloc(s->peer, cap, &p));
        buf.p = p;
        buf.cap = cap;
        return EMB_OK;
    }
    if (s->cs_known) HIP_TRY(hipStreamSynchronize(s->cs));
    HIP_TRY(hipStreamSynchronize(s->s_comm));
    if (buf.p) HIP_TRY(hipFree(buf.p));
    buf.p = nullptr;
    buf.cap = 0;
    const size_t cap = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&buf.p, cap));
    buf.cap = cap;
    return EMB_OK;
}

// Kernel timing (off by default: an event between two kernels of a stream costs GPU time of its own).  which: 0 R, 1 L, 2 S, 3 U, 4 D (direct one-hot lookup).
int tick(emb_shard *s, Batch &b, int which, bool stop) {
    if (!s->kernel_timing) return EMB_OK;
    hipEvent_t &ev = b.tev[2 * which + (stop ? 1 : 0)];
    if (!ev) HIP_TRY(hipEventCreate(&ev));
    HIP_TRY(hipEventRecord(ev, s->cs));
    if (stop) b.timed[which] = b.harvest = true;
    return EMB_OK;
}

// Add a finished batch's kernel times to the stats (waits for the last of its brackets).
void harvest(emb_shard *s, Batch &b) {
    if (!b.harvest) return;
    b.harvest = false;
    double *dst[5] = {&s->st.us_kernel_route, &s->st.us_kernel_local, &s->st.us_kernel_serve, &s->st.us_kernel_unroute, &s->st.us_kernel_direct};
    for (int w = 0; w < 5; w++) {
        if (!b.timed[w]) continue;
        b.timed[w] = false;
        float ms = 0.f;
        if (hipEventSynchronize(b.tev[2 * w + 1]) == hipSuccess && hipEventElapsedTime(&ms, b.tev[2 * w], b.tev[2 * w + 1]) == hipSuccess)
            *dst[w] += ms * 1000.0;
    }
    s->st.n_timed_batches++;
    (void)hipGetLastError();
}

// how peer p's pieces travel: not at all (p is this rank: served in place), through RCCL, or by direct loads / stores
// into p's mapped arena
inline Via via(const emb_shard *s, int p) {
    if (p == s->rank && !s->self_via_comm) return SELF;
    return s->peer_mode ? PEER : COMM;
}
inline bool via_comm(const emb_shard *s, int p) { return via(s, p) == COMM; }
inline uint64_t arena_off(const emb_shard *s, const void *ptr) { return pimemb::peer_offset(s->peer, ptr, 1); }

// Poll a mailbox word until it carries `want` (written by a peer's GPU through the job's shared segment).
int poll_word(emb_shard *s, volatile unsigned long long *w, unsigned long long want, const char *what, int peer, uint64_t seq) {
    const double t0 = now_us(), limit = pimemb::peer_timeout_s(s->peer) * 1e6;
    for (uint64_t spin = 0; *w != want; spin++)
        if ((spin & 0xfff) == 0xfff && now_us() - t0 > limit)
            return fail(EMB_ERR_DEVICE, "emb_shard: rank %d %s batch %llu within %.0f s -- it is missing, or the ranks did not make the same calls",
                        peer, what, (unsigned long long)seq, limit / 1e6);
    return EMB_OK;
}

int exchange(emb_shard *s) {
    if (s->ops.empty()) return EMB_OK;
    if (!s->comm) return fail(EMB_ERR_INVALID, "emb_shard: a transfer to a peer without a communicator");
    for (const emb_comm_op &o : s->ops)
        if (!o.is_recv) (o.peer == s->rank ? s->st.bytes_to_self : s->st.bytes_to_peers) += o.bytes;
    return emb_comm_exchange(s->comm, s->ops.data(), (uint32_t)s->ops.size(), s->s_comm);
}

inline void add_op(emb_shard *s, int peer, bool recv, const void *ptr, uint64_t bytes) {
    if (bytes) s->ops.push_back(emb_comm_op{peer, recv ? 1 : 0, const_cast<void *>(ptr), bytes});
}

inline uint32_t *sent_counts(const emb_shard *s, const Batch &b) { (void)s; return b.counts_host; }
inline uint32_t *recv_counts(const emb_shard *s, const Batch &b) { return b.counts_host + (size_t)s->N * (s->Kr + 1) * 2; }
inline uint32_t *recv_whole(const emb_shard *s, const Batch &b) { return b.counts_host + (size_t)s->N * (s->Kr + 1) * 4; }

// checked shards: the served-bag counter of source p's i-th piece (i < Kr: row-split table i; Kr + j: the j-th whole table
// owned here), the counter of the r-th replicated table, and where the counts for a peer sit in its mailbox
// (a counter = EMB_SERVED_LANES words spread over EMB_SERVED_BYTES: pimemb.h, emb_lookup_ranged_counted)
constexpr size_t kCtrWords = EMB_SERVED_BYTES / 4;
inline size_t n_counters(const emb_shard *s) { return (size_t)s->N * (s->Kr + s->M) + s->rep.size(); }
inline uint32_t *ctr_of(const emb_shard *s, const Batch &b, uint32_t p, uint32_t i) { return static_cast<uint32_t *>(b.chk_ctr.p) + ((size_t)p * (s->Kr + s->M) + i) * kCtrWords; }
inline uint32_t *ctr_rep(const emb_shard *s, const Batch &b, uint32_t r) { return static_cast<uint32_t *>(b.chk_ctr.p) + ((size_t)s->N * (s->Kr + s->M) + r) * kCtrWords; }
// a served-count entry is one 64-bit word, (tag << 32) | count: in a peer's mailbox they take the last 2 x (Kr + max_whole)
// uint32 words (8-byte aligned: the message starts 64 bytes into the mailbox, kPeerMsgWords is even)
inline uint32_t served_tail(const emb_shard *s) { return pimemb::kPeerMsgWords - 2u * (s->Kr + s->max_whole); }
// the tag of batch seq's entries: distinct for every batch of this shard object (2^31 of them), and -- the group's n-th user of
// the mailboxes folds n in -- not what an earlier user left behind in the same slot; never 0 (what zeroed memory holds)
inline uint32_t served_tag(const emb_shard *s, uint64_t seq) { return 0x80000000u | ((uint32_t)(seq + 1) ^ ((uint32_t)(s->peer_tag >> 40) * 0x9E3779B1u)); }
inline bool all_one_hot(const std::vector<emb_lookup_desc> &v) {
    for (const emb_lookup_desc &d : v)
        if (d.offsets != nullptr || d.fixed_pooling != 1) return false;
    return true;
}

// One fused lookup over s->descs (all of index width `it`) on the caller's stream.  A call that recurs byte for byte (same
// tables, same buffers, same lengths: static batch slots, fixed-size whole-table pieces) is served by a prepared plan from its
// second sighting on -- one kernel enqueue, no descriptor resolution.  A plan holds addresses, never values.  Checked lookups
// never use plans.
// ranged: s->row_lo holds a row range start per descriptor (emb_lookup_ranged: one index per bag; a whole table is row_lo 0).
int fused_lookup_one(emb_shard *s, Batch &b, bool cacheable, bool ranged, uint32_t it) {
    const uint32_t n = (uint32_t)s->descs.size();
    if (n == 0) return EMB_OK;
    const emb_index_type itype = (emb_index_type)it;
    if (ranged && s->row_lo.size() != n) return fail(EMB_ERR_INVALID, "emb_shard: internal: %u descriptors, %zu row ranges", n, s->row_lo.size());
    const bool counted = ranged && s->check_served;      // a checked shard's ranged launch counts the bags it serves
    if (counted && s->desc_ctr.size() != n) return fail(EMB_ERR_INVALID, "emb_shard: internal: %u descriptors, %zu counters", n, s->desc_ctr.size());
    if (s->check_served && !ranged) {
        uint64_t bad = 0;
        int rc = emb_lookup_batched_checked(s->e, s->descs.data(), n, itype, EMB_MEM_DEVICE, s->cs, &bad);
        if (rc == EMB_ERR_RANGE) {          // nothing was gathered: the pieces pool to zero rows, the batch still completes
            for (const emb_lookup_desc &d : s->descs)
                HIP_TRY(hipMemsetAsync(d.pooled, 0, d.n_bags * (size_t)s->dim * 4, s->cs));
            s->refused |= 1u << it;
            return EMB_OK;
        }
        return rc;
    }
    if (cacheable) {
        s->plan_clock++;
        emb_shard::CachedPlan *hit = nullptr, *victim = nullptr;
        for (emb_shard::CachedPlan &c : s->plans) {
            if (c.key.size() == n && c.itype == it && c.key_lo.size() == (ranged ? n : 0u) && c.key_ctr.size() == (counted ? n : 0u) &&
                memcmp(c.key.data(), s->descs.data(), n * sizeof(emb_lookup_desc)) == 0 &&
                (!ranged || memcmp(c.key_lo.data(), s->row_lo.data(), n * 8) == 0) &&
                (!counted || memcmp(c.key_ctr.data(), s->desc_ctr.data(), n * sizeof(uint32_t *)) == 0))
                hit = &c;
            if (!victim || c.last_use < victim->last_use) victim = &c;
        }
        if (hit) {
            hit->last_use = s->plan_clock;
            if (hit->plan && emb_plan_launch(hit->plan, s->cs) == EMB_OK) return EMB_OK;
            if (hit->plan) {                 // stale (a table was reloaded): rebuild below
                (void)emb_plan_destroy(hit->plan);
                hit->plan = nullptr;
            }
            if (++hit->seen >= 2 && (ranged ? emb_plan_create_ranged_typed(s->e, s->descs.data(), s->row_lo.data(), counted ? s->desc_ctr.data() : nullptr, n, itype, &hit->plan)
                                            : emb_plan_create(s->e, s->descs.data(), n, itype, &hit->plan)) == EMB_OK)
                return emb_plan_launch(hit->plan, s->cs);
        } else {
            if (s->plans.size() < emb_shard::kPlanCache) {
                s->plans.emplace_back();
                victim = &s->plans.back();
            } else if (victim->plan) {       // destroying a plan waits for the device: not here, between two launches
                s->retired.push_back(victim->plan);
                victim->plan = nullptr;
                if (s->retired.size() > 4 * emb_shard::kPlanCache) {        // (a caller whose buffers never recur: pay the wait once in a long while)
                    for (emb_plan *p : s->retired) (void)emb_plan_destroy(p);
                    s->retired.clear();
                }
            }
            victim->key.assign(s->descs.begin(), s->descs.end());
            victim->key_lo.clear();
            if (ranged) victim->key_lo.assign(s->row_lo.begin(), s->row_lo.end());
            victim->key_ctr.clear();
            if (counted) victim->key_ctr.assign(s->desc_ctr.begin(), s->desc_ctr.end());
            victim->itype = it;
            victim->seen = 1;
            victim->last_use = s->plan_clock;
        }
    }
    if (ranged) return emb_lookup_ranged_typed(s->e, s->descs.data(), s->row_lo.data(), counted ? s->desc_ctr.data() : nullptr, n, itype, s->cs);
    return emb_lookup_batched(s->e, s->descs.data(), n, itype, EMB_MEM_DEVICE, s->cs);
}

// One fused lookup over s->descs, whose index arrays are of the widths s->desc_it names.  One width (every call of a uint32 job,
// every one-index call of an int64 job): ONE launch.  Two widths -- an int64 job's ROUTED step, whose request pieces are uint32
// local row ids next to the callers' own int64 arrays of replicated / whole tables: one launch per width (pieces change size with
// every batch, so nothing of such a call recurs: never cached).
int fused_lookup(emb_shard *s, Batch &b, bool cacheable, bool ranged = false) {
    const size_t n = s->descs.size();
    if (n == 0) return EMB_OK;
    if (s->desc_it.size() != n) return fail(EMB_ERR_INVALID, "emb_shard: internal: %zu descriptors, %zu index widths", n, s->desc_it.size());
    bool mixed = false;
    for (size_t i = 1; i < n; i++) mixed = mixed || s->desc_it[i] != s->desc_it[0];
    if (!mixed) return fused_lookup_one(s, b, cacheable, ranged, s->desc_it[0]);
    std::vector<emb_lookup_desc> all_d;
    std::vector<uint64_t> all_lo;
    std::vector<uint32_t *> all_ctr;
    std::vector<uint32_t> all_it;
    all_d.swap(s->descs);
    all_lo.swap(s->row_lo);
    all_ctr.swap(s->desc_ctr);
    all_it.swap(s->desc_it);
    int rc = EMB_OK;
    for (uint32_t it = EMB_IDX_U32; it <= EMB_IDX_I64 && rc == EMB_OK; it++) {
        s->descs.clear();
        s->row_lo.clear();
        s->desc_ctr.clear();
        for (size_t i = 0; i < n; i++) {
            if (all_it[i] != it) continue;
            s->descs.push_back(all_d[i]);
            if (all_lo.size() == n) s->row_lo.push_back(all_lo[i]);
            if (all_ctr.size() == n) s->desc_ctr.push_back(all_ctr[i]);
        }
        rc = fused_lookup_one(s, b, false, ranged, it);
    }
    s->descs.swap(all_d);
    s->row_lo.swap(all_lo);
    s->desc_ctr.swap(all_ctr);
    s->desc_it.swap(all_it);
    return rc;
}

int publish_lazy(emb_shard *s);

// ---- R(b) + counts out + L(b) -------------------------------------------------------------------------------------
int stage_route(emb_shard *s, Batch &b) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr, M = s->M;
    g_hp.start();
    b.req_recorded = b.ret_recorded = b.out_recorded = false;
    if (b.lazy_unpublished) EMB_TRY(publish_lazy(s));       // (never reached: submit has collected this slot's previous occupant; the counters must be zero before new adds)
    b.deferred_rc = EMB_OK;
    b.reported = b.check_pending = false;
    b.rep_counted = b.self_published = false;
    const uint32_t isz = isz_of(b.itype);

    // One index per bag and no peer behind RCCL (every peer is this rank itself or reachable by loads / stores): the
    // row-split tables need NO routing at all -- every shard scans the requester's raw index array and serves the bags whose
    // row it holds, straight into the requester's output (emb_lookup_ranged); each bag has exactly one shard, so there are
    // no counts to exchange, no partial rows to add.  Decided per batch and per rank; a peer learns it from the mailbox.
    b.direct = false;
    if (Kr && s->allow_direct) {
        bool ok = true;
        for (uint32_t p = 0; p < N && ok; p++) ok = via(s, (int)p) != COMM;
        for (uint32_t k = 0; k < Kr && ok; k++) {
            const emb_shard_input &u = b.in[s->rows[k]];
            ok = u.offsets == nullptr && u.fixed_pooling == 1;
            if (ok && s->peer_mode && N > 1 && b.n_bags)       // peers gather from / store into these in place
                ok = pimemb::peer_owns(s->peer, u.indices, b.n_bags * isz) && pimemb::peer_owns(s->peer, u.pooled, b.n_bags * (uint64_t)s->dim * 4);
        }
        b.direct = ok;
    }
    uint64_t total_idx = 0;
    for (uint32_t k = 0; k < Kr; k++) total_idx += b.in[s->rows[k]].n_indices;
    if (b.direct) {          // accounted on the requester's side: every bag is served exactly once, by one of the shards
        for (uint32_t k = 0; k < Kr; k++)
            s->st.served_algorithmic_bytes += b.n_bags * ((uint64_t)s->dim * s->elem_bytes[s->rows[k]] + isz + (uint64_t)s->dim * 4) +
                                              (uint64_t)(N - 1) * b.n_bags * isz;
        s->st.served_sub_bags += b.n_bags * Kr;
        s->st.served_indices += b.n_bags * Kr;
    }
    // row-split tables: cut every bag into per-shard sub-bags; the counts sit at the head of `meta`
    if (Kr && !b.direct) {
        uint64_t sb = 0, mb = 0, lb = 0, wb = 0;
        EMB_TRY(emb_route_bags_sizes(Kr, std::max<uint64_t>(b.n_bags, 1), total_idx, N, &sb, &mb, &lb, &wb));
        EMB_TRY(ensure(s, b.req_send, sb));
        if (s->peer_mode)        // peers store their partial rows straight into ret_recv: it must exist (and be known) before they do
            EMB_TRY(ensure(s, b.ret_recv, std::min<uint64_t>(total_idx, b.n_bags * (uint64_t)N * Kr) * s->dim * 4 + 256));
        EMB_TRY(ensure(s, b.meta, mb));
        EMB_TRY(ensure(s, b.slotmap, lb));
        EMB_TRY(ensure(s, s->work, wb));
        if (b.n_bags) {
            EMB_TRY(tick(s, b, 0, false));
            emb_route_table rt[pimemb::kRouteBagMaxTables];
            for (uint32_t k = 0; k < Kr; k++) {
                const emb_shard_input &u = b.in[s->rows[k]];
                rt[k] = emb_route_table{u.indices, u.offsets, u.n_indices, u.fixed_pooling, s->tabs[s->rows[k]].rows_per_shard};
            }
            EMB_TRY(emb_route_bags_typed(s->e, rt, Kr, (emb_index_type)b.itype, b.n_bags, N, b.req_send.p, static_cast<uint32_t *>(b.meta.p),
                                         static_cast<uint32_t *>(b.slotmap.p), s->work.p, s->cs));
            EMB_TRY(tick(s, b, 0, true));
        } else {         // nothing to ask for: all counts (and peaks) zero; this rank still serves
            HIP_TRY(pimemb::launch_zero_words(static_cast<uint32_t *>(b.meta.p), pimemb::route_meta_counts_words(Kr, N), s->cs));
        }
    }
    g_hp.lap(1);
    // whole tables: what this rank asks each owner for is known on the host
    bool remote_whole = false;
    if (s->Wtot) {
        for (uint32_t p = 0, w = 0; p < N; p++)
            for (uint32_t t : s->whole_of[p]) {
                const emb_shard_input &u = b.in[t];
                uint32_t *c = b.wc_host + (size_t)w++ * kWholeWords;
                c[0] = (uint32_t)b.n_bags;
                c[1] = (uint32_t)u.n_indices;
                c[2] = u.offsets ? 0u : u.fixed_pooling;
                c[3] = b.itype;                                // width of the arrays that travel (or are gathered in place)
                if (via(s, (int)p) == PEER && b.n_bags) {     // the owner gathers / stores in place: the arrays must be where it can
                    if (!pimemb::peer_owns(s->peer, u.indices, u.n_indices * isz) || (u.offsets && !pimemb::peer_owns(s->peer, u.offsets, b.n_bags * isz)) ||
                        !pimemb::peer_owns(s->peer, u.pooled, b.n_bags * (uint64_t)s->dim * 4))
                        return fail(EMB_ERR_INVALID, "emb_shard_submit: table %u is held by rank %u: with EMB_SHARD_PEER_STORES its indices / offsets / "
                                    "pooled buffers must come from emb_peer_alloc", t, p);
                    const uint64_t io = arena_off(s, u.indices), oo = u.offsets ? arena_off(s, u.offsets) : 0, ro = arena_off(s, u.pooled);
                    c[4] = (uint32_t)io; c[5] = (uint32_t)(io >> 32);
                    c[6] = (uint32_t)oo; c[7] = (uint32_t)(oo >> 32);
                    c[8] = (uint32_t)ro; c[9] = (uint32_t)(ro >> 32);
                }
            }
        for (uint32_t p = 0; p < N; p++) remote_whole |= via_comm(s, (int)p) && !s->whole_of[p].empty();
        if (remote_whole)
            HIP_TRY(hipMemcpyAsync(b.wc_send.p, b.wc_host, (size_t)s->Wtot * kWholeWords * 4, hipMemcpyHostToDevice, s->cs));
    }
    g_hp.lap(2);

    // the counts leave FIRST (emb_host.h:280-287 sends the lengths before every launch)
    s->ops.clear();
    const size_t row_msg = (size_t)(Kr + 1) * 8;      // bytes of one peer's row-count message
    for (uint32_t p = 0; p < N; p++) {
        if (!via_comm(s, (int)p)) continue;
        if (Kr) {
            add_op(s, (int)p, false, static_cast<char *>(b.meta.p) + p * row_msg, row_msg);
            add_op(s, (int)p, true, static_cast<char *>(b.counts_in.p) + p * row_msg, row_msg);
        }
        add_op(s, (int)p, false, static_cast<char *>(b.wc_send.p) + (size_t)s->wc_off[p] * kWholeWords * 4,
               s->whole_of[p].size() * kWholeWords * 4);
        add_op(s, (int)p, true, static_cast<char *>(b.counts_in.p) + N * row_msg + (size_t)p * M * kWholeWords * 4,
               (size_t)M * kWholeWords * 4);
    }
    const bool peers = !s->ops.empty();
    if (peers) {         // the transfer stream takes over behind the router (and behind whatever produced the caller's inputs)
        HIP_TRY(hipEventRecord(b.ev_routed, s->cs));
        HIP_TRY(hipStreamWaitEvent(s->s_comm, b.ev_routed, 0));
        g_hp.lap(3);
        EMB_TRY(exchange(s));
    }
    g_hp.lap(4);
    // ... and reach the host behind a flag word
    *b.flag = 0;
    b.counts_posted = false;
    const bool routed = Kr && !b.direct;
    if (routed || peers) {
        const uint32_t *src[3] = {routed ? static_cast<const uint32_t *>(b.meta.p) : nullptr,
                                  routed && peers ? static_cast<const uint32_t *>(b.counts_in.p) : nullptr,
                                  M && peers ? static_cast<const uint32_t *>(b.counts_in.p) + (size_t)N * (Kr + 1) * 2 : nullptr};
        const uint32_t n[3] = {N * (Kr + 1) * 2, N * (Kr + 1) * 2, N * M * kWholeWords};
        HIP_TRY(pimemb::launch_publish_words(src, n, b.counts_host, b.flag, b.seq + 1, peers ? s->s_comm : s->cs));
        b.counts_posted = true;
    }
    if (s->peer_mode) {        // the same news for the peers: counts + where things are, into their mailboxes, behind the router
        pimemb::PeerPostArgs pa{};
        bool any = false;
        const uint64_t rs = routed ? arena_off(s, b.req_send.p) : 0, rr = routed ? arena_off(s, b.ret_recv.p) : 0;
        uint32_t at = 0;
        for (uint32_t p = 0; p < N; p++) {
            if (via(s, (int)p) != PEER) continue;
            uint32_t *c = b.pc_host + at;
            const uint32_t nw = (uint32_t)s->whole_of[p].size();
            c[0] = (uint32_t)rs; c[1] = (uint32_t)(rs >> 32);
            c[2] = (uint32_t)rr; c[3] = (uint32_t)(rr >> 32);
            c[4] = (uint32_t)b.n_bags; c[5] = nw; c[6] = (b.direct ? 1u : 0u) | (s->check_served ? 2u : 0u) | (b.itype == EMB_IDX_I64 ? 4u : 0u); c[7] = 0;
            memcpy(c + kPeerConstHead, b.wc_host + (size_t)s->wc_off[p] * kWholeWords, (size_t)nw * kWholeWords * 4);
            uint32_t extra = 0;
            if (b.direct) {          // where this rank's raw index arrays and output buffers of the row-split tables sit
                uint32_t *r = c + kPeerConstHead + nw * kWholeWords;
                for (uint32_t k = 0; k < Kr; k++) {
                    const emb_shard_input &u = b.in[s->rows[k]];
                    const uint64_t io = b.n_bags ? arena_off(s, u.indices) : 0, ro = b.n_bags ? arena_off(s, u.pooled) : 0;
                    r[4 * k] = (uint32_t)io; r[4 * k + 1] = (uint32_t)(io >> 32);
                    r[4 * k + 2] = (uint32_t)ro; r[4 * k + 3] = (uint32_t)(ro >> 32);
                }
                extra = 4 * Kr;
            }
            pa.box[p] = (unsigned long long)(uintptr_t)pimemb::peer_box_dev(s->peer, (int)p, s->rank, (uint32_t)(b.seq % pimemb::kPeerSlots));
            pa.consts[p] = c;
            pa.n_consts[p] = kPeerConstHead + nw * kWholeWords + extra;
            at += pa.n_consts[p];
            any = true;
        }
        if (any)
            HIP_TRY(pimemb::launch_peer_post(routed ? static_cast<const uint32_t *>(b.meta.p) : nullptr, Kr, N, pa, s->peer_tag + b.seq + 1, s->cs));
    }
    g_hp.lap(5);

    // L(b): replicated tables, this rank's own bags -- described here, launched by launch_local / together with an older
    // batch's S (stage_serve)
    s->local.clear();
    s->local_ctr.clear();
    s->local_it.clear();
    s->local_of = nullptr;
    if (!s->rep.empty() && b.n_bags) {
        for (uint32_t t : s->rep) {
            const emb_shard_input &u = b.in[t];
            emb_lookup_desc d{};
            d.table_id = s->tabs[t].engine_table;
            d.fixed_pooling = u.offsets ? 0u : u.fixed_pooling;
            d.indices = u.indices;
            d.offsets = u.offsets;
            d.n_indices = u.n_indices;
            d.n_bags = b.n_bags;
            d.pooled = u.pooled;
            s->local.push_back(d);
            s->local_it.push_back(b.itype);
            if (s->check_served) s->local_ctr.push_back(ctr_rep(s, b, (uint32_t)s->local_ctr.size()));
            s->st.local_algorithmic_bytes += u.n_indices * ((uint64_t)s->dim * s->elem_bytes[t] + isz) +
                                             (u.offsets ? b.n_bags * isz : 0) + b.n_bags * (uint64_t)s->dim * 4;
        }
        s->local_of = &b;
    }
    g_hp.lap(6);
    b.stage = ROUTED;
    return EMB_OK;
}

// Carry every collected counter segment to its words with ONE kernel on the caller's stream (see emb_shard::lazy_segs).
int publish_lazy(emb_shard *s) {
    s->lazy_age = 0;
    if (s->lazy_segs.empty()) return EMB_OK;
    for (size_t at = 0; at < s->lazy_segs.size(); at += pimemb::kServedSegs) {
        pimemb::ServedArgs sa{};
        for (size_t i = at; i < s->lazy_segs.size() && sa.n_seg < pimemb::kServedSegs; i++) sa.seg[sa.n_seg++] = s->lazy_segs[i];
        HIP_TRY(pimemb::launch_served_counts(sa, s->cs));
    }
    s->lazy_segs.clear();
    for (Batch &b : s->ring) b.lazy_unpublished = false;
    return EMB_OK;
}

// The served counters of a counted launch -> whoever asked.  Segments that raise no peer's "served" word and are read by this rank
// alone may wait for the shared publish kernel (deferred report); everything else is published behind the launch, as before.
int publish_served(emb_shard *s, const pimemb::ServedArgs &sa, Batch *owners[2]) {
    bool lazy = s->defer_report && sa.n_flag == 0;
    for (uint32_t p = 0; p < (uint32_t)s->N && lazy; p++) lazy = via(s, (int)p) != PEER;
    if (!lazy) return pimemb::launch_served_counts(sa, s->cs) == hipSuccess ? EMB_OK : fail(EMB_ERR_DEVICE, "emb_shard: the served-counts kernel could not be enqueued");
    for (uint32_t i = 0; i < sa.n_seg; i++) s->lazy_segs.push_back(sa.seg[i]);
    for (int k = 0; k < 2; k++)
        if (owners[k]) owners[k]->lazy_unpublished = true;
    return EMB_OK;
}

// L(b) on its own (no older batch is served in this call: the pipeline is filling, or depth 0 / flush order).
int launch_local(emb_shard *s) {
    if (s->local.empty()) return EMB_OK;
    Batch &b = *s->local_of;
    s->descs.swap(s->local);
    s->desc_ctr.swap(s->local_ctr);
    s->desc_it.swap(s->local_it);
    s->local.clear();
    s->local_ctr.clear();
    s->local_it.clear();
    s->local_of = nullptr;
    // a checked shard whose replicated tables all take one index per bag: the counted ranged launch (a whole table is the
    // range from row 0) instead of validation kernel + host round trip; the requester -- this rank -- compares at U(b)
    const bool counted = s->check_served && s->allow_direct && all_one_hot(s->descs);
    if (counted) s->row_lo.assign(s->descs.size(), EMB_RANGE_OPEN_END);     // (a whole table: the range from row 0, ids beyond it pool to zero rows)
    EMB_TRY(tick(s, b, 1, false));
    s->refused = 0;
    EMB_TRY(fused_lookup(s, b, true, /*ranged=*/counted));
    if (s->refused) {
        b.deferred_rc = EMB_ERR_RANGE;
        b.reported = false;
    }
    EMB_TRY(tick(s, b, 1, true));
    if (counted) {
        pimemb::ServedArgs sa{};
        sa.seg[sa.n_seg++] = pimemb::ServedSeg{ctr_rep(s, b, 0), b.chk_host, (uint32_t)s->descs.size(), 0u, served_tag(s, b.seq), 0u};
        Batch *owners[2] = {&b, nullptr};
        EMB_TRY(publish_served(s, sa, owners));
        b.rep_counted = true;
    }
    return EMB_OK;
}

// ---- Q(b): the one host wait, then the request pieces ----------------------------------------------------------------
int stage_request(emb_shard *s, Batch &b) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr, M = s->M;
    g_hp.start();
    const size_t counts_words = (size_t)N * ((Kr + 1) * 4 + M * kWholeWords);
    if (b.counts_posted) {
        const double t0 = now_us();
        volatile unsigned long long *flag = b.flag;
        uint64_t spins = 0;
        while (*flag != b.seq + 1) {
            if ((++spins & 0xfffu) == 0 && now_us() - t0 > s->timeout_s * 1e6) {
                const hipError_t q = hipStreamQuery(s->s_comm);
                return fail(EMB_ERR_DEVICE, "emb_shard: the counts of batch %llu did not arrive within %.0f s (transfer stream: %s) -- a peer "
                            "is missing, or the ranks did not make the same calls", (unsigned long long)b.seq, s->timeout_s,
                            hipGetErrorString(q));
            }
        }
        s->st.us_host_wait_counts += now_us() - t0;
    } else {
        memset(b.counts_host, 0, counts_words * 4);
    }
    g_hp.lap(7);
    uint32_t *sent = sent_counts(s, b), *recv = recv_counts(s, b), *rwhole = recv_whole(s, b);
    if (Kr && b.direct) memset(sent, 0, (size_t)N * (Kr + 1) * 8);      // nothing was routed: this rank asked no shard for sub-bags
    if (!via_comm(s, s->rank)) {       // what this rank asked ITSELF for never travelled
        if (Kr) memcpy(recv + (size_t)s->rank * (Kr + 1) * 2, sent + (size_t)s->rank * (Kr + 1) * 2, (size_t)(Kr + 1) * 8);
        if (M) memcpy(rwhole + (size_t)s->rank * M * kWholeWords, b.wc_host + (size_t)s->wc_off[s->rank] * kWholeWords, (size_t)M * kWholeWords * 4);
    }
    b.from.assign(N, PeerFrom{});
    if (s->peer_mode) {        // what every peer posted for this batch: its counts and where its buffers are
        const double t0 = now_us();
        for (uint32_t p = 0; p < N; p++) {
            if (via(s, (int)p) != PEER) continue;
            pimemb::PeerMsg *box = pimemb::peer_box(s->peer, s->rank, (int)p, (uint32_t)(b.seq % pimemb::kPeerSlots));
            EMB_TRY(poll_word(s, &box->posted, s->peer_tag + b.seq + 1, "did not post its requests of", (int)p, b.seq));
            std::atomic_thread_fence(std::memory_order_acquire);
            const uint32_t *w = box->words;
            uint32_t at = 0;
            if (Kr) {
                memcpy(recv + (size_t)p * (Kr + 1) * 2, w, (size_t)(Kr + 1) * 8);
                b.from[p].piece_word = w[2 * (Kr + 1)];
                b.from[p].ret_row0 = w[2 * (Kr + 1) + 1];
                at = 2 * (Kr + 1) + 2;
            }
            b.from[p].req_send_off = (uint64_t)w[at] | ((uint64_t)w[at + 1] << 32);
            b.from[p].ret_recv_off = (uint64_t)w[at + 2] | ((uint64_t)w[at + 3] << 32);
            if (w[at + 5] != M) return fail(EMB_ERR_INVALID, "emb_shard: rank %u posted %u whole-table entries, this rank owns %u (different placements?)", p, w[at + 5], M);
            if (M) memcpy(rwhole + (size_t)p * M * kWholeWords, w + at + kPeerConstHead, (size_t)M * kWholeWords * 4);
            b.from[p].n_bags = w[at + 4];
            b.from[p].direct = (w[at + 6] & 1u) != 0;
            b.from[p].itype = (w[at + 6] & 4u) ? EMB_IDX_I64 : EMB_IDX_U32;
            if (((w[at + 6] & 2u) != 0) != s->check_served)
                return fail(EMB_ERR_INVALID, "emb_shard: rank %u created its shard %s EMB_SHARD_CHECK_SERVED, this rank %s it (the flags must agree)", p,
                            (w[at + 6] & 2u) ? "with" : "without", s->check_served ? "with" : "without");
            if (b.from[p].direct) {
                const uint32_t *r = w + at + kPeerConstHead + M * kWholeWords;
                b.from[p].idx_off.resize(Kr);
                b.from[p].out_off.resize(Kr);
                for (uint32_t k = 0; k < Kr; k++) {
                    b.from[p].idx_off[k] = (uint64_t)r[4 * k] | ((uint64_t)r[4 * k + 1] << 32);
                    b.from[p].out_off[k] = (uint64_t)r[4 * k + 2] | ((uint64_t)r[4 * k + 3] << 32);
                }
            }
        }
        s->st.us_host_wait_counts += now_us() - t0;
    }
    if (via(s, s->rank) == SELF) {
        b.from[(size_t)s->rank].direct = b.direct;
        b.from[(size_t)s->rank].itype = b.itype;
        b.from[(size_t)s->rank].n_bags = b.n_bags;
    }
    b.out_words.assign(N, 0);
    b.in_words.assign(N, 0);
    b.rows_back.assign(N, 0);
    b.rows_served.assign(N, 0);
    if (Kr) {
        uint64_t pk_req = 0, pk_ret = 0;
        EMB_TRY(emb_route_exchange_sizes(sent, recv, Kr, N, s->dim, b.out_words.data(), b.in_words.data(), b.rows_back.data(),
                                         b.rows_served.data(), &pk_req, &pk_ret));
    }
    // sizes of the three payload buffers.  req_recv: row pieces of source 0, 1, ... then the whole-table arrays; ret_send: partial
    // rows for source 0, 1, ... then the pooled rows of whole tables; ret_recv: partial rows from shard 0, 1, ...
    uint64_t in_w = 0, served = 0, back = 0, whole_in_w = 0, whole_rows = 0;
    for (uint32_t p = 0; p < N; p++) {
        in_w += b.in_words[p];
        served += b.rows_served[p];
        back += b.rows_back[p];
        if (via_comm(s, (int)p))
            for (uint32_t j = 0; j < M; j++) {
                const uint32_t *c = rwhole + ((size_t)p * M + j) * kWholeWords;
                whole_in_w += (c[2] ? 0 : idx_words(c[0], c[3])) + idx_words(c[1], c[3]);
                whole_rows += c[0];
            }
    }
    EMB_TRY(ensure(s, b.req_recv, (in_w + whole_in_w) * 4 + 16));
    EMB_TRY(ensure(s, b.ret_send, (served + whole_rows) * (uint64_t)s->dim * 4 + 16));
    EMB_TRY(ensure(s, b.ret_recv, back * (uint64_t)s->dim * 4 + 16));
    g_hp.lap(8);

    s->ops.clear();
    uint64_t out_at = 0, in_at = 0, win_at = in_w;
    for (uint32_t p = 0; p < N; p++) {
        if (via_comm(s, (int)p)) {
            add_op(s, (int)p, false, static_cast<uint32_t *>(b.req_send.p) + out_at, b.out_words[p] * 4);
            add_op(s, (int)p, true, static_cast<uint32_t *>(b.req_recv.p) + in_at, b.in_words[p] * 4);
            for (uint32_t t : s->whole_of[p]) {          // straight out of the caller's buffers
                const emb_shard_input &u = b.in[t];
                if (u.offsets) add_op(s, (int)p, false, u.offsets, b.n_bags * isz_of(b.itype));
                add_op(s, (int)p, false, u.indices, u.n_indices * isz_of(b.itype));
            }
            for (uint32_t j = 0; j < M; j++) {
                const uint32_t *c = rwhole + ((size_t)p * M + j) * kWholeWords;
                if (c[3] > EMB_IDX_I64) return fail(EMB_ERR_INVALID, "emb_shard: rank %u sent an unknown index width (%u)", p, c[3]);
                if (!c[2]) {
                    add_op(s, (int)p, true, static_cast<uint32_t *>(b.req_recv.p) + win_at, (uint64_t)c[0] * isz_of(c[3]));
                    win_at += idx_words(c[0], c[3]);
                }
                add_op(s, (int)p, true, static_cast<uint32_t *>(b.req_recv.p) + win_at, (uint64_t)c[1] * isz_of(c[3]));
                win_at += idx_words(c[1], c[3]);
            }
        }
        out_at += b.out_words[p];
        in_at += b.in_words[p];
    }
    if (!s->ops.empty()) {          // (the transfer stream is already behind R(b): it carried the counts)
        EMB_TRY(exchange(s));
        HIP_TRY(hipEventRecord(b.ev_req, s->s_comm));
        b.req_recorded = true;
    }
    g_hp.lap(9);
    b.stage = REQUESTED;
    return EMB_OK;
}

// ---- S(b) + T(b) ---------------------------------------------------------------------------------------------------------
// The request pieces this rank serves for batch b (s->piece_refs: uint32 words at piece_words[source]) widened into b.wide, the
// piece descriptors pointed there and marked int64.  Source p's piece keeps its place: word w of it is word (words before p) + w.
int widen_pieces(emb_shard *s, Batch &b, const std::vector<const uint32_t *> &piece_words) {
    const uint32_t N = (uint32_t)s->N;
    uint64_t total = 0;
    for (uint32_t p = 0; p < N; p++) total += b.in_words[p];
    if (total == 0 || s->piece_refs.empty()) return EMB_OK;
    EMB_TRY(ensure(s, b.wide, total * 8));
    long long *wide = static_cast<long long *>(b.wide.p);
    std::vector<uint64_t> at(N, 0);
    pimemb::WidenArgs a{};
    uint64_t run = 0;
    for (uint32_t p = 0; p < N; p++) {
        at[p] = run;
        run += b.in_words[p];
        if (b.in_words[p] == 0 || !piece_words[p]) continue;
        a.src[a.n_seg] = piece_words[p];
        a.dst[a.n_seg] = wide + at[p];
        a.n[a.n_seg] = b.in_words[p];
        if (++a.n_seg == pimemb::kWidenSegs) {
            HIP_TRY(pimemb::launch_widen_words(a, s->cs));
            a.n_seg = 0;
        }
    }
    HIP_TRY(pimemb::launch_widen_words(a, s->cs));
    for (const emb_shard::PieceRef &r : s->piece_refs) {
        emb_lookup_desc &d = s->descs[r.desc];
        d.offsets = wide + at[r.src] + r.off_word;
        d.indices = wide + at[r.src] + r.idx_word;
        s->desc_it[r.desc] = EMB_IDX_I64;
    }
    return EMB_OK;
}

int stage_serve(emb_shard *s, Batch &b) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr, M = s->M, dim = s->dim;
    uint32_t *recv = recv_counts(s, b), *rwhole = recv_whole(s, b);
    g_hp.start();
    if (b.req_recorded) HIP_TRY(hipStreamWaitEvent(s->cs, b.ev_req, 0));      // the pieces have arrived
    uint64_t in_w = 0, served = 0;
    for (uint32_t p = 0; p < N; p++) {
        in_w += b.in_words[p];
        served += b.rows_served[p];
    }
    // ONE launch for everything this call looks up: the replicated tables of the batch being submitted (another batch when
    // pipelined: L(n) next to S(n - 2); this very batch at depth 0) and every piece received for this one
    Batch *fused_with = (s->local_of && s->local_of != &b) ? s->local_of : nullptr;
    s->descs.clear();
    s->desc_ctr.clear();
    s->desc_it.clear();
    Batch *local_batch = s->local_of;       // whose replicated tables ride in this call's launch (this batch, a younger one, or none)
    if (s->local_of) {
        s->descs.swap(s->local);
        s->desc_ctr.swap(s->local_ctr);
        s->desc_it.swap(s->local_it);
        s->local.clear();
        s->local_ctr.clear();
        s->local_it.clear();
        s->local_of = nullptr;
    }
    const size_t n_local_descs = s->descs.size();
    const bool counting = s->check_served;
    bool own_width[2] = {false, false};      // index widths among THIS batch's descriptors (a refused launch of that width zeroed them)
    if (local_batch == &b && n_local_descs) own_width[b.itype] = true;
    uint64_t alg = 0, n_sub = 0, n_idx = 0;
    uint32_t n_piece_descs = 0;
    // row pieces: source s asked for sub-bags of my shard of table k -- an ordinary lookup each
    uint64_t in_at = 0, out_at = 0, served_at = 0, back_at = 0;
    s->piece_refs.clear();
    std::vector<const uint32_t *> piece_words(N, nullptr);
    for (uint32_t p = 0; p < N; p++) {
        const Via how = via(s, (int)p);
        // the request piece: where it arrived (RCCL), where it was written (my own), or where it sits in the peer's HBM
        const uint32_t *words = how == COMM ? static_cast<uint32_t *>(b.req_recv.p) + in_at
                              : how == SELF ? static_cast<uint32_t *>(b.req_send.p) + out_at
                                            : reinterpret_cast<uint32_t *>(pimemb::peer_ptr(s->peer, (int)p, b.from[p].req_send_off + 4ull * b.from[p].piece_word, b.in_words[p] * 4));
        // rows for a remote source go to ret_send (T sends them); my own go where the un-router reads shard `rank`'s rows; a
        // peer's go straight into ITS ret_recv, at the row where it expects this shard's
        float *rows_dst = how == COMM ? static_cast<float *>(b.ret_send.p) + served_at * dim
                        : how == SELF ? static_cast<float *>(b.ret_recv.p) + back_at * dim
                                      : reinterpret_cast<float *>(pimemb::peer_ptr(s->peer, (int)p, b.from[p].ret_recv_off + (uint64_t)b.from[p].ret_row0 * dim * 4, b.rows_served[p] * (uint64_t)dim * 4));
        if (how == PEER && (b.in_words[p] || b.rows_served[p]) && (!words || !rows_dst))
            return fail(EMB_ERR_INVALID, "emb_shard: rank %u posted buffer offsets outside its arena", p);
        if (how == PEER) s->st.bytes_to_peers += b.in_words[p] * 4 + b.rows_served[p] * (uint64_t)dim * 4;     // read from / stored into the peer's HBM
        piece_words[p] = words;
        uint64_t cur = 0, row = 0;
        for (uint32_t k = 0; k < Kr; k++) {
            const uint64_t ns = recv[((size_t)p * (Kr + 1) + k) * 2], ni = recv[((size_t)p * (Kr + 1) + k) * 2 + 1];
            if (ns) {
                emb_lookup_desc d{};
                d.table_id = s->tabs[s->rows[k]].engine_table;
                d.offsets = words + cur;
                d.indices = words + cur + pad4(ns);
                d.n_indices = ni;
                d.n_bags = ns;
                d.pooled = rows_dst + row * dim;
                s->piece_refs.push_back(emb_shard::PieceRef{(uint32_t)s->descs.size(), p, cur, cur + pad4(ns)});
                s->descs.push_back(d);
                s->desc_it.push_back(EMB_IDX_U32);                 // (a routed piece: uint32 local row ids + sub-bag starts, whatever the source handed in)
                own_width[EMB_IDX_U32] = true;
                if (counting) s->desc_ctr.push_back(nullptr);      // (a routed piece has offsets: never part of a counted launch)
                n_piece_descs++;
                alg += ni * ((uint64_t)dim * s->elem_bytes[s->rows[k]] + 4) + ns * (4 + (uint64_t)dim * 4);
                n_sub += ns;
                n_idx += ni;
            }
            cur += pad4(ns) + pad4(ni);
            row += ns;
        }
        in_at += b.in_words[p];
        out_at += b.out_words[p];
        served_at += b.rows_served[p];
        back_at += b.rows_back[p];
    }
    const uint64_t n_sub_pieces = n_sub, n_idx_pieces = n_idx;
    // whole tables owned here: source p's bags exactly as p's caller passed them
    uint64_t win_at = in_w, wrow_at = served;
    for (uint32_t p = 0; p < N; p++) {
        const Via how = via(s, (int)p);
        const bool remote = how == COMM;
        for (uint32_t j = 0; j < M; j++) {
            const uint32_t t = s->whole_of[s->rank][j];
            const uint32_t *c = rwhole + ((size_t)p * M + j) * kWholeWords;
            const uint64_t nb = c[0], ni = c[1];
            emb_lookup_desc d{};
            d.table_id = s->tabs[t].engine_table;
            d.n_indices = ni;
            d.n_bags = nb;
            uint32_t it = c[3];              // the requester's width (its arrays arrive / are gathered as they are)
            if (it > EMB_IDX_I64) return fail(EMB_ERR_INVALID, "emb_shard: rank %u posted an unknown index width (%u)", p, it);
            if (remote) {
                d.fixed_pooling = c[2];
                if (!c[2]) {
                    d.offsets = static_cast<uint32_t *>(b.req_recv.p) + win_at;
                    win_at += idx_words(nb, it);
                }
                d.indices = static_cast<uint32_t *>(b.req_recv.p) + win_at;
                win_at += idx_words(ni, it);
                d.pooled = static_cast<float *>(b.ret_send.p) + wrow_at * dim;
                wrow_at += nb;
            } else if (how == PEER) {      // the requester's arrays in place, its output buffer in place
                const uint64_t io = (uint64_t)c[4] | ((uint64_t)c[5] << 32), oo = (uint64_t)c[6] | ((uint64_t)c[7] << 32), ro = (uint64_t)c[8] | ((uint64_t)c[9] << 32);
                d.fixed_pooling = c[2];
                d.offsets = c[2] ? nullptr : pimemb::peer_ptr(s->peer, (int)p, oo, nb * isz_of(it));
                d.indices = pimemb::peer_ptr(s->peer, (int)p, io, ni * isz_of(it));
                d.pooled = reinterpret_cast<float *>(pimemb::peer_ptr(s->peer, (int)p, ro, nb * (uint64_t)dim * 4));
                if (nb && ((ni && !d.indices) || !d.pooled || (!c[2] && !d.offsets)))
                    return fail(EMB_ERR_INVALID, "emb_shard: rank %u posted buffer offsets outside its arena", p);
                s->st.bytes_to_peers += ni * isz_of(it) + (c[2] ? 0 : nb * isz_of(it)) + nb * (uint64_t)dim * 4;
            } else {           // my own bags of my own table: in place, straight into the caller's buffer
                const emb_shard_input &u = b.in[t];
                it = b.itype;
                d.fixed_pooling = u.offsets ? 0u : u.fixed_pooling;
                d.offsets = u.offsets;
                d.indices = u.indices;
                d.pooled = u.pooled;
            }
            if (nb) {
                s->descs.push_back(d);
                s->desc_it.push_back(it);
                own_width[it] = true;
                if (counting) s->desc_ctr.push_back(ctr_of(s, b, p, Kr + j));
                alg += ni * ((uint64_t)dim * s->elem_bytes[t] + isz_of(it)) + (d.offsets ? nb * isz_of(it) : 0) + nb * (uint64_t)dim * 4;
                n_sub += nb;
                n_idx += ni;
            }
        }
    }
    g_hp.lap(10);
    s->st.served_algorithmic_bytes += alg;
    s->st.served_sub_bags += n_sub;
    s->st.served_indices += n_idx;
    // sources that handed their one-index-per-bag row-split tables over directly: scan their raw index arrays, serve the bags
    // whose row this shard holds, straight into their outputs
    s->rdescs.clear();
    s->rdesc_ctr.clear();
    s->rdesc_it.clear();
    s->row_lo.clear();
    // (a checked shard: the LAST shard of a table also answers for ids beyond every range -- their bags pool to zero rows)
    const uint64_t open_end = (counting && s->rank + 1 == s->N) ? EMB_RANGE_OPEN_END : 0ull;
    for (uint32_t p = 0; p < N; p++) {
        const PeerFrom &f = b.from[p];
        if (!f.direct || f.n_bags == 0) continue;
        const Via how = via(s, (int)p);
        for (uint32_t k = 0; k < Kr; k++) {
            const uint32_t t = s->rows[k];
            emb_lookup_desc d{};
            d.table_id = s->tabs[t].engine_table;
            d.fixed_pooling = 1;
            d.n_indices = d.n_bags = f.n_bags;
            if (how == SELF) {
                d.indices = b.in[t].indices;
                d.pooled = b.in[t].pooled;
            } else {
                d.indices = pimemb::peer_ptr(s->peer, (int)p, f.idx_off[k], f.n_bags * isz_of(f.itype));
                d.pooled = reinterpret_cast<float *>(pimemb::peer_ptr(s->peer, (int)p, f.out_off[k], f.n_bags * (uint64_t)dim * 4));
                if (!d.indices || !d.pooled) return fail(EMB_ERR_INVALID, "emb_shard: rank %u posted buffer offsets outside its arena", p);
                s->st.bytes_to_peers += f.n_bags * isz_of(f.itype);       // (+ the rows this shard stores: their number is not known on the host)
            }
            s->rdescs.push_back(d);
            s->rdesc_it.push_back(f.itype);
            if (counting) s->rdesc_ctr.push_back(ctr_of(s, b, p, k));
            s->row_lo.push_back(((uint64_t)s->rank * s->tabs[t].rows_per_shard) | open_end);
        }
    }
    // ... in the SAME launch as everything else this call looks up when that is one index per bag as well (a whole table is
    // the range starting at row 0): the tuned one-hot kernel once, over replicated tables, whole tables and shards alike
    // (a checked shard's merged launch COUNTS what it serves -- the requesters compare at U(b) -- instead of validating first)
    // ... and a checked shard with no peer behind RCCL takes the counted launch for ANY all-one-index call, row-split tables
    // or not: counting costs nothing, validating first costs a kernel and a host round trip on the serving rank
    bool no_comm = true;
    for (uint32_t p = 0; p < N; p++) no_comm = no_comm && via(s, (int)p) != COMM;
    const bool one_launch = !s->descs.empty() && s->merge_direct && all_one_hot(s->descs) &&
                            (!s->rdescs.empty() || (counting && s->allow_direct && no_comm));
    const bool ranged_launch = one_launch || !s->rdescs.empty();
    if (one_launch) {
        s->row_lo.insert(s->row_lo.begin(), s->descs.size(), counting ? EMB_RANGE_OPEN_END : 0ull);      // (whole tables: the range from row 0)
        s->descs.insert(s->descs.end(), s->rdescs.begin(), s->rdescs.end());
        s->desc_ctr.insert(s->desc_ctr.end(), s->rdesc_ctr.begin(), s->rdesc_ctr.end());
        s->desc_it.insert(s->desc_it.end(), s->rdesc_it.begin(), s->rdesc_it.end());
        s->rdescs.clear();
        EMB_TRY(tick(s, b, 4, false));
        EMB_TRY(fused_lookup(s, b, /*cacheable=*/n_piece_descs == 0, /*ranged=*/true));
        EMB_TRY(tick(s, b, 4, true));
    } else {
        if (!s->descs.empty()) {
            EMB_TRY(tick(s, b, 2, false));
            s->refused = 0;
            // an int64 job: the pieces are the only uint32 arrays of the call, and with about one index per sub-bag a launch of
            // their own is a SMALL launch (C4's share, one index per bag: 6 x 16 384 sub-bags = the lane-group kernel at 27 us;
            // riding in the big launch they cost 5) -- widen them once (2.4 MB there) and everything goes out as ONE int64 launch:
            // 115 -> 95 us per step, uint32 job 91 (tools/ab_widen_pieces.sh).  Pooled pieces stay uint32 and a launch of their own:
            // at 32 indices per bag they are a big launch anyway, and 8-byte ids cost more than the second launch (612 vs 597 us).
            if (n_piece_descs && s->widen && n_idx_pieces <= 2 * n_sub_pieces) {
                bool any64 = false;
                for (uint32_t it : s->desc_it) any64 = any64 || it == EMB_IDX_I64;
                if (any64) {
                    EMB_TRY(widen_pieces(s, b, piece_words));
                    own_width[0] = own_width[1] = false;
                    if (local_batch == &b && n_local_descs) own_width[b.itype] = true;
                    for (size_t i = n_local_descs; i < s->desc_it.size(); i++) own_width[s->desc_it[i]] = true;
                }
            }
            EMB_TRY(fused_lookup(s, b, /*cacheable=*/n_piece_descs == 0));      // (row pieces change size with every batch: nothing recurs)
            EMB_TRY(tick(s, b, 2, true));
            // a validating launch that found something gathered nothing: whoever had descriptors in it holds zero rows and says so
            for (uint32_t w = 0; w < 2; w++) {
                if (!(s->refused & (1u << w))) continue;
                if (own_width[w]) {
                    b.deferred_rc = EMB_ERR_RANGE;
                    b.reported = false;
                }
                if (fused_with && fused_with->itype == w) {       // (its replicated tables rode in the refused launch)
                    fused_with->deferred_rc = EMB_ERR_RANGE;
                    fused_with->reported = false;
                }
            }
        }
        if (!s->rdescs.empty()) {
            s->descs.swap(s->rdescs);
            s->desc_ctr.swap(s->rdesc_ctr);
            s->desc_it.swap(s->rdesc_it);
            EMB_TRY(tick(s, b, 4, false));
            EMB_TRY(fused_lookup(s, b, /*cacheable=*/true, /*ranged=*/true));
            EMB_TRY(tick(s, b, 4, true));
        }
    }
    if (counting && (ranged_launch || s->peer_mode)) {
        // checked: the counters of the counted launch go to whoever asked -- this rank's own pinned words, the tail of a
        // peer's mailbox -- and the "served" words are raised behind them (this kernel instead of launch_peer_done).  What
        // went through the checked (validating) launch instead is marked 0xffffffff: "not counted, validated by its server".
        pimemb::ServedArgs sa{};
        if (one_launch && n_local_descs && local_batch) {
            sa.seg[sa.n_seg++] = pimemb::ServedSeg{ctr_rep(s, *local_batch, 0), local_batch->chk_host, (uint32_t)n_local_descs, 0u,
                                                   served_tag(s, local_batch->seq), 0u};
            local_batch->rep_counted = true;
        }
        for (uint32_t p = 0; p < N; p++) {
            const Via how = via(s, (int)p);
            if (how == COMM) continue;          // (no direct path, no counted launch with a peer behind RCCL)
            const uint32_t n_counted = one_launch ? Kr + M : (b.from[p].direct ? Kr : 0u);
            unsigned long long *dst = b.chk_host + s->rep.size();
            if (how == PEER) {
                pimemb::PeerMsg *box = pimemb::peer_box_dev(s->peer, (int)p, s->rank, (uint32_t)(b.seq % pimemb::kPeerSlots));
                dst = reinterpret_cast<unsigned long long *>(box->words + served_tail(s));
                sa.flag[sa.n_flag] = (unsigned long long)(uintptr_t)&box->served;
                sa.value[sa.n_flag++] = s->peer_tag + b.seq + 1;
            } else {
                b.self_published = true;
            }
            sa.seg[sa.n_seg++] = pimemb::ServedSeg{ctr_of(s, b, p, 0), dst, n_counted, Kr + M - n_counted, served_tag(s, b.seq), 0u};
        }
        Batch *owners[2] = {&b, (one_launch && n_local_descs) ? local_batch : nullptr};
        EMB_TRY(publish_served(s, sa, owners));
    } else if (s->peer_mode) {        // behind the lookup (its kernel boundary completes the stores into the peers' HBM): "served"
        pimemb::PeerDoneArgs da{};
        for (uint32_t p = 0; p < N; p++)
            if (via(s, (int)p) == PEER)
                da.box[da.n++] = (unsigned long long)(uintptr_t)pimemb::peer_box_dev(s->peer, (int)p, s->rank, (uint32_t)(b.seq % pimemb::kPeerSlots));
        HIP_TRY(pimemb::launch_peer_done(da, s->peer_tag + b.seq + 1, s->cs));
    }
    g_hp.lap(11);

    // T(b): partial rows back to the bags' owners; whole tables' pooled rows straight into the callers' buffers
    s->ops.clear();
    served_at = back_at = 0;
    wrow_at = served;
    for (uint32_t p = 0; p < N; p++) {
        if (via_comm(s, (int)p)) {
            add_op(s, (int)p, false, static_cast<float *>(b.ret_send.p) + served_at * dim, b.rows_served[p] * (uint64_t)dim * 4);
            add_op(s, (int)p, true, static_cast<float *>(b.ret_recv.p) + back_at * dim, b.rows_back[p] * (uint64_t)dim * 4);
            for (uint32_t j = 0; j < M; j++) {
                const uint64_t nb = rwhole[((size_t)p * M + j) * kWholeWords];
                add_op(s, (int)p, false, static_cast<float *>(b.ret_send.p) + wrow_at * dim, nb * (uint64_t)dim * 4);
                wrow_at += nb;
            }
            for (uint32_t t : s->whole_of[p]) add_op(s, (int)p, true, b.in[t].pooled, b.n_bags * (uint64_t)dim * 4);
        }
        served_at += b.rows_served[p];
        back_at += b.rows_back[p];
    }
    if (!s->ops.empty()) {
        HIP_TRY(hipEventRecord(b.ev_served, s->cs));
        HIP_TRY(hipStreamWaitEvent(s->s_comm, b.ev_served, 0));
        EMB_TRY(exchange(s));
        HIP_TRY(hipEventRecord(b.ev_ret, s->s_comm));
        b.ret_recorded = true;
    }
    g_hp.lap(12);
    b.stage = SERVED;
    return EMB_OK;
}

// Checked shards, on the REQUESTING rank: every bag of a one-index batch has exactly one server -- the replicated copy here, the
// whole table's owner, or the one shard holding the row -- so the counts of the counted launches must add up to the batch's bag
// count per table.  A sum short of it is an index no shard holds: a bag nobody wrote (the ranged lookup leaves it untouched).
// Called at U(b): the launches that counted were enqueued one or more calls ago (depth 0: in this call), the peers' "served"
// words have been polled; what is left is the wait for this rank's own publish kernels.
// wait = false: look only -- if any entry does not carry this batch's tag yet (its publish kernel has not run), return
// kNotReady and change nothing; the caller asks again later.
constexpr int kNotReady = 1;
int check_served_counts(emb_shard *s, Batch &b, bool wait = true) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr;
    const uint32_t R = (uint32_t)s->rep.size();
    const uint32_t tag = served_tag(s, b.seq);
    const double t0 = now_us();
    int rc = EMB_OK;
    bool probing = !wait, not_ready = false;
    // one entry: poll the self-describing word until it carries this batch's tag (the kernel that stores it was enqueued a
    // call or more ago; a peer's: its "served" word has been seen, the entries are on their way), hand back the count
    auto entry = [&](volatile unsigned long long *w, const char *what, uint32_t who) -> uint32_t {
        for (uint64_t spin = 0;; spin++) {
            const unsigned long long v = *w;
            if ((uint32_t)(v >> 32) == tag) return (uint32_t)v;
            if (probing) {
                not_ready = true;
                return 0u;
            }
            if ((spin & 0xfff) == 0xfff && now_us() - t0 > s->timeout_s * 1e6) {
                if (rc == EMB_OK)
                    rc = fail(EMB_ERR_DEVICE, "emb_shard: the served counts of batch %llu (%s, rank %u) did not arrive within %.0f s", (unsigned long long)b.seq, what, who, s->timeout_s);
                return 0xffffffffu;
            }
        }
    };
    auto bad = [&](uint32_t t, uint64_t got) {
        if (probing) return;
        if (b.deferred_rc == EMB_OK && !s->range_msg[0])
            snprintf(s->range_msg, sizeof s->range_msg, "emb_shard: batch %llu, table %u: %llu of %llu bags were served -- the others name rows no rank holds "
                     "(their pooled rows are zeros)", (unsigned long long)b.seq, t, (unsigned long long)got, (unsigned long long)b.n_bags);
        b.deferred_rc = EMB_ERR_RANGE;
        b.reported = false;
    };
    // what shard / owner q published for this rank's i-th piece (0xffffffff: not counted -- q validated it itself)
    auto word_from = [&](uint32_t q, uint32_t i) -> uint32_t {
        const Via how = via(s, (int)q);
        if (how == SELF) return b.self_published ? entry(b.chk_host + R + i, "this rank's own pieces", q) : 0xffffffffu;
        if (how == PEER) {
            pimemb::PeerMsg *box = pimemb::peer_box(s->peer, s->rank, (int)q, (uint32_t)(b.seq % pimemb::kPeerSlots));
            return entry(reinterpret_cast<volatile unsigned long long *>(box->words + served_tail(s)) + i, "a peer's", q);
        }
        return 0xffffffffu;
    };
    auto pass = [&]() {
        if (b.rep_counted)
            for (uint32_t r = 0; r < R; r++) {
                const uint32_t w = entry(b.chk_host + r, "replicated tables", (uint32_t)s->rank);
                if (rc == EMB_OK && w != b.n_bags) bad(s->rep[r], w);
            }
        for (uint32_t q = 0; q < N && rc == EMB_OK; q++)
            for (uint32_t j = 0; j < s->whole_of[q].size(); j++) {
                const uint32_t w = word_from(q, Kr + j);
                if (rc == EMB_OK && w != 0xffffffffu && w != b.n_bags) bad(s->whole_of[q][j], w);
            }
        if (b.direct)
            for (uint32_t k = 0; k < Kr && rc == EMB_OK; k++) {
                uint64_t sum = 0;
                for (uint32_t q = 0; q < N && rc == EMB_OK; q++) {
                    const uint32_t w = word_from(q, k);
                    if (!probing && rc == EMB_OK && w == 0xffffffffu)
                        rc = fail(EMB_ERR_INVALID, "emb_shard: rank %u did not count what it served of batch %llu (different EMB_SHARD_* flags?)", q, (unsigned long long)b.seq);
                    sum += w;
                }
                if (rc == EMB_OK && sum != b.n_bags) bad(s->rows[k], sum);
            }
    };
    if (probing) {
        pass();
        if (not_ready) return kNotReady;
        probing = false;
    }
    pass();
    std::atomic_thread_fence(std::memory_order_acquire);
    s->st.us_host_wait_served += now_us() - t0;
    return rc;
}

// ---- U(b): partial rows added in shard order, into the caller's buffers ------------------------------------------------------
// A wait that times out here (a peer that never served, served counts that never arrived) is REPORTED, not left behind: the
// stage still runs to its end and the batch is marked done, so that later calls do not poll for the same words again -- one
// lost 8-byte store would otherwise make every later submit / flush block for the full timeout, and emb_shard_wait never return.
int stage_unroute(emb_shard *s, Batch &b) {
    g_hp.start();
    int late = EMB_OK;
    if (s->peer_mode) {        // every peer has stored what it owes this rank for the batch (host poll: nothing to enqueue)
        const double t0 = now_us();
        for (uint32_t p = 0; p < (uint32_t)s->N && late == EMB_OK; p++)
            if (via(s, (int)p) == PEER) {
                pimemb::PeerMsg *box = pimemb::peer_box(s->peer, s->rank, (int)p, (uint32_t)(b.seq % pimemb::kPeerSlots));
                late = poll_word(s, &box->served, s->peer_tag + b.seq + 1, "did not serve", (int)p, b.seq);
            }
        std::atomic_thread_fence(std::memory_order_acquire);
        s->st.us_host_wait_served += now_us() - t0;
    }
    if (b.ret_recorded) HIP_TRY(hipStreamWaitEvent(s->cs, b.ev_ret, 0));       // the rows are back (whole tables: already in place)
    if (s->check_served && b.n_bags && late == EMB_OK) {
        if (s->defer_report) b.check_pending = true;       // compared at the start of the next call (collect_pending): nothing to wait for here
        else late = check_served_counts(s, b);
    }
    if (s->Kr && b.n_bags && !b.direct) {
        float *outs[pimemb::kRouteBagMaxTables];
        for (uint32_t k = 0; k < s->Kr; k++) outs[k] = b.in[s->rows[k]].pooled;
        EMB_TRY(tick(s, b, 3, false));
        HIP_TRY(pimemb::launch_unroute_bags_to(static_cast<float *>(b.ret_recv.p), static_cast<uint32_t *>(b.meta.p),
                                               static_cast<uint32_t *>(b.slotmap.p), s->Kr, b.n_bags, (uint32_t)s->N, s->dim, outs, s->cs));
        EMB_TRY(tick(s, b, 3, true));
    }
    g_hp.lap(13);
    b.stage = DONE;
    s->st.n_batches++;
    return late;
}

// A batch's finding (EMB_ERR_RANGE: the serving side's validation, or the requester's served counts) is handed to the caller
// ONCE, by the call in which the batch completes -- or, for a deferred comparison, by the call that makes it.
inline int take_report(Batch &b) {
    if (b.deferred_rc == EMB_OK || b.reported) return EMB_OK;
    b.reported = true;
    return b.deferred_rc;
}

// EMB_SHARD_DEFER_REPORT: compare the served counts of the completed batches not looked at yet -- all of them, or batch `only`.
// wait = false (what a submit does): only those whose counts have ARRIVED; a batch whose publish kernel has not run yet stays
// pending -- polling for it would make the host wait for the GPU step it has just enqueued, which is the very round trip the
// flag exists to remove (first version: "at the start of the next call" = 73 us per synchronous call against 63 unchecked) --
// except `must`, the batch whose ring slot the caller is about to recycle.
int collect_pending(emb_shard *s, bool wait = true, uint64_t only = ~0ull, uint64_t must = ~0ull) {
    int deferred = EMB_OK;
    for (Batch &b : s->ring) {
        if (!b.check_pending || b.stage != DONE || (only != ~0ull && b.seq != only)) continue;
        if ((wait || b.seq == must) && b.lazy_unpublished) EMB_TRY(publish_lazy(s));       // (about to be waited for: its counters must be on their way)
        const int rc = check_served_counts(s, b, wait || b.seq == must);
        if (rc == kNotReady) continue;
        b.check_pending = false;
        if (rc != EMB_OK) return rc;
        if (take_report(b) != EMB_OK) deferred = EMB_ERR_RANGE;
    }
    return deferred;
}

// (requests, lookups + returns, un-routing) of a batch happen (d_req, d_serve, d_un) submits after its own
struct Lag { uint64_t req, serve, un; };
constexpr Lag kLag[4] = {{0, 0, 0}, {0, 1, 1}, {1, 2, 2}, {1, 2, 3}};

// Advance every live batch that is old enough, oldest first within a stage -- the same order on every rank.  Per call the
// transfer stream gets: counts of the newest batch, THEN the requests of the one before it, THEN the returned rows of the
// one before that -- so a lookup never queues behind a return transfer -- and the un-router of a batch runs at the END of a
// call, behind the lookup of a younger one: its rows have had that whole call to come back.
// Returns EMB_ERR_RANGE when a batch that COMPLETED in this call carries a finding (each batch's once: take_report).
int advance(emb_shard *s, const Lag &lag, bool everything) {
    int deferred = EMB_OK;
    const uint64_t newest = s->next_seq;        // one past the last submitted
    const uint64_t first = newest > kRing ? newest - kRing : 0;
    auto old_enough = [&](uint64_t q, uint64_t d) { return everything || q + d < newest; };
    for (uint64_t q = first; q < newest; q++) {
        Batch &b = s->ring[q % kRing];
        if (b.seq == q && b.stage == ROUTED && old_enough(q, lag.req)) EMB_TRY(stage_request(s, b));
    }
    for (uint64_t q = first; q < newest; q++) {
        Batch &b = s->ring[q % kRing];
        if (b.seq == q && b.stage == REQUESTED && old_enough(q, lag.serve)) EMB_TRY(stage_serve(s, b));
    }
    EMB_TRY(launch_local(s));          // nothing was served in this call: L(n) alone
    int late = EMB_OK;
    for (uint64_t q = first; q < newest; q++) {
        Batch &b = s->ring[q % kRing];
        if (b.seq == q && b.stage == SERVED && old_enough(q, lag.un)) {
            const int rc = stage_unroute(s, b);
            if (rc != EMB_OK && late == EMB_OK) late = rc;       // (a timeout: the other batches still go through their stage)
            if (take_report(b) != EMB_OK) deferred = EMB_ERR_RANGE;
        }
    }
    return late != EMB_OK ? late : deferred;
}

int range_error(emb_shard *s) {
    if (s->range_msg[0]) {          // found by this rank as the REQUESTER (served counts short of the bag count)
        const int rc = fail(EMB_ERR_RANGE, "%s", s->range_msg);
        s->range_msg[0] = 0;
        return rc;
    }
    return fail(EMB_ERR_RANGE, "emb_shard: a piece served by this rank named rows outside its table (pooled to zero rows)");
}

}  // namespace

extern "C" {

int emb_shard_create(emb_engine *e, emb_comm *comm, const emb_shard_config *cfg, emb_shard **out) {
    if (!e || !cfg || !out || !cfg->tables) return fail(EMB_ERR_INVALID, "emb_shard_create: NULL argument");
    *out = nullptr;
    if (cfg->n_tables == 0 || cfg->dim == 0 || cfg->dim % 4) return fail(EMB_ERR_INVALID, "emb_shard_create: n_tables > 0 and dim a multiple of 4");
    if (cfg->depth > 3) return fail(EMB_ERR_INVALID, "emb_shard_create: depth is 0, 1, 2 or 3");
    emb_shard *s = new (std::nothrow) emb_shard();
    if (!s) return fail(EMB_ERR_NOMEM, "out of host memory");
    s->e = e;
    s->comm = comm;
    int32_t dev = 0, rank = 0, world = 1;
    int rc = emb_device_of(e, &dev);
    s->peer_mode = (cfg->flags & EMB_SHARD_PEER_STORES) != 0;
    if (s->peer_mode) {
        if (!cfg->peer) {
            delete s;
            return fail(EMB_ERR_INVALID, "emb_shard_create: EMB_SHARD_PEER_STORES needs emb_shard_config.peer (emb_peer_create)");
        }
        s->peer = cfg->peer;
        s->peer_tag = pimemb::peer_next_epoch(cfg->peer) << 40;
        rank = pimemb::peer_rank(cfg->peer);
        world = pimemb::peer_world(cfg->peer);
        if (cfg->flags & EMB_SHARD_SELF_VIA_COMM) {
            delete s;
            return fail(EMB_ERR_INVALID, "emb_shard_create: EMB_SHARD_SELF_VIA_COMM and EMB_SHARD_PEER_STORES exclude each other");
        }
    } else if (rc == EMB_OK && comm) rc = emb_comm_rank(comm, &rank, &world);
    if (rc) {
        delete s;
        return rc;
    }
    s->device = dev;
    s->rank = rank;
    s->N = world;
    s->T = cfg->n_tables;
    s->dim = cfg->dim;
    s->depth = cfg->depth;
    s->flags = cfg->flags;
    s->self_via_comm = (cfg->flags & EMB_SHARD_SELF_VIA_COMM) != 0;
    s->check_served = (cfg->flags & EMB_SHARD_CHECK_SERVED) != 0;
    s->defer_report = s->check_served && (cfg->flags & EMB_SHARD_DEFER_REPORT) != 0;
    if (s->self_via_comm && !comm) {
        delete s;
        return fail(EMB_ERR_INVALID, "emb_shard_create: EMB_SHARD_SELF_VIA_COMM needs a communicator");
    }
    if (const char *t = getenv("PIMEMB_SHARD_TIMEOUT_S")) s->timeout_s = atof(t) > 0 ? atof(t) : s->timeout_s;
    if (const char *t = getenv("PIMEMB_SHARD_DIRECT")) s->allow_direct = t[0] != '0';
    if (const char *t = getenv("PIMEMB_SHARD_DIRECT_MERGE")) s->merge_direct = t[0] != '0';
    if (const char *t = getenv("PIMEMB_SHARD_WIDEN")) s->widen = t[0] != '0';
    if (cfg->flags & EMB_SHARD_NO_DIRECT) s->allow_direct = false;
    s->tabs.assign(cfg->tables, cfg->tables + cfg->n_tables);
    s->whole_of.assign((size_t)world, {});
    s->elem_bytes.assign(cfg->n_tables, 0);
    auto bail = [&](int code) {
        delete s;
        return code;
    };
    for (uint32_t t = 0; t < s->T; t++) {
        const emb_shard_table &tb = s->tabs[t];
        bool held = false;
        if (tb.placement == EMB_PLACE_REPLICATED) {
            s->rep.push_back(t);
            held = true;
        } else if (tb.placement == EMB_PLACE_WHOLE) {
            if (tb.owner < 0 || tb.owner >= world) return bail(fail(EMB_ERR_INVALID, "emb_shard_create: table %u: owner %d of %d ranks", t, tb.owner, world));
            s->whole_of[(size_t)tb.owner].push_back(t);
            held = tb.owner == rank;
        } else if (tb.placement == EMB_PLACE_ROWS) {
            if (tb.rows_per_shard == 0) return bail(fail(EMB_ERR_INVALID, "emb_shard_create: table %u: rows_per_shard is 0", t));
            s->rows.push_back(t);
            held = true;
        } else {
            return bail(fail(EMB_ERR_INVALID, "emb_shard_create: table %u: placement %u", t, tb.placement));
        }
        if (held) {
            uint32_t d = 0;
            emb_dtype dt = EMB_F32;
            rc = emb_table_info(e, tb.engine_table, nullptr, nullptr, &d, &dt);
            if (rc) return bail(rc);
            if (d != s->dim) return bail(fail(EMB_ERR_INVALID, "emb_shard_create: table %u: engine table %u has dim %u, not %u", t, tb.engine_table, d, s->dim));
            s->elem_bytes[t] = dt == EMB_F16 ? 2u : 4u;
        }
    }
    // the ranged (direct-path / counted) launches run on the 16-byte lane-piece kernels only: rows of 16..1024 bytes in 16-byte steps
    for (uint32_t t = 0; t < s->T; t++)
        if (s->elem_bytes[t] && (((uint64_t)s->dim * s->elem_bytes[t]) % 16 != 0 || (uint64_t)s->dim * s->elem_bytes[t] > 1024)) s->allow_direct = false;
    if (s->rows.size() > pimemb::kRouteBagMaxTables) return bail(fail(EMB_ERR_UNSUPPORTED, "emb_shard_create: at most %u row-split tables", pimemb::kRouteBagMaxTables));
    if (world > 255) return bail(fail(EMB_ERR_UNSUPPORTED, "emb_shard_create: at most 255 ranks"));
    for (const auto &w : s->whole_of) s->max_whole = std::max<uint32_t>(s->max_whole, (uint32_t)w.size());
    if (s->peer_mode) {
        const size_t most = s->max_whole;
        // (a checked shard keeps the last 2 x (Kr + max_whole) words of a mailbox for the served counts)
        if (2 * (s->rows.size() + 1) + 2 + kPeerConstHead + most * kWholeWords + 4 * s->rows.size() + (s->check_served ? 2 * (s->rows.size() + most) : 0) > pimemb::kPeerMsgWords)
            return bail(fail(EMB_ERR_UNSUPPORTED, "emb_shard_create: %zu whole tables on one owner do not fit a mailbox message (%u words)", most, pimemb::kPeerMsgWords));
    }
    s->Kr = (uint32_t)s->rows.size();
    s->M = (uint32_t)s->whole_of[(size_t)rank].size();
    s->wc_off.assign((size_t)world + 1, 0);
    for (int p = 0; p < world; p++) s->wc_off[(size_t)p + 1] = s->wc_off[(size_t)p] + (uint32_t)s->whole_of[(size_t)p].size();
    s->Wtot = s->wc_off[(size_t)world];

    DeviceGuard g(dev);
    hipError_t err = hipStreamCreateWithFlags(&s->s_comm, hipStreamNonBlocking);
    if (err == hipSuccess) err = hipEventCreateWithFlags(&s->ev_switch, hipEventDisableTiming);
    const size_t N = (size_t)world;
    const size_t counts_words = N * ((s->Kr + 1) * 4 + (size_t)s->M * kWholeWords);
    for (Batch &b : s->ring) {
        hipEvent_t *evs[5] = {&b.ev_routed, &b.ev_req, &b.ev_served, &b.ev_ret, &b.ev_out};
        for (hipEvent_t *ev : evs)
            if (err == hipSuccess) err = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        void *p = nullptr;
        if (err == hipSuccess) err = hipHostMalloc(&p, counts_words * 4 + 64, hipHostMallocMapped | hipHostMallocCoherent);
        b.counts_host = static_cast<uint32_t *>(p);
        p = nullptr;
        if (err == hipSuccess) err = hipHostMalloc(&p, (size_t)s->Wtot * kWholeWords * 4 + 64, hipHostMallocMapped | hipHostMallocCoherent);
        b.wc_host = static_cast<uint32_t *>(p);
        p = nullptr;
        if (err == hipSuccess && s->peer_mode) {
            err = hipHostMalloc(&p, (N * (kPeerConstHead + 4 * (size_t)s->Kr) + (size_t)s->Wtot * kWholeWords) * 4 + 64, hipHostMallocMapped | hipHostMallocCoherent);
            b.pc_host = static_cast<uint32_t *>(p);
            b.req_send.arena = b.ret_recv.arena = true;      // peers gather from / store into these two
        }
        p = nullptr;
        if (err == hipSuccess) err = hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent);
        b.flag = static_cast<unsigned long long *>(p);
        if (err == hipSuccess) {
            *b.flag = 0;
            err = hipMalloc(&b.counts_in.p, N * ((s->Kr + 1) * 2 + (size_t)s->M * kWholeWords) * 4 + 64);
        }
        if (err == hipSuccess) err = hipMemset(b.counts_in.p, 0, N * ((s->Kr + 1) * 2 + (size_t)s->M * kWholeWords) * 4 + 64);
        if (err == hipSuccess) err = hipMalloc(&b.wc_send.p, (size_t)s->Wtot * kWholeWords * 4 + 64);
        if (s->check_served) {       // served-bag counters (zero between uses: the publish kernel reads them with an exchange)
            const size_t ctr_bytes = n_counters(s) * EMB_SERVED_BYTES + 256;
            if (err == hipSuccess) err = hipMalloc(&b.chk_ctr.p, ctr_bytes);
            if (err == hipSuccess) err = hipMemset(b.chk_ctr.p, 0, ctr_bytes);
            p = nullptr;
            const size_t n_entries = s->rep.size() + s->Kr + (size_t)s->M;
            if (err == hipSuccess) err = hipHostMalloc(&p, n_entries * 8 + 64, hipHostMallocMapped | hipHostMallocCoherent);
            b.chk_host = static_cast<unsigned long long *>(p);
            if (b.chk_host) memset(b.chk_host, 0, n_entries * 8 + 64);           // (tag 0 is no batch's: served_tag)
        }
    }
    if (err != hipSuccess) {
        const int code = fail(err == hipErrorOutOfMemory ? EMB_ERR_NOMEM : EMB_ERR_DEVICE, "emb_shard_create: %s", hipGetErrorString(err));
        (void)emb_shard_destroy(s);
        return code;
    }
    *out = s;
    return EMB_OK;
}

int emb_shard_submit(emb_shard *s, const emb_shard_input *in, uint64_t n_bags, void *stream, uint64_t *seq) {
    if (!s || (!in && n_bags)) return fail(EMB_ERR_INVALID, "emb_shard_submit: NULL argument");
    if (n_bags >= (1ull << 32)) return fail(EMB_ERR_UNSUPPORTED, "emb_shard_submit: more than 2^32-1 bags");
    const double t0 = now_us();
    DeviceGuard g(s->device);
    Batch &b = s->ring[s->next_seq % kRing];
    if (b.stage != FREE && b.stage != DONE) return fail(EMB_ERR_INVALID, "emb_shard_submit: internal: slot of batch %llu is still in stage %d", (unsigned long long)b.seq, (int)b.stage);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (s->cs_known && st != s->cs) {      // the caller moved to another stream: everything queued so far comes first
        HIP_TRY(hipEventRecord(s->ev_switch, s->cs));
        HIP_TRY(hipStreamWaitEvent(st, s->ev_switch, 0));
    }
    s->cs = st;
    s->cs_known = true;
    harvest(s, b);
    // a deferred report: what the served counts of the batches completed so far say -- before this one's slot is recycled
    const int rc_prev = s->defer_report ? collect_pending(s, /*wait=*/false, ~0ull, /*must=*/b.seq) : EMB_OK;
    if (rc_prev != EMB_OK && rc_prev != EMB_ERR_RANGE) return rc_prev;
    b.in.assign(s->T, emb_shard_input{});
    uint32_t itype = (in && s->T) ? in[0].index_type : (uint32_t)EMB_IDX_U32;
    if (itype != EMB_IDX_U32 && itype != EMB_IDX_I64) return fail(EMB_ERR_INVALID, "emb_shard_submit: table 0: index_type %u (EMB_IDX_U32 or EMB_IDX_I64)", itype);
    for (uint32_t t = 0; t < s->T && in; t++) {
        const emb_shard_input &u = in[t];
        if (u.index_type != itype)
            return fail(EMB_ERR_INVALID, "emb_shard_submit: table %u: index_type %u, table 0 has %u (one width per batch)", t, u.index_type, itype);
        if (n_bags) {
            if (!u.pooled) return fail(EMB_ERR_INVALID, "emb_shard_submit: table %u: pooled is NULL", t);
            if (u.n_indices && !u.indices) return fail(EMB_ERR_INVALID, "emb_shard_submit: table %u: indices is NULL", t);
            if (u.n_indices >= (1ull << 32)) return fail(EMB_ERR_UNSUPPORTED, "emb_shard_submit: table %u: more than 2^32-1 indices", t);
            if (!u.offsets && (uint64_t)u.fixed_pooling * n_bags != u.n_indices)
                return fail(EMB_ERR_INVALID, "emb_shard_submit: table %u: offsets is NULL and fixed_pooling*n_bags != n_indices", t);
        }
        b.in[t] = u;
        if (!n_bags) b.in[t].n_indices = 0;
    }
    b.seq = s->next_seq;
    b.n_bags = n_bags;
    b.itype = itype;
    b.stage = FREE;
    int rc = stage_route(s, b);
    if (rc) return rc;
    s->next_seq++;
    if (seq) *seq = b.seq;
    rc = advance(s, kLag[s->depth], false);
    if (!s->lazy_segs.empty() && ++s->lazy_age >= 2 && (rc == EMB_OK || rc == EMB_ERR_RANGE)) {     // one publish kernel per two batches
        const int rc2 = publish_lazy(s);
        if (rc2 != EMB_OK) rc = rc2;
    }
    s->st.us_host_submit += now_us() - t0;
    if (rc == EMB_OK) rc = rc_prev;
    return rc == EMB_ERR_RANGE ? range_error(s) : rc;
}

}  // extern "C"

namespace {
// collect: also compare the served counts whose report was deferred (the caller's own flush does; the one inside
// emb_shard_lookup does not -- that is the round trip EMB_SHARD_DEFER_REPORT exists to take out of the synchronous call)
int flush_impl(emb_shard *s, bool collect) {
    if (!s->cs_known) return EMB_OK;
    const double t0 = now_us();
    DeviceGuard g(s->device);
    int rc = advance(s, kLag[0], true);     // requests of all, then lookups + returns of all, then un-routing: same order on every rank
    if (collect && s->defer_report && (rc == EMB_OK || rc == EMB_ERR_RANGE)) {
        const int rc2 = collect_pending(s);
        if (rc2 != EMB_OK) rc = (rc2 == EMB_ERR_RANGE && rc != EMB_OK) ? rc : rc2;
    }
    s->st.us_host_submit += now_us() - t0;
    return rc == EMB_ERR_RANGE ? range_error(s) : rc;
}

int wait_impl(emb_shard *s, uint64_t seq, void *stream, bool collect) {
    if (seq >= s->next_seq) return fail(EMB_ERR_INVALID, "emb_shard_wait: batch %llu was never submitted", (unsigned long long)seq);
    Batch &b = s->ring[seq % kRing];
    if (b.seq != seq) return fail(EMB_ERR_INVALID, "emb_shard_wait: batch %llu is no longer tracked (wait within %d submits)", (unsigned long long)seq, kRing);
    if (b.stage != DONE)
        return fail(EMB_ERR_INVALID, "emb_shard_wait: batch %llu has not been through all its stages yet (submit %u more batch(es) or call emb_shard_flush)",
                    (unsigned long long)seq, s->depth);
    hipStream_t other = static_cast<hipStream_t>(stream);
    DeviceGuard g(s->device);
    if (other != s->cs) {                       // (the submit stream itself: the batch's last kernel is already queued there)
        if (!b.out_recorded) {                  // (recorded now: behind the batch's last kernel, and possibly a little more)
            HIP_TRY(hipEventRecord(b.ev_out, s->cs));
            b.out_recorded = true;
        }
        HIP_TRY(hipStreamWaitEvent(other, b.ev_out, 0));
    }
    if (collect && b.check_pending) {           // a deferred report: the caller is about to consume this batch -- say what its counts said IF they
        // have arrived.  Only a look: emb_shard_wait orders a stream, it must not make the HOST wait for the GPU (the first version
        // did, and a pipelined loop that calls wait(b) every step ran behind its own publish kernels: 63.9 us per step against 61.0
        // with the comparison inside the call, profiles/r06/dist_world1_wait_blocks.md) -- a finding not yet visible surfaces at a
        // later submit, the caller's flush, or emb_shard_report
        const int rc = collect_pending(s, /*wait=*/false, seq);
        return rc == EMB_ERR_RANGE ? range_error(s) : rc;
    }
    return EMB_OK;
}
}  // namespace

extern "C" {

int emb_shard_flush(emb_shard *s) {
    if (!s) return fail(EMB_ERR_INVALID, "emb_shard_flush: shard is NULL");
    return flush_impl(s, /*collect=*/true);
}

int emb_shard_report(emb_shard *s) {
    if (!s) return fail(EMB_ERR_INVALID, "emb_shard_report: shard is NULL");
    if (!s->defer_report) return EMB_OK;
    DeviceGuard g(s->device);
    const int rc = collect_pending(s);
    return rc == EMB_ERR_RANGE ? range_error(s) : rc;
}

int emb_shard_wait(emb_shard *s, uint64_t seq, void *stream) {
    if (!s) return fail(EMB_ERR_INVALID, "emb_shard_wait: shard is NULL");
    return wait_impl(s, seq, stream, /*collect=*/true);
}

int emb_shard_lookup(emb_shard *s, const emb_shard_input *in, uint64_t n_bags, void *stream) {
    uint64_t seq = 0;
    int rc = emb_shard_submit(s, in, n_bags, stream, &seq);
    const int rc2 = (rc == EMB_OK || rc == EMB_ERR_RANGE) ? flush_impl(s, /*collect=*/false) : rc;
    if (rc2 != EMB_OK && rc2 != EMB_ERR_RANGE) return rc2;
    const int rc3 = wait_impl(s, seq, stream, /*collect=*/false);
    if (rc3) return rc3;
    return rc != EMB_OK ? rc : rc2;
}

int emb_shard_get_stats(emb_shard *s, emb_shard_stats *out, int reset) {
    if (!s || !out) return fail(EMB_ERR_INVALID, "emb_shard_get_stats: NULL argument");
    DeviceGuard g(s->device);          // (harvest waits for / reads timing events of the shard's GPU)
    for (Batch &b : s->ring)
        if (b.stage == DONE) harvest(s, b);
    *out = s->st;
    if (reset) s->st = emb_shard_stats{};
    return EMB_OK;
}

int emb_shard_set_kernel_timing(emb_shard *s, int on) {
    if (!s) return fail(EMB_ERR_INVALID, "emb_shard_set_kernel_timing: shard is NULL");
    s->kernel_timing = on != 0;
    return EMB_OK;
}

int emb_shard_sent_counts(emb_shard *s, uint64_t seq, uint32_t *counts, uint32_t capacity_words) {
    if (!s || !counts) return fail(EMB_ERR_INVALID, "emb_shard_sent_counts: NULL argument");
    Batch &b = s->ring[seq % kRing];
    if (seq >= s->next_seq || b.seq != seq || b.stage < REQUESTED)
        return fail(EMB_ERR_INVALID, "emb_shard_sent_counts: batch %llu is not tracked or its counts have not been read yet", (unsigned long long)seq);
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr;
    if (capacity_words < N * Kr * 2) return fail(EMB_ERR_INVALID, "emb_shard_sent_counts: need %u words", N * Kr * 2);
    for (uint32_t p = 0; p < N; p++)
        for (uint32_t k = 0; k < Kr; k++) {
            counts[(p * Kr + k) * 2] = b.counts_host[((size_t)p * (Kr + 1) + k) * 2];
            counts[(p * Kr + k) * 2 + 1] = b.counts_host[((size_t)p * (Kr + 1) + k) * 2 + 1];
        }
    return EMB_OK;
}

int emb_shard_destroy(emb_shard *s) {
    if (!s) return EMB_OK;
    DeviceGuard g(s->device);
    if (g_hp.on && s->next_seq) {
        fprintf(stderr, "[pimemb shard host profile] %llu batches, us per batch:", (unsigned long long)s->next_seq);
        for (int k = 0; k < 14; k++) fprintf(stderr, "  %s %.1f", g_hp.name[k], g_hp.acc[k] / (double)s->next_seq);
        fprintf(stderr, "\n");
        for (double &a : g_hp.acc) a = 0;
    }
    if (s->cs_known) (void)hipStreamSynchronize(s->cs);
    if (s->s_comm) (void)hipStreamSynchronize(s->s_comm);
    (void)hipGetLastError();
    if (s->defer_report && s->cs_known) {
        (void)publish_lazy(s);
        (void)hipStreamSynchronize(s->cs);
    }
    if (s->defer_report && collect_pending(s) == EMB_ERR_RANGE)      // a finding nobody collected: never lost silently
        fprintf(stderr, "[pimemb] emb_shard_destroy: an uncollected report: %s\n", s->range_msg[0] ? s->range_msg : "a batch named rows no rank holds");
    for (emb_shard::CachedPlan &c : s->plans)
        if (c.plan) (void)emb_plan_destroy(c.plan);
    for (emb_plan *p : s->retired) (void)emb_plan_destroy(p);
    for (Batch &b : s->ring) {
        DevBuf *bufs[10] = {&b.req_send, &b.meta, &b.slotmap, &b.counts_in, &b.wc_send, &b.req_recv, &b.ret_send, &b.ret_recv, &b.chk_ctr, &b.wide};
        for (DevBuf *d : bufs)
            if (d->p && !d->arena) (void)hipFree(d->p);
        if (b.chk_host) (void)hipHostFree(b.chk_host);
        if (b.pc_host) (void)hipHostFree(b.pc_host);
        if (b.counts_host) (void)hipHostFree(b.counts_host);
        if (b.wc_host) (void)hipHostFree(b.wc_host);
        if (b.flag) (void)hipHostFree(b.flag);
        hipEvent_t evs[5] = {b.ev_routed, b.ev_req, b.ev_served, b.ev_ret, b.ev_out};
        for (hipEvent_t ev : evs)
            if (ev) (void)hipEventDestroy(ev);
        for (hipEvent_t ev : b.tev)
            if (ev) (void)hipEventDestroy(ev);
    }
    if (s->work.p) (void)hipFree(s->work.p);
    if (s->ev_switch) (void)hipEventDestroy(s->ev_switch);
    if (s->s_comm) (void)hipStreamDestroy(s->s_comm);
    delete s;
    return EMB_OK;
}

}  // extern "C"
