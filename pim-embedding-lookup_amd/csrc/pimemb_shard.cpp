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
// software-pipelined over consecutive batches (depth 0..2).  Everything between two kernels is an RCCL group issued from
// here (emb_comm_exchange) or nothing at all: pieces a rank addresses to itself are served in place, whole tables travel
// straight out of / into the caller's buffers.  The only host wait of a batch is for the counts, which a small kernel
// drops into pinned memory behind a flag word (polled; no event, no copy engine).
//
// No lookup is computed here: S and L are emb_lookup_batched launches (pimemb_engine.cpp), R and U the routing kernels
// (pimemb_kernels.hip).  This file is streams, events, byte offsets and the order in which all ranks issue transfers.
#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "pimemb_internal.h"

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

constexpr int kRing = 4;            // batches in flight at most: depth 2 keeps three live, the fourth slot is being filled
constexpr uint32_t kWholeWords = 4; // words of a whole-table count entry: {bags, indices, fixed pooling (0 = offsets travel), 0}

inline uint64_t pad4(uint64_t v) { return (v + 3u) & ~(uint64_t)3u; }

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
};

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

enum Stage : int { FREE = 0, ROUTED = 1, REQUESTED = 2, SERVED = 3 };

struct Batch {
    uint64_t seq = ~0ull;
    Stage stage = FREE;
    uint64_t n_bags = 0;
    std::vector<emb_shard_input> in;
    // HBM, grow-only, owned by the slot
    DevBuf req_send, meta, slotmap, counts_in, wc_send, req_recv, ret_send, ret_recv;
    // pinned host
    uint32_t *wc_host = nullptr;        // whole-table counts this rank sends: [sum_p |whole_of[p]|][kWholeWords]
    uint32_t *counts_host = nullptr;    // [sent row counts N(Kr+1)2 | received row counts N(Kr+1)2 | received whole counts N*M*4]
    unsigned long long *flag = nullptr; // raised (= seq + 1) behind counts_host by publish_words_kernel
    hipEvent_t ev_in = nullptr, ev_routed = nullptr, ev_req = nullptr, ev_served = nullptr, ev_ret = nullptr,
               ev_local = nullptr, ev_out = nullptr;
    bool out_recorded = false, ret_recorded = false, local_recorded = false, counts_posted = false;
    bool prev_pending = false;          // ev_out still carries the record of the slot's PREVIOUS occupant (until this batch's U)
    // per-peer sizes of this batch (row-split path), from the counts
    std::vector<uint64_t> out_words, in_words, rows_back, rows_served;
    int deferred_rc = EMB_OK;
};

}  // namespace

struct emb_shard {
    emb_engine *e = nullptr;
    emb_comm *comm = nullptr;
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
    hipStream_t s_route = nullptr, s_comm = nullptr, s_comp = nullptr, s_un = nullptr;
    DevBuf work;                                     // scratch of one route call (ordered on s_route)
    Batch ring[kRing];
    uint64_t next_seq = 0;
    double timeout_s = 60.0;
    emb_shard_stats st{};
    std::vector<emb_comm_op> ops;                    // scratch
    std::vector<emb_lookup_desc> descs;              // scratch
};

namespace {

int ensure(emb_shard *s, Batch &b, DevBuf &buf, size_t bytes) {
    if (buf.cap >= bytes && buf.p) return EMB_OK;
    // the slot's previous occupant may still be running on the device: its last reader is behind ev_out
    if (b.prev_pending || b.out_recorded) HIP_TRY(hipEventSynchronize(b.ev_out));
    b.prev_pending = false;
    if (buf.p) HIP_TRY(hipFree(buf.p));
    buf.p = nullptr;
    buf.cap = 0;
    const size_t cap = bytes + bytes / 4 + 256;
    HIP_TRY(hipMalloc(&buf.p, cap));
    buf.cap = cap;
    (void)s;
    return EMB_OK;
}

// whether peer p's pieces travel through RCCL (false: p is this rank and is served in place)
inline bool via_comm(const emb_shard *s, int p) { return p != s->rank || s->self_via_comm; }

int exchange(emb_shard *s, hipStream_t st) {
    if (s->ops.empty()) return EMB_OK;
    if (!s->comm) return fail(EMB_ERR_INVALID, "emb_shard: a transfer to a peer without a communicator");
    return emb_comm_exchange(s->comm, s->ops.data(), (uint32_t)s->ops.size(), st);
}

inline void add_op(emb_shard *s, int peer, bool recv, const void *ptr, uint64_t bytes) {
    if (bytes) s->ops.push_back(emb_comm_op{peer, recv ? 1 : 0, const_cast<void *>(ptr), bytes});
}

inline uint32_t *sent_counts(const emb_shard *s, const Batch &b) { (void)s; return b.counts_host; }
inline uint32_t *recv_counts(const emb_shard *s, const Batch &b) { return b.counts_host + (size_t)s->N * (s->Kr + 1) * 2; }
inline uint32_t *recv_whole(const emb_shard *s, const Batch &b) { return b.counts_host + (size_t)s->N * (s->Kr + 1) * 4; }

// ---- R(b) + counts out + L(b) -------------------------------------------------------------------------------------
int stage_route(emb_shard *s, Batch &b, hipStream_t caller) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr, M = s->M;
    HIP_TRY(hipEventRecord(b.ev_in, caller));
    HIP_TRY(hipStreamWaitEvent(s->s_route, b.ev_in, 0));
    if (b.out_recorded && hipEventQuery(b.ev_out) != hipSuccess) {   // the slot's previous occupant (four batches ago): normally long done
        HIP_TRY(hipStreamWaitEvent(s->s_route, b.ev_out, 0));
        HIP_TRY(hipStreamWaitEvent(s->s_comm, b.ev_out, 0));
        HIP_TRY(hipStreamWaitEvent(s->s_comp, b.ev_out, 0));
    }
    (void)hipGetLastError();
    b.prev_pending = b.out_recorded;
    b.out_recorded = b.ret_recorded = b.local_recorded = false;
    b.deferred_rc = EMB_OK;

    // row-split tables: cut every bag into per-shard sub-bags; the counts sit at the head of `meta`
    uint64_t total_idx = 0;
    for (uint32_t k = 0; k < Kr; k++) total_idx += b.in[s->rows[k]].n_indices;
    if (Kr) {
        uint64_t sb = 0, mb = 0, lb = 0, wb = 0;
        EMB_TRY(emb_route_bags_sizes(Kr, std::max<uint64_t>(b.n_bags, 1), total_idx, N, &sb, &mb, &lb, &wb));
        EMB_TRY(ensure(s, b, b.req_send, sb));
        EMB_TRY(ensure(s, b, b.meta, mb));
        EMB_TRY(ensure(s, b, b.slotmap, lb));
        if (s->work.cap < wb) {
            HIP_TRY(hipStreamSynchronize(s->s_route));      // the previous route call's scratch
            if (s->work.p) HIP_TRY(hipFree(s->work.p));
            s->work.p = nullptr;
            s->work.cap = 0;
            HIP_TRY(hipMalloc(&s->work.p, wb + wb / 4));
            s->work.cap = wb + wb / 4;
        }
        if (b.n_bags) {
            emb_route_table rt[pimemb::kRouteBagMaxTables];
            for (uint32_t k = 0; k < Kr; k++) {
                const emb_shard_input &u = b.in[s->rows[k]];
                rt[k] = emb_route_table{u.indices, u.offsets, u.n_indices, u.fixed_pooling, s->tabs[s->rows[k]].rows_per_shard};
            }
            EMB_TRY(emb_route_bags(s->e, rt, Kr, b.n_bags, N, b.req_send.p, static_cast<uint32_t *>(b.meta.p),
                                   static_cast<uint32_t *>(b.slotmap.p), s->work.p, s->s_route));
        } else {         // nothing to ask for: all counts (and peaks) zero; this rank still serves
            HIP_TRY(pimemb::launch_zero_words(static_cast<uint32_t *>(b.meta.p), pimemb::route_meta_counts_words(Kr, N), s->s_route));
        }
    }
    // whole tables: what this rank asks each owner for is known on the host
    if (s->Wtot) {
        for (uint32_t p = 0, w = 0; p < N; p++)
            for (uint32_t t : s->whole_of[p]) {
                const emb_shard_input &u = b.in[t];
                uint32_t *c = b.wc_host + (size_t)w++ * kWholeWords;
                c[0] = (uint32_t)b.n_bags;
                c[1] = (uint32_t)u.n_indices;
                c[2] = u.offsets ? 0u : u.fixed_pooling;
                c[3] = 0;
            }
        bool any_remote = false;
        for (uint32_t p = 0; p < N; p++) any_remote |= via_comm(s, (int)p) && !s->whole_of[p].empty();
        if (any_remote)
            HIP_TRY(hipMemcpyAsync(b.wc_send.p, b.wc_host, (size_t)s->Wtot * kWholeWords * 4, hipMemcpyHostToDevice, s->s_route));
    }
    HIP_TRY(hipEventRecord(b.ev_routed, s->s_route));
    HIP_TRY(hipStreamWaitEvent(s->s_comm, b.ev_routed, 0));

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
    for (const emb_comm_op &o : s->ops)
        if (!o.is_recv) (o.peer == s->rank ? s->st.bytes_to_self : s->st.bytes_to_peers) += o.bytes;
    EMB_TRY(exchange(s, s->s_comm));
    // ... and reach the host behind a flag word
    *b.flag = 0;
    b.counts_posted = false;
    if (Kr || !s->ops.empty()) {
        const uint32_t *src[3] = {Kr ? static_cast<const uint32_t *>(b.meta.p) : nullptr,
                                  Kr ? static_cast<const uint32_t *>(b.counts_in.p) : nullptr,
                                  M ? static_cast<const uint32_t *>(b.counts_in.p) + (size_t)N * (Kr + 1) * 2 : nullptr};
        const uint32_t n[3] = {N * (Kr + 1) * 2, N * (Kr + 1) * 2, N * M * kWholeWords};
        HIP_TRY(pimemb::launch_publish_words(src, n, b.counts_host, b.flag, b.seq + 1, s->s_comm));
        b.counts_posted = true;
    }

    // L(b): replicated tables, this rank's own bags
    if (!s->rep.empty() && b.n_bags) {
        HIP_TRY(hipStreamWaitEvent(s->s_comp, b.ev_in, 0));
        s->descs.clear();
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
            s->descs.push_back(d);
            s->st.local_algorithmic_bytes += u.n_indices * ((uint64_t)s->dim * s->elem_bytes[t] + 4) +
                                             (u.offsets ? b.n_bags * 4 : 0) + b.n_bags * (uint64_t)s->dim * 4;
        }
        if (s->check_served) {      // the caller's own ids against the replicated tables
            uint64_t bad = 0;
            int rc = emb_lookup_batched_checked(s->e, s->descs.data(), (uint32_t)s->descs.size(), EMB_IDX_U32, EMB_MEM_DEVICE, s->s_comp, &bad);
            if (rc == EMB_ERR_RANGE) {
                for (const emb_lookup_desc &d : s->descs)
                    HIP_TRY(hipMemsetAsync(d.pooled, 0, d.n_bags * (size_t)s->dim * 4, s->s_comp));
                b.deferred_rc = EMB_ERR_RANGE;
            } else if (rc) {
                return rc;
            }
        } else {
            EMB_TRY(emb_lookup_batched(s->e, s->descs.data(), (uint32_t)s->descs.size(), EMB_IDX_U32, EMB_MEM_DEVICE, s->s_comp));
        }
        HIP_TRY(hipEventRecord(b.ev_local, s->s_comp));
        b.local_recorded = true;
    }
    b.stage = ROUTED;
    return EMB_OK;
}

// ---- Q(b): the one host wait, then the request pieces ----------------------------------------------------------------
int stage_request(emb_shard *s, Batch &b) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr, M = s->M;
    if (b.counts_posted) {
        const double t0 = now_us();
        volatile unsigned long long *flag = b.flag;
        uint64_t spins = 0;
        while (*flag != b.seq + 1) {
            if ((++spins & 0xfffu) == 0 && now_us() - t0 > s->timeout_s * 1e6) {
                const hipError_t q = hipStreamQuery(s->s_comm);
                return fail(EMB_ERR_DEVICE, "emb_shard: the counts of batch %llu did not arrive within %.0f s (comm stream: %s) -- a peer is "
                            "missing, or the ranks did not make the same calls", (unsigned long long)b.seq, s->timeout_s,
                            hipGetErrorString(q));
            }
        }
        s->st.us_host_wait_counts += now_us() - t0;
    } else {
        memset(b.counts_host, 0, (size_t)N * ((Kr + 1) * 4 + M * kWholeWords) * 4);
    }
    uint32_t *sent = sent_counts(s, b), *recv = recv_counts(s, b), *rwhole = recv_whole(s, b);
    if (!via_comm(s, s->rank)) {       // what this rank asked ITSELF for never travelled
        if (Kr) memcpy(recv + (size_t)s->rank * (Kr + 1) * 2, sent + (size_t)s->rank * (Kr + 1) * 2, (size_t)(Kr + 1) * 8);
        if (M) memcpy(rwhole + (size_t)s->rank * M * kWholeWords, b.wc_host + (size_t)s->wc_off[s->rank] * kWholeWords, (size_t)M * kWholeWords * 4);
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
                whole_in_w += (c[2] ? 0 : pad4(c[0])) + pad4(c[1]);
                whole_rows += c[0];
            }
    }
    EMB_TRY(ensure(s, b, b.req_recv, (in_w + whole_in_w) * 4 + 16));
    EMB_TRY(ensure(s, b, b.ret_send, (served + whole_rows) * (uint64_t)s->dim * 4 + 16));
    EMB_TRY(ensure(s, b, b.ret_recv, back * (uint64_t)s->dim * 4 + 16));

    s->ops.clear();
    uint64_t out_at = 0, in_at = 0, win_at = in_w;
    for (uint32_t p = 0; p < N; p++) {
        if (via_comm(s, (int)p)) {
            add_op(s, (int)p, false, static_cast<uint32_t *>(b.req_send.p) + out_at, b.out_words[p] * 4);
            add_op(s, (int)p, true, static_cast<uint32_t *>(b.req_recv.p) + in_at, b.in_words[p] * 4);
            for (uint32_t t : s->whole_of[p]) {          // straight out of the caller's buffers
                const emb_shard_input &u = b.in[t];
                if (u.offsets) add_op(s, (int)p, false, u.offsets, b.n_bags * 4);
                add_op(s, (int)p, false, u.indices, u.n_indices * 4);
            }
            for (uint32_t j = 0; j < M; j++) {
                const uint32_t *c = rwhole + ((size_t)p * M + j) * kWholeWords;
                if (!c[2]) {
                    add_op(s, (int)p, true, static_cast<uint32_t *>(b.req_recv.p) + win_at, (uint64_t)c[0] * 4);
                    win_at += pad4(c[0]);
                }
                add_op(s, (int)p, true, static_cast<uint32_t *>(b.req_recv.p) + win_at, (uint64_t)c[1] * 4);
                win_at += pad4(c[1]);
            }
        }
        out_at += b.out_words[p];
        in_at += b.in_words[p];
    }
    for (const emb_comm_op &o : s->ops)
        if (!o.is_recv) (o.peer == s->rank ? s->st.bytes_to_self : s->st.bytes_to_peers) += o.bytes;
    EMB_TRY(exchange(s, s->s_comm));
    HIP_TRY(hipEventRecord(b.ev_req, s->s_comm));
    b.stage = REQUESTED;
    return EMB_OK;
}

// ---- S(b), T(b), U(b) --------------------------------------------------------------------------------------------------
int stage_serve(emb_shard *s, Batch &b) {
    const uint32_t N = (uint32_t)s->N, Kr = s->Kr, M = s->M, dim = s->dim;
    uint32_t *recv = recv_counts(s, b), *rwhole = recv_whole(s, b);
    HIP_TRY(hipStreamWaitEvent(s->s_comp, b.ev_req, 0));      // the pieces have arrived (and, with it, R(b) has run)
    uint64_t in_w = 0, served = 0;
    for (uint32_t p = 0; p < N; p++) {
        in_w += b.in_words[p];
        served += b.rows_served[p];
    }
    s->descs.clear();
    uint64_t alg = 0, n_sub = 0, n_idx = 0;
    // row pieces: source s asked for sub-bags of my shard of table k -- an ordinary lookup each
    uint64_t in_at = 0, out_at = 0, served_at = 0, back_at = 0;
    for (uint32_t p = 0; p < N; p++) {
        const bool remote = via_comm(s, (int)p);
        const uint32_t *words = remote ? static_cast<uint32_t *>(b.req_recv.p) + in_at : static_cast<uint32_t *>(b.req_send.p) + out_at;
        // rows for a remote source go to ret_send (T sends them); my own go where the un-router reads shard `rank`'s rows
        float *rows_dst = remote ? static_cast<float *>(b.ret_send.p) + served_at * dim : static_cast<float *>(b.ret_recv.p) + back_at * dim;
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
                s->descs.push_back(d);
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
    // whole tables owned here: source p's bags exactly as p's caller passed them
    uint64_t win_at = in_w, wrow_at = served;
    for (uint32_t p = 0; p < N; p++) {
        const bool remote = via_comm(s, (int)p);
        for (uint32_t j = 0; j < M; j++) {
            const uint32_t t = s->whole_of[s->rank][j];
            const uint32_t *c = rwhole + ((size_t)p * M + j) * kWholeWords;
            const uint64_t nb = c[0], ni = c[1];
            emb_lookup_desc d{};
            d.table_id = s->tabs[t].engine_table;
            d.n_indices = ni;
            d.n_bags = nb;
            if (remote) {
                d.fixed_pooling = c[2];
                if (!c[2]) {
                    d.offsets = static_cast<uint32_t *>(b.req_recv.p) + win_at;
                    win_at += pad4(nb);
                }
                d.indices = static_cast<uint32_t *>(b.req_recv.p) + win_at;
                win_at += pad4(ni);
                d.pooled = static_cast<float *>(b.ret_send.p) + wrow_at * dim;
                wrow_at += nb;
            } else {           // my own bags of my own table: in place, straight into the caller's buffer
                const emb_shard_input &u = b.in[t];
                d.fixed_pooling = u.offsets ? 0u : u.fixed_pooling;
                d.offsets = u.offsets;
                d.indices = u.indices;
                d.pooled = u.pooled;
            }
            if (nb) {
                s->descs.push_back(d);
                alg += ni * ((uint64_t)dim * s->elem_bytes[t] + 4) + (d.offsets ? nb * 4 : 0) + nb * (uint64_t)dim * 4;
                n_sub += nb;
                n_idx += ni;
            }
        }
    }
    if (!s->descs.empty()) {
        int rc;
        if (s->check_served) {
            uint64_t bad = 0;
            rc = emb_lookup_batched_checked(s->e, s->descs.data(), (uint32_t)s->descs.size(), EMB_IDX_U32, EMB_MEM_DEVICE, s->s_comp, &bad);
            if (rc == EMB_ERR_RANGE) {          // nothing was gathered: the pieces pool to zero rows, the batch still completes
                for (const emb_lookup_desc &d : s->descs)
                    HIP_TRY(hipMemsetAsync(d.pooled, 0, d.n_bags * (size_t)dim * 4, s->s_comp));
                b.deferred_rc = EMB_ERR_RANGE;
                rc = EMB_OK;
            }
        } else {
            rc = emb_lookup_batched(s->e, s->descs.data(), (uint32_t)s->descs.size(), EMB_IDX_U32, EMB_MEM_DEVICE, s->s_comp);
        }
        if (rc) return rc;
    }
    s->st.served_algorithmic_bytes += alg;
    s->st.served_sub_bags += n_sub;
    s->st.served_indices += n_idx;
    HIP_TRY(hipEventRecord(b.ev_served, s->s_comp));

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
        for (const emb_comm_op &o : s->ops)
            if (!o.is_recv) (o.peer == s->rank ? s->st.bytes_to_self : s->st.bytes_to_peers) += o.bytes;
        HIP_TRY(hipStreamWaitEvent(s->s_comm, b.ev_served, 0));
        EMB_TRY(exchange(s, s->s_comm));
        HIP_TRY(hipEventRecord(b.ev_ret, s->s_comm));
        b.ret_recorded = true;
    }

    // U(b): on its own stream, so that neither the next lookup nor the next transfer queues behind this batch's return
    HIP_TRY(hipStreamWaitEvent(s->s_un, b.ev_served, 0));
    if (b.ret_recorded) HIP_TRY(hipStreamWaitEvent(s->s_un, b.ev_ret, 0));
    if (b.local_recorded) HIP_TRY(hipStreamWaitEvent(s->s_un, b.ev_local, 0));
    if (Kr && b.n_bags) {
        float *outs[pimemb::kRouteBagMaxTables];
        for (uint32_t k = 0; k < Kr; k++) outs[k] = b.in[s->rows[k]].pooled;
        HIP_TRY(pimemb::launch_unroute_bags_to(static_cast<float *>(b.ret_recv.p), static_cast<uint32_t *>(b.meta.p),
                                               static_cast<uint32_t *>(b.slotmap.p), Kr, b.n_bags, N, dim, outs, s->s_un));
    }
    HIP_TRY(hipEventRecord(b.ev_out, s->s_un));
    b.out_recorded = true;
    b.prev_pending = false;
    b.stage = SERVED;
    s->st.n_batches++;
    return EMB_OK;
}

// Advance every live batch that is at least (d_req, d_serve) submits old, oldest first -- the same order on every rank.
int advance(emb_shard *s, uint64_t d_req, uint64_t d_serve) {
    int deferred = EMB_OK;
    const uint64_t newest = s->next_seq;        // one past the last submitted
    const uint64_t first = newest > kRing ? newest - kRing : 0;
    // requests of younger batches go out BEFORE the return transfer of older ones (see the header: Q(n-1) ahead of T(n-2) on
    // the comm stream, so the next lookup never waits out a return)
    for (uint64_t q = first; q < newest; q++) {
        Batch &b = s->ring[q % kRing];
        if (b.seq == q && b.stage == ROUTED && q + d_req < newest) EMB_TRY(stage_request(s, b));
    }
    for (uint64_t q = first; q < newest; q++) {
        Batch &b = s->ring[q % kRing];
        if (b.seq == q && b.stage == REQUESTED && q + d_serve < newest) {
            EMB_TRY(stage_serve(s, b));
            if (b.deferred_rc != EMB_OK) deferred = b.deferred_rc;
        }
    }
    return deferred;
}

}  // namespace

extern "C" {

int emb_shard_create(emb_engine *e, emb_comm *comm, const emb_shard_config *cfg, emb_shard **out) {
    if (!e || !cfg || !out || !cfg->tables) return fail(EMB_ERR_INVALID, "emb_shard_create: NULL argument");
    *out = nullptr;
    if (cfg->n_tables == 0 || cfg->dim == 0 || cfg->dim % 4) return fail(EMB_ERR_INVALID, "emb_shard_create: n_tables > 0 and dim a multiple of 4");
    if (cfg->depth > 2) return fail(EMB_ERR_INVALID, "emb_shard_create: depth is 0, 1 or 2");
    emb_shard *s = new (std::nothrow) emb_shard();
    if (!s) return fail(EMB_ERR_NOMEM, "out of host memory");
    s->e = e;
    s->comm = comm;
    int32_t dev = 0, rank = 0, world = 1;
    int rc = emb_device_of(e, &dev);
    if (rc == EMB_OK && comm) rc = emb_comm_rank(comm, &rank, &world);
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
    if (s->self_via_comm && !comm) {
        delete s;
        return fail(EMB_ERR_INVALID, "emb_shard_create: EMB_SHARD_SELF_VIA_COMM needs a communicator");
    }
    if (const char *t = getenv("PIMEMB_SHARD_TIMEOUT_S")) s->timeout_s = atof(t) > 0 ? atof(t) : s->timeout_s;
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
    if (s->rows.size() > pimemb::kRouteBagMaxTables) return bail(fail(EMB_ERR_UNSUPPORTED, "emb_shard_create: at most %u row-split tables", pimemb::kRouteBagMaxTables));
    if (world > 255) return bail(fail(EMB_ERR_UNSUPPORTED, "emb_shard_create: at most 255 ranks"));
    s->Kr = (uint32_t)s->rows.size();
    s->M = (uint32_t)s->whole_of[(size_t)rank].size();
    s->wc_off.assign((size_t)world + 1, 0);
    for (int p = 0; p < world; p++) s->wc_off[(size_t)p + 1] = s->wc_off[(size_t)p] + (uint32_t)s->whole_of[(size_t)p].size();
    s->Wtot = s->wc_off[(size_t)world];

    DeviceGuard g(dev);
    hipError_t err = hipSuccess;
    hipStream_t *streams[4] = {&s->s_route, &s->s_comm, &s->s_comp, &s->s_un};
    for (hipStream_t *st : streams)
        if (err == hipSuccess) err = hipStreamCreateWithFlags(st, hipStreamNonBlocking);
    const size_t N = (size_t)world;
    const size_t counts_words = N * ((s->Kr + 1) * 4 + (size_t)s->M * kWholeWords);
    for (Batch &b : s->ring) {
        hipEvent_t *evs[7] = {&b.ev_in, &b.ev_routed, &b.ev_req, &b.ev_served, &b.ev_ret, &b.ev_local, &b.ev_out};
        for (hipEvent_t *ev : evs)
            if (err == hipSuccess) err = hipEventCreateWithFlags(ev, hipEventDisableTiming);
        void *p = nullptr;
        if (err == hipSuccess) err = hipHostMalloc(&p, counts_words * 4 + 64, hipHostMallocMapped | hipHostMallocCoherent);
        b.counts_host = static_cast<uint32_t *>(p);
        p = nullptr;
        if (err == hipSuccess) err = hipHostMalloc(&p, (size_t)s->Wtot * kWholeWords * 4 + 64, hipHostMallocMapped | hipHostMallocCoherent);
        b.wc_host = static_cast<uint32_t *>(p);
        p = nullptr;
        if (err == hipSuccess) err = hipHostMalloc(&p, 64, hipHostMallocMapped | hipHostMallocCoherent);
        b.flag = static_cast<unsigned long long *>(p);
        if (err == hipSuccess) {
            *b.flag = 0;
            err = hipMalloc(&b.counts_in.p, N * ((s->Kr + 1) * 2 + (size_t)s->M * kWholeWords) * 4 + 64);
        }
        if (err == hipSuccess) err = hipMemset(b.counts_in.p, 0, N * ((s->Kr + 1) * 2 + (size_t)s->M * kWholeWords) * 4 + 64);
        if (err == hipSuccess) err = hipMalloc(&b.wc_send.p, (size_t)s->Wtot * kWholeWords * 4 + 64);
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
    if (b.stage != FREE && b.stage != SERVED) return fail(EMB_ERR_INVALID, "emb_shard_submit: internal: slot of batch %llu is still in stage %d", (unsigned long long)b.seq, (int)b.stage);
    b.in.assign(s->T, emb_shard_input{});
    for (uint32_t t = 0; t < s->T && in; t++) {
        const emb_shard_input &u = in[t];
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
    b.stage = FREE;
    int rc = stage_route(s, b, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    s->next_seq++;
    if (seq) *seq = b.seq;
    static const uint64_t dq[3] = {0, 0, 1}, ds[3] = {0, 1, 2};
    rc = advance(s, dq[s->depth], ds[s->depth]);
    s->st.us_host_submit += now_us() - t0;
    if (rc == EMB_ERR_RANGE) return fail(EMB_ERR_RANGE, "emb_shard: a piece served by this rank named rows outside its table (pooled to zero rows)");
    return rc;
}

int emb_shard_flush(emb_shard *s) {
    if (!s) return fail(EMB_ERR_INVALID, "emb_shard_flush: shard is NULL");
    const double t0 = now_us();
    DeviceGuard g(s->device);
    // oldest first, each through all its remaining stages: the same order on every rank
    int deferred = EMB_OK;
    const uint64_t newest = s->next_seq, first = newest > kRing ? newest - kRing : 0;
    for (uint64_t q = first; q < newest; q++) {
        Batch &b = s->ring[q % kRing];
        if (b.seq != q) continue;
        if (b.stage == ROUTED) EMB_TRY(stage_request(s, b));
        if (b.stage == REQUESTED) {
            EMB_TRY(stage_serve(s, b));
            if (b.deferred_rc != EMB_OK) deferred = b.deferred_rc;
        }
    }
    s->st.us_host_submit += now_us() - t0;
    if (deferred == EMB_ERR_RANGE) return fail(EMB_ERR_RANGE, "emb_shard: a piece served by this rank named rows outside its table (pooled to zero rows)");
    return deferred;
}

int emb_shard_wait(emb_shard *s, uint64_t seq, void *stream) {
    if (!s) return fail(EMB_ERR_INVALID, "emb_shard_wait: shard is NULL");
    if (seq >= s->next_seq) return fail(EMB_ERR_INVALID, "emb_shard_wait: batch %llu was never submitted", (unsigned long long)seq);
    Batch &b = s->ring[seq % kRing];
    if (b.seq != seq) return fail(EMB_ERR_INVALID, "emb_shard_wait: batch %llu is no longer tracked (wait within %d submits)", (unsigned long long)seq, kRing);
    if (b.stage != SERVED)
        return fail(EMB_ERR_INVALID, "emb_shard_wait: batch %llu has not been through all its stages yet (submit %u more batch(es) or call emb_shard_flush)",
                    (unsigned long long)seq, s->depth);
    DeviceGuard g(s->device);
    HIP_TRY(hipStreamWaitEvent(static_cast<hipStream_t>(stream), b.ev_out, 0));
    return EMB_OK;
}

int emb_shard_lookup(emb_shard *s, const emb_shard_input *in, uint64_t n_bags, void *stream) {
    uint64_t seq = 0;
    int rc = emb_shard_submit(s, in, n_bags, stream, &seq);
    const int rc2 = (rc == EMB_OK || rc == EMB_ERR_RANGE) ? emb_shard_flush(s) : rc;
    if (rc2 != EMB_OK && rc2 != EMB_ERR_RANGE) return rc2;
    const int rc3 = emb_shard_wait(s, seq, stream);
    if (rc3) return rc3;
    return rc != EMB_OK ? rc : rc2;
}

int emb_shard_get_stats(emb_shard *s, emb_shard_stats *out, int reset) {
    if (!s || !out) return fail(EMB_ERR_INVALID, "emb_shard_get_stats: NULL argument");
    *out = s->st;
    if (reset) s->st = emb_shard_stats{};
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
    hipStream_t streams[4] = {s->s_route, s->s_comm, s->s_comp, s->s_un};
    for (hipStream_t st : streams)
        if (st) (void)hipStreamSynchronize(st);
    for (Batch &b : s->ring) {
        DevBuf *bufs[8] = {&b.req_send, &b.meta, &b.slotmap, &b.counts_in, &b.wc_send, &b.req_recv, &b.ret_send, &b.ret_recv};
        for (DevBuf *d : bufs)
            if (d->p) (void)hipFree(d->p);
        if (b.counts_host) (void)hipHostFree(b.counts_host);
        if (b.wc_host) (void)hipHostFree(b.wc_host);
        if (b.flag) (void)hipHostFree(b.flag);
        hipEvent_t evs[7] = {b.ev_in, b.ev_routed, b.ev_req, b.ev_served, b.ev_ret, b.ev_local, b.ev_out};
        for (hipEvent_t ev : evs)
            if (ev) (void)hipEventDestroy(ev);
    }
    if (s->work.p) (void)hipFree(s->work.p);
    for (hipStream_t st : streams)
        if (st) (void)hipStreamDestroy(st);
    delete s;
    return EMB_OK;
}

}  // extern "C"
