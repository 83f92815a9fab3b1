// Internal view of a peer group (pimemb_peer.cpp) for the sharded step (pimemb_shard.cpp) and its handshake kernels.
#pragma once

#include "pimemb_internal.h"

namespace pimemb {

constexpr uint32_t kPeerSlots = 8;          // mailbox slots per (dst, src): a batch uses slot seq % kPeerSlots
constexpr uint32_t kPeerMsgBytes = 32768;   // one mailbox: two flag words + the request message (up to ~680 whole tables per owner)
constexpr uint32_t kPeerMsgWords = (kPeerMsgBytes - 64) / 4;

// mailbox [dst][src][slot], in the job's shared segment: written by SRC's GPU (tiny kernels, in stream order behind the
// data they announce), read by DST's host with plain loads
struct PeerMsg {
    volatile unsigned long long posted;    // = seq + 1 once `words` describe what src asks dst for in batch seq
    volatile unsigned long long served;    // = seq + 1 once everything src OWES dst for batch seq is in dst's HBM
    unsigned long long pad[6];
    uint32_t words[kPeerMsgWords];
};
static_assert(sizeof(PeerMsg) == kPeerMsgBytes, "mailbox size");

// what the posting kernel needs besides the routing `meta` block: where this rank's constants for destination p sit
// (pinned host memory the kernel reads) and how many words they are
struct PeerPostArgs {
    unsigned long long box[64];     // device address of mailbox [p][me][slot] (0: p is not a remote peer)
    const uint32_t *consts[64];     // per destination: host-written words appended to the message
    uint32_t n_consts[64];
};

PeerMsg *peer_box(emb_peer *p, int dst, int src, uint32_t slot);
PeerMsg *peer_box_dev(emb_peer *p, int dst, int src, uint32_t slot);
int peer_rank(const emb_peer *p);
int peer_world(const emb_peer *p);
// `bytes` at arena offset `off` of rank r as this process addresses them (nullptr: outside the arena / across two of its chunks)
char *peer_ptr(const emb_peer *p, int r, uint64_t off, uint64_t bytes);
// arena offset of a pointer into THIS rank's arena (~0: not inside it)
uint64_t peer_offset(const emb_peer *p, const void *ptr, uint64_t bytes);
uint64_t peer_arena_bytes(const emb_peer *p, int r);
double peer_timeout_s(const emb_peer *p);
// Users of the group's mailboxes (shard objects) are created in the same order on every rank: the n-th one tags its flag
// words with n, so words left behind by an earlier user never look like news.
uint64_t peer_next_epoch(emb_peer *p);
bool peer_owns(const emb_peer *p, const void *ptr, uint64_t bytes);

// Message of src for dst, batch seq (all uint32 words):
//   [0 .. 2(Kr+1))  {sub-bags, indices} per row-split table + the peaks entry   (copied from meta by the kernel)
//   [2(Kr+1)]       first word of dst's request piece inside src's req_send      (meta.piece[dst])
//   [2(Kr+1)+1]     first partial row of shard dst inside src's ret_recv        (meta.ret_row0[dst * Kr])
//   then the host-written constants of this destination (PeerPostArgs::consts)
hipError_t launch_peer_post(const uint32_t *meta, uint32_t n_row_tables, uint32_t n_shards, const PeerPostArgs &args,
                            unsigned long long value, hipStream_t stream);
// served-flags: boxes[i]->served = value for every listed mailbox (device addresses), behind everything queued before
struct PeerDoneArgs {
    unsigned long long box[64];
    uint32_t n;
};
hipError_t launch_peer_done(const PeerDoneArgs &args, unsigned long long value, hipStream_t stream);

// Checked shards (EMB_SHARD_CHECK_SERVED): behind a COUNTED ranged launch, carry the per-descriptor served-bag counts
// (emb_lookup_ranged_counted: EMB_SERVED_LANES words per counter) to whoever asked.  Segment i: n_counted counters, each summed
// over its lanes (the lanes zeroed for their next use: atomic exchanges), then n_fill "not counted here: the serving rank
// validated these pieces itself" entries (count 0xffffffff).  Every entry is ONE self-describing 64-bit word, (tag << 32) | count,
// stored with a single 8-byte store into pinned host memory of this rank (its own requests) or the tail of a requester's
// mailbox: the reader polls the word until it carries the batch's tag -- no fence, no ticket, no flag on the counts' path (the
// first version of this kernel published plain words behind two system-scope fences, a last-workgroup ticket and a flag: 6.5 us
// per launch, profiles/r05/README.md).  The `served` words of peer-store mode (flag / value) are raised by workgroup 0 behind one
// system-scope fence, as launch_peer_done does -- they speak for the lookup's stores, which the kernel boundary completed.
struct ServedSeg {
    uint32_t *ctr;               // first counter (HBM of this rank), counters EMB_SERVED_BYTES apart
    unsigned long long *dst;     // where the requester polls: one 64-bit word per entry
    uint32_t n_counted, n_fill;
    uint32_t tag, pad;
};
constexpr uint32_t kServedSegs = 66;
struct ServedArgs {
    ServedSeg seg[kServedSegs];
    unsigned long long flag[kServedSegs];   // addresses of 64-bit flag words (peer-store mode: the requesters' `served`)
    unsigned long long value[kServedSegs];
    uint32_t n_seg, n_flag;
};
hipError_t launch_served_counts(const ServedArgs &args, hipStream_t stream);

}  // namespace pimemb
