"""GPU: the multi-GPU routing pieces of the sharded lookup on the real HIP engine -- bag routing to row-range
shards (counts first), partial-sum un-routing, the generic ShardedLookup over the engine backend, and the
bench's N > 1 legs with two ranks sharing cuda:0 (collectives over gloo).  Through the C ABI throughout."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng(pel):
    e = pel.EmbeddingEngine(device=0, max_tables=256)
    yield e
    e.close()


def _route_bags_reference(idx, off, n_idx, rps, N):
    """Host restatement of the routing rule (tests only): per shard d the sub-bag lists in bag order."""
    B = off.shape[0]
    end = np.concatenate([off[1:], [n_idx]]).astype(np.int64)
    dest = np.minimum(idx.astype(np.int64) // rps, N - 1)
    out = []
    for d in range(N):
        sub_off, lists, slots = [], [], np.full(B, 0xffffffff, dtype=np.uint32)
        pos = 0
        for b in range(B):
            sel = idx[off[b]:end[b]][dest[off[b]:end[b]] == d]
            if sel.size:
                slots[b] = len(sub_off)
                sub_off.append(pos)
                lists.append(sel.astype(np.int64) - d * rps)
                pos += sel.size
        out.append((np.array(sub_off, dtype=np.uint32),
                    np.concatenate(lists).astype(np.uint32) if lists else np.zeros(0, np.uint32), slots))
    return out


@pytest.mark.parametrize("dim,ragged", [(16, True), (128, False), (64, True)])
def test_route_bags_row_range_shards_pooled(pel, eng, oracle, dim, ragged):
    """Pooled lookups over row-split tables in ONE process: emb_route_bags cuts every bag into per-shard sub-bags
    (counts, layout, request pieces), every 'shard' serves its piece with the ordinary fused lookup, and
    emb_unroute_bags adds the partial rows in shard order.  Checked: the counts / slots / request lists against a
    host restatement of the routing rule; the result bit for bit against 'partials in shard order' built from the
    oracle, and within 1e-6 of the oracle's unsharded sum; bit-identical on a second run."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(dim)
    K, N, B = 3, 4, 3001
    rows = [100_003, 5_000, 40_001]
    rps = [-(-r // N) for r in rows]
    tabs = [pel.workloads.dlrm_table(rng, r, dim) for r in rows]
    idxs, offs = [], []
    for k in range(K):
        if ragged:
            off, n_idx = pel.workloads.ragged_offsets(rng, B, 40, p_empty=0.15)
        else:
            off, n_idx = pel.workloads.fixed_offsets(B, 32), 32 * B
        gen = pel.workloads.zipf_indices if k == 0 else pel.workloads.uniform_indices
        idxs.append(gen(rng, rows[k], n_idx))
        offs.append(off)
    # shard tables: engine table id 60 + k*N + d holds rows [d*rps, (d+1)*rps) of table k
    for k in range(K):
        for d in range(N):
            lo, hi = min(d * rps[k], rows[k]), min((d + 1) * rps[k], rows[k])
            eng.load_table(60 + k * N + d, tabs[k][lo:hi] if hi > lo else np.zeros((1, dim), np.float32))
    total = sum(i.shape[0] for i in idxs)
    sz = eng.route_bags_sizes(K, B, total, N)
    u8 = lambda n: torch.zeros(max(n, 16), dtype=torch.uint8, device=dev)
    send, meta, slots, work = u8(sz["send"]), u8(sz["meta"]), u8(sz["slots"]), u8(sz["work"])
    d_idx = [torch.from_numpy(i.view(np.int32)).to(dev) for i in idxs]
    d_off = [torch.from_numpy(o.view(np.int32)).to(dev) for o in offs]
    spec = [(d_idx[k].data_ptr(), d_off[k].data_ptr() if ragged else None, idxs[k].shape[0], 0 if ragged else 32, rps[k])
            for k in range(K)]

    def run_once():
        eng.route_bags(spec, B, N, send.data_ptr(), meta.data_ptr(), slots.data_ptr(), work.data_ptr())
        torch.cuda.synchronize()
        m = meta.view(torch.int32).cpu().numpy().view(np.uint32)
        nk = N * K
        counts = m[:2 * nk].reshape(N, K, 2)
        base = m[2 * nk:4 * nk].reshape(N, K, 2)
        piece = m[4 * nk:4 * nk + N + 1]
        row0 = m[4 * nk + N + 1:5 * nk + N + 1].reshape(N, K)
        words = send.view(torch.int32)
        # every shard serves its piece: one fused lookup per shard over the K request lists
        rets = []
        for d in range(N):
            ids, ii, oo = [], [], []
            for k in range(K):
                ns, ni = int(counts[d, k, 0]), int(counts[d, k, 1])
                if ns == 0:
                    continue
                ids.append(60 + k * N + d)
                oo.append(words[int(base[d, k, 0]):int(base[d, k, 0]) + ns])
                ii.append(words[int(base[d, k, 1]):int(base[d, k, 1]) + ni])
            n_rows = int(counts[d, :, 0].sum())
            ret = torch.empty((max(n_rows, 1), dim), dtype=torch.float32, device=dev)
            outs, cur = [], 0
            for k in range(K):
                ns = int(counts[d, k, 0])
                if ns:
                    outs.append(ret[cur:cur + ns])
                    cur += ns
            if ids:
                eng.lookup_batched(ids, ii, oo, outs)
            rets.append(ret[:n_rows])
        recv = torch.cat(rets)
        pooled = torch.full((K, B, dim), float("nan"), device=dev)
        eng.unroute_bags(recv.data_ptr(), meta.data_ptr(), slots.data_ptr(), K, B, N, dim, pooled.data_ptr())
        torch.cuda.synchronize()
        return counts.copy(), base.copy(), piece.copy(), row0.copy(), pooled.cpu().numpy()

    counts, base, piece, row0, pooled = run_once()
    words = send.view(torch.int32).cpu().numpy().view(np.uint32)
    sl = slots.view(torch.int32).cpu().numpy().view(np.uint32)[:K * N * B].reshape(K, N, B)
    pad4 = lambda v: (v + 3) & ~3
    cursor, row = 0, 0
    for d in range(N):
        assert piece[d] == cursor
        for k in range(K):
            ref = _route_bags_reference(idxs[k], offs[k].astype(np.int64), idxs[k].shape[0], rps[k], N)[d]
            ns, ni = int(counts[d, k, 0]), int(counts[d, k, 1])
            assert (ns, ni) == (ref[0].shape[0], ref[1].shape[0])
            assert base[d, k, 0] == cursor and base[d, k, 1] == cursor + pad4(ns) and row0[d, k] == row
            assert np.array_equal(words[cursor:cursor + ns], ref[0])
            assert np.array_equal(words[cursor + pad4(ns):cursor + pad4(ns) + ni], ref[1])
            assert np.array_equal(sl[k, d], ref[2])
            cursor += pad4(ns) + pad4(ni)
            row += ns
    assert piece[N] == cursor and cursor * 4 <= sz["send"]
    # expected: partial sums per shard (oracle, in index order), added in shard order from +0
    for k in range(K):
        ref = _route_bags_reference(idxs[k], offs[k].astype(np.int64), idxs[k].shape[0], rps[k], N)
        want = np.zeros((B, dim), np.float32)
        for d in range(N):
            sub_off, lst, slot = ref[d]
            if sub_off.shape[0] == 0:
                continue
            lo, hi = min(d * rps[k], rows[k]), min((d + 1) * rps[k], rows[k])
            part = oracle.c_bag_sum(tabs[k][lo:hi], lst, sub_off)
            has = slot != 0xffffffff
            want[has] = want[has] + part[slot[has]]
        assert np.array_equal(pooled[k], want), f"table {k}: not the shard-ordered sum of partials"
        full = oracle.c_bag_sum(tabs[k], idxs[k], offs[k])
        assert float(np.abs(pooled[k] - full).max()) <= 1e-6
    again = run_once()[-1]
    assert np.array_equal(again, pooled)                   # deterministic: same bits on a second run


def test_route_bags_one_index_per_bag_is_exact(pel, eng, oracle):
    """One index per bag through the bag router: every bag lives in one shard, so the un-routed rows equal the
    unsharded lookup bit for bit -- also with every request landing on ONE shard (skew: no capacity to overflow)."""
    import torch
    dev = torch.device("cuda", 0)
    rng = np.random.default_rng(11)
    K, N, B, dim = 2, 8, 20_000, 16
    rows = [1_000_000, 64_000]
    rps = [-(-r // N) for r in rows]
    tabs = [pel.workloads.dlrm_table(rng, r, dim) for r in rows]
    idxs = [pel.workloads.zipf_indices(rng, rows[0], B, permute=False),          # un-permuted Zipf: all in shard 0
            pel.workloads.uniform_indices(rng, rows[1], B)]
    assert (idxs[0] // rps[0] == 0).mean() > 0.9
    for k in range(K):
        for d in range(N):
            lo, hi = min(d * rps[k], rows[k]), min((d + 1) * rps[k], rows[k])
            eng.load_table(60 + k * N + d, tabs[k][lo:hi])
    sz = eng.route_bags_sizes(K, B, K * B, N)
    u8 = lambda n: torch.zeros(max(n, 16), dtype=torch.uint8, device=dev)
    send, meta, slots, work = u8(sz["send"]), u8(sz["meta"]), u8(sz["slots"]), u8(sz["work"])
    d_idx = [torch.from_numpy(i.view(np.int32)).to(dev) for i in idxs]
    spec = [(d_idx[k].data_ptr(), None, B, 1, rps[k]) for k in range(K)]
    eng.route_bags(spec, B, N, send.data_ptr(), meta.data_ptr(), slots.data_ptr(), work.data_ptr())
    torch.cuda.synchronize()
    m = meta.view(torch.int32).cpu().numpy().view(np.uint32)
    nk = N * K
    counts, base = m[:2 * nk].reshape(N, K, 2), m[2 * nk:4 * nk].reshape(N, K, 2)
    assert counts[:, :, 0].sum() == K * B and np.array_equal(counts[:, :, 0], counts[:, :, 1])
    words = send.view(torch.int32)
    rets = []
    for d in range(N):
        for k in range(K):
            ns = int(counts[d, k, 0])
            if ns:
                rets.append(eng.lookup(60 + k * N + d, words[int(base[d, k, 1]):int(base[d, k, 1]) + ns],
                                       words[int(base[d, k, 0]):int(base[d, k, 0]) + ns]))
    recv = torch.cat(rets)
    pooled = torch.empty((K, B, dim), device=dev)
    eng.unroute_bags(recv.data_ptr(), meta.data_ptr(), slots.data_ptr(), K, B, N, dim, pooled.data_ptr())
    torch.cuda.synchronize()
    for k in range(K):
        assert np.array_equal(pooled[k].cpu().numpy(), oracle.c_bag_sum(tabs[k], idxs[k], np.arange(B, dtype=np.uint32)))
