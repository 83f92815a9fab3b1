"""Soak of what round 4 added -- minutes of randomised calls, every result compared with the CPU oracle:
  * the sharded call with one rank (`ShardedEmbeddingBags`): replicated / whole / row-split tables, batches of random size
    (empty ones too), ragged bags -> routed, one index per bag -> the direct path (one ranged launch), fixed pooling 3,
    shard objects re-created with a random pipeline depth and checking on / off, waits that lag by a random amount;
  * the request queue (`RequestQueue`): host and device queues, a random number of random small requests per flush;
  * `emb_lookup_ranged` over whole tables + the shards of row-split tables in one launch, N emulated shards;
  * round 6: checked calls of the plain engine (`check=True` / `"deferred"`), two streams, bad values injected.
    python tests/soak_round4.py [seconds [seed]]      (lives under tests/: the oracle is its checker)"""
import ctypes as C
import os
import sys
import time
from importlib import import_module

import numpy as np

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import pim_embedding_lookup_amd as pel  # noqa: E402
from oracle import oracle  # noqa: E402  (tool, not product: the checker)

sh = import_module("pim-embedding-lookup_amd.sharding")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
dev = torch.device("cuda", 0)
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(seed)
L = pel.lib.load()
t_start = time.time()
n_calls = {"shard": 0, "queue": 0, "ranged": 0}
n_bags_checked = 0


def ragged(nb, max_len):
    lens = rng.integers(0, max_len + 1, size=nb)
    off = np.zeros(nb, dtype=np.int64)
    if nb:
        off[1:] = np.cumsum(lens)[:-1]
    return off, int(lens.sum())


# ---- the sharded call, one rank ---------------------------------------------------------------------------------------
rows, dim = [900, 40_000, 77, 12_000, 31_000, 5_000, 64_000, 150], 32
kinds = [sh.REPLICATED, sh.ROW_SPLIT, sh.REPLICATED, sh.WHOLE, sh.ROW_SPLIT, sh.WHOLE, sh.ROW_SPLIT, sh.REPLICATED]
units, uot = [], []
for t, k in enumerate(kinds):
    units.append(sh.Unit(t, -1 if k == sh.REPLICATED else 0, 0, rows[t], len(units)))
    uot.append([len(units) - 1])
plan = sh.ShardPlan(1, rows, dim, 4, kinds, units, uot)
tabs = [(np.random.default_rng(50 + t).standard_normal((n, dim)) * 0.05).astype(np.float32) for t, n in enumerate(rows)]
eng = pel.EmbeddingEngine(device=0, max_tables=len(units) + 40)


def shard_phase(seconds):
    global n_bags_checked
    t0 = time.time()
    while time.time() - t0 < seconds:
        # round 6: the three checking modes (off / compared inside the completing call / deferred report) and both index widths
        # (uint32 bits, or DLRM's int64 handed over in place)
        depth, check = int(rng.integers(0, 4)), [False, "sync", "deferred"][int(rng.integers(0, 3))]
        idt = [np.int32, np.int64][int(rng.integers(0, 2))]
        S = sh.ShardedEmbeddingBags(plan, eng, 0, None, depth=depth, check=check)
        S.load_tables(lambda t, lo, hi: torch.from_numpy(tabs[t][lo:hi]).to(dev))
        pending = []

        def collect(upto):
            global n_bags_checked
            while len(pending) > upto:
                seq, outs, idx, off, nb = pending.pop(0)
                if seq is not None:
                    S.wait(seq)
                torch.cuda.synchronize()
                for t in range(len(rows)):
                    want = oracle.c_bag_sum(tabs[t], idx[t], off[t]) if nb else np.zeros((0, dim), np.float32)
                    assert np.array_equal(outs[t].cpu().numpy(), want), ("shard", depth, check, t, kinds[t], nb)
                n_bags_checked += nb * len(rows)

        for _ in range(int(rng.integers(5, 40))):
            nb = 0 if rng.integers(0, 25) == 0 else int(rng.integers(1, 3000))
            shape = int(rng.integers(0, 3))             # 0 ragged (routed), 1 one index per bag (direct), 2 fixed pooling 3 (routed)
            idx, off = [], []
            for n in rows:
                if shape == 0:
                    o, ni = ragged(nb, 5)
                else:
                    fixed = 1 if shape == 1 else 3
                    o, ni = np.arange(nb, dtype=np.int64) * fixed, nb * fixed
                off.append(o)
                idx.append(rng.integers(0, n, size=ni).astype(np.int64))
            d_i = [torch.from_numpy(i.astype(idt)).to(dev) for i in idx]
            d_o = [torch.from_numpy(o.astype(idt)).to(dev) for o in off]
            if check and shape == 1 and nb and rng.integers(0, 8) == 0:
                # round 5: a checked shard keeps the direct path and COUNTS what it serves -- one index no table row answers to
                # (replicated, whole or row-split table alike) must be refused on this, the requesting, rank; the batches
                # after it are clean
                if depth:
                    S.flush()
                    collect(0)
                t_bad = int(rng.integers(0, len(rows)))
                spoiled = idx[t_bad].copy()
                b_bad = int(rng.integers(0, nb))
                spoiled[b_bad] = rows[t_bad] + int(rng.integers(0, 1000)) if (idt is np.int32 or rng.integers(0, 2)) else \
                    [-1 - int(rng.integers(0, 1000)), (1 << 32) + int(rng.integers(0, 1000))][int(rng.integers(0, 2))]     # int64: negative / >= 2^32
                bad_i = list(d_i)
                bad_i[t_bad] = torch.from_numpy(spoiled.astype(idt)).to(dev)
                raised, outs_bad = 0, None
                try:                          # "sync": the call raises; "deferred": a later call does -- report() here
                    outs_bad = S.forward(None, bad_i, fixed_pooling=1)
                except IndexError:
                    raised += 1
                try:
                    S.report()
                except IndexError:
                    raised += 1
                if raised != 1:
                    raise AssertionError(("an index beyond its table must be refused exactly once by a checked shard", raised, check, depth, t_bad, kinds[t_bad], nb))
                n_calls["refused"] = n_calls.get("refused", 0) + 1
                torch.cuda.synchronize()
                if outs_bad is not None:      # (deferred: the call returned its rows) the refused bag pooled to ZEROS, every other bag is right
                    got = outs_bad[t_bad].cpu().numpy()
                    want = tabs[t_bad][np.clip(spoiled, 0, rows[t_bad] - 1)]
                    want[b_bad] = 0
                    assert np.array_equal(got, want), ("the refused bag must hold a zero row, the others their rows", check, depth, t_bad, kinds[t_bad])
            if depth == 0 or rng.integers(0, 6) == 0:
                if depth:                 # the synchronous form in the middle of a pipelined run: drain first
                    S.flush()
                    collect(0)
                outs = S.forward(d_o, d_i) if shape == 0 else S.forward(None, d_i, fixed_pooling=1 if shape == 1 else 3)
                pending.append((None, outs, idx, off, nb))
                collect(0)
            else:
                seq, outs = S.submit(d_i, d_o) if shape == 0 else S.submit(d_i, None, fixed_pooling=1 if shape == 1 else 3)
                pending.append((seq, outs, idx, off, nb))
                # batch `seq` is complete `depth` submits later: keep at least that many pending, sometimes more (a late consumer)
                collect(depth + int(rng.integers(0, 2)))
            n_calls["shard"] += 1
        S.flush()
        collect(0)
        S.close()


# ---- the request queue -------------------------------------------------------------------------------------------------
q_tabs = list(range(20, 26))
for t in q_tabs:
    eng.load_table(t, tabs[0][:600] + np.float32(0.001 * t))
q_rows = [tabs[0][:600] + np.float32(0.001 * t) for t in q_tabs]


def queue_phase(seconds):
    global n_bags_checked
    t0 = time.time()
    qh = pel.RequestQueue(eng, pel.EMB_IDX_U32, pel.EMB_MEM_HOST)
    qd = pel.RequestQueue(eng, pel.EMB_IDX_U32, pel.EMB_MEM_DEVICE)
    while time.time() - t0 < seconds:
        host = bool(rng.integers(0, 2))
        q = qh if host else qd
        reqs = []
        for _ in range(int(rng.integers(1, 24))):
            nt = int(rng.integers(1, len(q_tabs) + 1))
            which = rng.choice(len(q_tabs), size=nt, replace=False)
            nb = int(rng.integers(1, 40))
            idx, off, outs = [], [], []
            for _t in which:
                o, ni = ragged(nb, 4)
                i = rng.integers(0, 600, size=ni).astype(np.uint32)
                idx.append(i)
                off.append(o.astype(np.uint32))
            if host:
                outs = [np.full((nb, dim), np.nan, np.float32) for _ in which]
                ticket = q.add([q_tabs[w] for w in which], idx, off, outs)
                keep = None
            else:
                d_i = [torch.from_numpy(i.view(np.int32)).to(dev) for i in idx]
                d_o = [torch.from_numpy(o.view(np.int32)).to(dev) for o in off]
                outs = [torch.full((nb, dim), float("nan"), device=dev) for _ in which]
                ticket = q.add([q_tabs[w] for w in which], d_i, d_o, outs)
                keep = (d_i, d_o)
            reqs.append((ticket, which, idx, off, outs, nb, keep))
        assert q.flush() == len(reqs)
        for ticket, which, idx, off, outs, nb, _keep in reqs:
            q.wait(ticket)
            for j, w in enumerate(which):
                want = oracle.c_bag_sum(q_rows[w], idx[j].astype(np.int64), off[j].astype(np.int64))
                got = outs[j] if host else outs[j].cpu().numpy()
                assert np.array_equal(got, want), ("queue", host, w, nb)
            n_bags_checked += nb * len(which)
        n_calls["queue"] += len(reqs)
    qh.close()
    qd.close()


# ---- ranged launches: whole tables + this "rank"'s shards, N emulated ranks ------------------------------------------------
def ranged_phase(seconds):
    global n_bags_checked
    t0 = time.time()
    while time.time() - t0 < seconds:
        N = int(rng.integers(1, 6))
        me = int(rng.integers(0, N))
        d2 = int(rng.choice([4, 16, 32, 64, 128]))
        whole_rows = [int(rng.integers(1, 3000)) for _ in range(int(rng.integers(0, 3)))]
        split_rows = [int(rng.integers(N, 60_000)) for _ in range(int(rng.integers(1, 4)))]
        per = [-(-n // N) for n in split_rows]
        e2 = pel.EmbeddingEngine(device=0, max_tables=8)
        t_all = [rng.standard_normal((n, d2)).astype(np.float32) for n in whole_rows + split_rows]
        lo_of, ok = [], True
        for t, w in enumerate(t_all):
            if t < len(whole_rows):
                e2.load_table(t, w)
                lo_of.append(0)
            else:
                k = t - len(whole_rows)
                lo, hi = min(me * per[k], w.shape[0]), min((me + 1) * per[k], w.shape[0])
                if hi <= lo:              # (an empty shard is never loaded: the planner does not make one)
                    ok = False
                    break
                e2.load_table(t, w[lo:hi])
                lo_of.append(lo)
        if ok:
            for _ in range(int(rng.integers(1, 8))):
                nb = int(rng.integers(1, 70_000 if rng.integers(0, 4) == 0 else 3000))
                n_t = len(t_all)
                idx = [rng.integers(0, w.shape[0], size=nb).astype(np.uint32) for w in t_all]
                d_i = [torch.from_numpy(i.view(np.int32)).to(dev) for i in idx]
                outs = [torch.full((nb, d2), float("nan"), device=dev) for _ in range(n_t)]
                descs = (pel.lib.EmbLookupDesc * n_t)()
                lo = (C.c_uint64 * n_t)()
                for t in range(n_t):
                    descs[t] = pel.lib.EmbLookupDesc(t, 1, d_i[t].data_ptr(), None, nb, nb, outs[t].data_ptr())
                    lo[t] = lo_of[t]
                pel.lib.check(L.emb_lookup_ranged(e2._h, descs, lo, n_t, None))
                torch.cuda.synchronize()
                for t, w in enumerate(t_all):
                    got = outs[t].cpu().numpy()
                    if t < len(whole_rows):
                        mine = np.ones(nb, bool)
                    else:
                        k = t - len(whole_rows)
                        mine = (idx[t] >= me * per[k]) & (idx[t] < min((me + 1) * per[k], w.shape[0]))
                    assert np.array_equal(got[mine], w[idx[t][mine]]), ("ranged", N, me, d2, t)
                    assert np.isnan(got[~mine]).all(), ("ranged: a bag of another shard was written", N, me, d2, t)
                    n_bags_checked += int(mine.sum())
                n_calls["ranged"] += 1
        e2.close()


# ---- round 6: checked calls of the plain engine, the verdict waited for or deferred --------------------------------------------
# Random multi-table calls on two streams, index width either, ragged bags; one call in seven carries 1..5 bad values (an index
# out of range, a broken offset).  A refused call leaves its outputs untouched (NaN-filled before), every other call equals the
# oracle, and every bad call is reported EXACTLY once -- by itself (sync), by a later call, by check_report() (deferred).
def checked_phase(seconds):
    global n_bags_checked
    t0 = time.time()
    e3 = pel.EmbeddingEngine(device=0, max_tables=8)
    c_rows = [33, 70_000, 1500, 4096, 9]
    c_dim = int(rng.choice([8, 16, 64]))
    c_tabs = [rng.standard_normal((n, c_dim)).astype(np.float32) for n in c_rows]
    for t, w in enumerate(c_tabs):
        e3.load_table(t, w)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    while time.time() - t0 < seconds:
        mode = ["deferred", True][int(rng.integers(0, 2))]
        idt = [np.int32, np.int64][int(rng.integers(0, 2))]
        calls, n_bad_calls, n_reports = [], 0, 0
        for _ in range(int(rng.integers(3, 90))):
            nt = int(rng.integers(1, len(c_rows) + 1))
            which = [int(x) for x in rng.choice(len(c_rows), size=nt, replace=False)]
            nb = int(rng.integers(1, 20_000 if rng.integers(0, 6) == 0 else 600))
            idx, off = [], []
            for t in which:
                o, n = ragged(nb, 3)
                idx.append(rng.integers(0, c_rows[t], size=n).astype(idt))
                off.append(o.astype(idt))
            spoil = rng.integers(0, 7) == 0
            if spoil:
                hit = False
                for _ in range(int(rng.integers(1, 6))):
                    k = int(rng.integers(0, nt))
                    if rng.integers(0, 2) and len(idx[k]):
                        idx[k][int(rng.integers(0, len(idx[k])))] = c_rows[which[k]] + int(rng.integers(0, 3))
                        hit = True
                    elif nb >= 2:
                        off[k][int(rng.integers(1, nb))] = len(idx[k]) + 1 + int(rng.integers(0, 3))
                        hit = True
                spoil = hit
            st = streams[int(rng.integers(0, 2))]
            with torch.cuda.stream(st):
                d_i = [torch.from_numpy(i).to(dev) for i in idx]
                d_o = [torch.from_numpy(o).to(dev) for o in off]
                outs = [torch.full((nb, c_dim), float("nan"), device=dev) for _ in which]
                n_bad_calls += int(spoil)
                try:
                    e3.lookup_batched(which, d_i, d_o, outs, check=mode)
                except IndexError:
                    n_reports += 1
            calls.append((which, idx, off, outs, spoil, nb))
        try:
            e3.check_report()
        except IndexError:
            n_reports += 1
        torch.cuda.synchronize()
        n_found = 0
        for which, idx, off, outs, spoil, nb in calls:
            for k, t in enumerate(which):
                got = outs[k].cpu().numpy()
                if spoil:
                    assert np.isnan(got).all(), ("checked: a refused call wrote rows", mode, t)
                else:
                    ii, oo = (idx[k].view(np.uint32), off[k].view(np.uint32)) if idx[k].dtype == np.int32 else (idx[k], off[k])
                    assert np.array_equal(got, oracle.c_bag_sum(c_tabs[t], ii, oo)), ("checked", mode, t, nb)
                    n_bags_checked += nb
            n_found += int(spoil)
        # deferred: several bad calls may be read in one go (one report for them); never more reports than bad calls, never none
        assert (n_reports == n_bad_calls) if mode is True else (n_bad_calls == 0) == (n_reports == 0) and n_reports <= n_bad_calls, (mode, n_reports, n_bad_calls)
        n_calls["checked"] = n_calls.get("checked", 0) + len(calls)
        n_calls["refused"] = n_calls.get("refused", 0) + n_found
    e3.close()


phase = 0
while time.time() - t_start < budget:
    slice_s = min(20.0, max(1.0, budget - (time.time() - t_start)))
    (shard_phase, queue_phase, ranged_phase, checked_phase)[phase % 4](slice_s)
    phase += 1
    print("  %4.0f s: %s, %d bags checked" % (time.time() - t_start, n_calls, n_bags_checked), flush=True)
torch.cuda.synchronize()
eng.close()
print("soak (round 4, seed %d): %s calls, %d bags checked against the oracle in %.0f s -- all equal" % (seed, n_calls, n_bags_checked, time.time() - t_start))
