"""GPU: the sharded lookup as one library call per batch (emb_shard_* / ShardedEmbeddingBags) against the oracle -- one
rank with every placement kind, and several REAL RCCL ranks on the one GPU (every rank claims a host of its own, so RCCL
connects them through its socket transport; the transfers, their order and their sizes are the ones an 8-GPU node sees,
minus the links).  The per-rank work is tests/shard_worker.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cfg, world, tmp_path, timeout=420):
    cfg = dict(cfg, out=str(tmp_path / "shard"))
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", PIMEMB_SHARD_TIMEOUT_S="90")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), json.dumps(cfg)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("sharded job timed out")
        outs.append(o)
    res = []
    for r in range(world):
        path = cfg["out"] + ".rank%d.json" % r
        assert os.path.exists(path), f"rank {r} left no result:\n{outs[r][-3000:]}"
        st = json.load(open(path))
        assert st["ok"], f"rank {r}:\n{st.get('error')}\n{outs[r][-2000:]}"
        res.append(st)
    return res


ONE_RANK = dict(rows=[7, 300, 5000, 64, 2000, 900], dim=16, bags=37, max_len=5,
                kinds=["replicated", "whole", "row_split", "replicated", "row_split", "whole"])


@pytest.mark.parametrize("variant", ["ragged", "one-hot", "one-hot-direct", "fixed-pooling", "int64", "int64-one-hot", "int64-direct", "f16-dim64", "f16-direct"])
def test_shard_one_rank_every_placement(variant, tmp_path):
    """A world of one rank (no communicator): replicated, whole and row-split tables in one call, ragged bags (empty ones
    included), depth 0 (forward) and depths 2 / 3 (submit / wait / flush) -- every table bit for bit the oracle's."""
    cfg = dict(ONE_RANK)
    if variant == "one-hot":
        cfg.update(max_len=1, fixed=True)
    elif variant == "one-hot-direct":       # unchecked shard + one index per bag: the row-split tables take the direct path
        cfg.update(max_len=1, fixed=True, check=False, expect_direct=True)
    elif variant == "f16-direct":
        cfg.update(max_len=1, fixed=True, check=False, expect_direct=True, f16=True, dim=64)
    elif variant == "fixed-pooling":
        cfg.update(max_len=4, fixed=True)
    elif variant == "int64":                # DLRM's dtype, handed over in place: the router reads int64, pieces are uint32 (two launches)
        cfg.update(int64=True)
    elif variant == "int64-one-hot":        # ... a checked shard's counted ranged launch over int64 arrays
        cfg.update(int64=True, max_len=1, fixed=True)
    elif variant == "int64-direct":
        cfg.update(int64=True, max_len=1, fixed=True, check=False, expect_direct=True)
    elif variant == "f16-dim64":
        cfg.update(f16=True, dim=64)
    res = _run(cfg, 1, tmp_path)
    for d in (2, 3):
        st = res[0]["stats_depth%d" % d]
        assert st["n_batches"] == 6 and st["bytes_to_peers"] == 0 and st["served_algorithmic_bytes"] > 0


@pytest.mark.parametrize("variant,world", [("ragged", 2), ("ragged", 3), ("one-hot", 4), ("pooled-whole", 2), ("self-via-comm", 2),
                                           ("empty-rank", 3), ("int64-ragged", 3), ("int64-pooled-whole", 2)])
def test_shard_rccl_ranks_on_one_gpu(variant, world, tmp_path):
    """2-4 RCCL ranks: planner-made placement (replicated + whole + row-split), every rank its own ragged batches of a
    different size, both forms of the call, all tables on all ranks against the oracle."""
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=29, max_len=6,
               expect_kinds=["replicated", "whole", "row_split"])
    if variant.startswith("int64"):        # int64 arrays in place: whole tables' index / offset arrays travel at 8 bytes per entry
        cfg.update(int64=True)
    if variant == "one-hot":
        cfg.update(max_len=1, fixed=True, dim=64, rep=64 * 64 * 4, split=3000 * 64 * 4)
    elif variant.endswith("pooled-whole"):        # the planner's return-volume term: the big pooled table stays whole
        cfg.update(plan_pooling=6.0, expect_kinds=["replicated", "whole"])
    elif variant == "self-via-comm":
        cfg.update(self_via_comm=True)
    elif variant == "empty-rank":
        cfg.update(empty_rank=1)
    res = _run(cfg, world, tmp_path)
    for st in res:
        assert st["stats_depth0"]["bytes_to_peers"] > 0 and st["stats_depth2"]["n_batches"] == 6 and st["stats_depth3"]["n_batches"] == 6
    if variant == "self-via-comm":
        assert all(st["stats_depth2"]["bytes_to_self"] > 0 for st in res)


@pytest.mark.parametrize("variant,world", [("ragged", 2), ("ragged", 3), ("one-hot", 4), ("one-hot-direct", 3), ("one-hot-direct-empty-rank", 2),
                                           ("whole-only", 2), ("empty-rank", 3), ("ragged-three-chunk-arena", 2), ("int64-ragged", 2),
                                           ("int64-one-hot-direct", 3)])
def test_shard_peer_stores_ranks_on_one_gpu(variant, world, tmp_path):
    """EMB_SHARD_PEER_STORES: the collective-free exchange with 2-4 PROCESSES on the one GPU (HIP IPC mappings of each
    other's arenas, handshake through the job's shared-memory segment; no RCCL communicator exists in these jobs).  The
    owner's fused lookup gathers the requesters' indices in place and stores pooled rows straight into their buffers; every
    table on every rank against the oracle, depth 0 / 2 / 3, and -- being the same arithmetic -- the same bits as the RCCL path."""
    import uuid
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=29, max_len=6,
               expect_kinds=["replicated", "whole", "row_split"], peer=True, peer_tag="t" + uuid.uuid4().hex[:12])
    if variant.startswith("int64"):        # the peers gather through each other's int64 arrays in place
        cfg.update(int64=True)
        variant = variant[len("int64-"):]
    if variant == "one-hot":
        cfg.update(max_len=1, fixed=True, dim=64, rep=64 * 64 * 4, split=3000 * 64 * 4)
    elif variant.startswith("one-hot-direct"):      # no router, no counts, no un-router: every shard scans the peers' raw index arrays
        cfg.update(max_len=1, fixed=True, check=False, expect_direct=True, dim=32, rep=64 * 32 * 4, split=3000 * 32 * 4)
        if variant.endswith("empty-rank"):
            cfg.update(empty_rank=1)
    elif variant == "whole-only":
        cfg.update(split=10 ** 12, expect_kinds=["replicated", "whole"], max_len=5, fixed=True)
    elif variant == "empty-rank":
        cfg.update(empty_rank=1)
    elif variant == "ragged-three-chunk-arena":     # an arena of 2.5 GiB = three IPC mappings (one of more than 2 GiB never maps on this
        cfg.update(arena_bytes=int(2.5 * (1 << 30)), arena_filler=[900 << 20, 600 << 20], depths=[0, 3])   # runtime); buffers land in chunks 1 and 2
    res = _run(cfg, world, tmp_path)
    for st in res:
        assert st["peer"]["world"] == world and st["stats_depth3"]["n_batches"] == 6 and st["stats_depth0"]["bytes_to_peers"] > 0
        if variant == "ragged-three-chunk-arena":
            assert st["peer"]["arena_bytes"] == int(2.5 * (1 << 30)) and st["peer"]["used"] > 1 << 30


def test_shard_peer_stores_missing_rank_times_out_nonzero(tmp_path):
    """A rank that never joins: the others give up after PIMEMB_SHARD_TIMEOUT_S with an error (non-zero exit), no hang."""
    import uuid
    cfg = dict(rows=[7, 300], dim=16, rep=0, split=10 ** 12, bags=5, max_len=2, peer=True, peer_tag="t" + uuid.uuid4().hex[:12],
               out=str(tmp_path / "shard"), depths=[0])
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY="0", PIMEMB_SHARD_TIMEOUT_S="3", PIMEMB_TEST_NO_TORCH_DIST="1")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), json.dumps(cfg)], env=env,
                       capture_output=True, text=True, timeout=180)
    assert p.returncode != 0
    assert "did not" in p.stdout + p.stderr or "within" in p.stdout + p.stderr


def test_sharded_apply_emb_harness_peer_stores_does_not_eat_the_arena(tmp_path):
    """A training / serving loop over peer stores: 1 500 `apply_emb` calls through an arena of 4 MiB.  The arena is a bump
    allocator (nothing is freed before the group closes); round 4's harness carved fresh index / offset / output blocks out
    of it on every call and would have run dry after ~250 of these batches -- it rotates three slots per table now.  Every
    batch against the oracle on both ranks."""
    import uuid
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=19, max_len=5, depths=[0], batches=1,
               harness=True, harness_batches=1500, peer=True, peer_tag="t" + uuid.uuid4().hex[:12], arena_bytes=4 << 20)
    res = _run(cfg, 2, tmp_path)
    assert all("replicated" in st["harness"] for st in res)
    assert all(st["peer"]["arena_bytes"] == 4 << 20 for st in res)


@pytest.mark.parametrize("transport", ["rccl", "peer"])
def test_sharded_apply_emb_harness_two_ranks(transport, tmp_path):
    """dlrm_harness.ShardedEmbeddingBagCollection: the `apply_emb(lS_o, lS_i)` contract over the sharded call (int64 tensors as
    DLRM passes them, ragged bags, the planner's placement), two ranks over RCCL and over peer stores, against the oracle."""
    import uuid
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=19, max_len=5, depths=[0], batches=1,
               harness=True)
    if transport == "peer":
        cfg.update(peer=True, peer_tag="t" + uuid.uuid4().hex[:12])
    res = _run(cfg, 2, tmp_path)
    assert all("replicated" in st["harness"] for st in res)


@pytest.mark.parametrize("ids", ["int32", "int64"])
def test_shard_bad_index_raises_on_the_serving_rank_and_nobody_hangs(ids, tmp_path):
    """A row id outside its table, passed by rank 0: the rank that SERVES it (the last shard of a row-split table) raises
    IndexError after the batch has gone through all its stages -- ONCE -- and every rank finishes; the requester's bag holds a
    zero row.  int64: the id travels as a local row id no table holds (never wrapped into range)."""
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=29, max_len=1, fixed=True,
               bad_index={"rank": 0, "table": 2, "value": "beyond" if ids == "int32" else "huge"}, batches=2, int64=ids == "int64")
    res = _run(cfg, 3, tmp_path)
    for depth in (0, 2, 3):
        raised = [st["n_raised_depth%d" % depth] for st in res]
        assert raised == [0, 0, 1], raised


@pytest.mark.parametrize("ids", ["int32", "int64", "int64-two-launches"])
def test_one_bad_batch_in_a_pipelined_stream_is_reported_once_per_batch_it_spoilt(ids, tmp_path, monkeypatch):
    """Round 5 returned a serving-side finding twice -- by the submit that ran S(b) and again by the one that ran U(b): at depth
    3 two IndexErrors for one bad batch, the second on a clean submit.  Three RCCL ranks, depth 3, eight pipelined submits, ONE
    bad index in batch 2 (rank 0, a row-split table -> served by the last shard, rank 2): the launch that finds it gathers
    nothing, so batch 2 (its pieces) and -- uint32 ids: one fused launch -- the younger batch whose replicated tables rode in it
    hold zero rows: exactly one IndexError each on rank 2, none anywhere else.  int64 ids: the same -- the request pieces
    (uint32 local row ids, about one per sub-bag here) are widened into the one int64 launch (round 6); with
    PIMEMB_SHARD_WIDEN=0 they are a launch of their own, as pooled pieces always are, and only that launch is refused: exactly one."""
    if ids == "int64-two-launches":
        monkeypatch.setenv("PIMEMB_SHARD_WIDEN", "0")
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=16, rep=64 * 16 * 4, split=3000 * 16 * 4, bags=29, max_len=3,
               bad_pipeline={"rank": 0, "table": 2}, batches=2, depths=[3], int64=ids != "int32")
    res = _run(cfg, 3, tmp_path)
    raised = [st["pipeline_raised_depth3"] for st in res]
    assert raised == [0, 0, 1 if ids == "int64-two-launches" else 2], raised


@pytest.mark.parametrize("table,mode", [(0, "deferred"), (1, "deferred"), (2, "deferred"), (2, "sync"), (0, "int64-negative"), (2, "int64-huge"), (1, "int64-sync")])
def test_checked_direct_path_reports_unserved_bags_one_rank(table, mode, tmp_path):
    """A CHECKED shard keeps the direct path for one-index batches (round 4: it fell back to routing) and COUNTS what every
    launch serves: an index outside its table -- replicated (0), whole (1) or row-split (2) -- is served by nobody, the
    requester's sum falls short of its bag count and the call raises IndexError (EMB_ERR_RANGE); a clean batch afterwards is
    the oracle's, bit for bit.  expect_direct: the kernel brackets show the ranged launch and no router / un-router."""
    cfg = dict(ONE_RANK, max_len=1, fixed=True, check="sync" if mode.endswith("sync") else True, expect_direct=True,
               bad_index={"rank": 0, "table": table, "value": {"int64-negative": "negative", "int64-huge": "huge"}.get(mode, "beyond")},
               batches=3, good_after_bad=True, int64=mode.startswith("int64"))
    res = _run(cfg, 1, tmp_path)
    for depth in (0, 2, 3):        # one IndexError per bad batch -- at the call (sync) or at the next one (deferred, the default)
        assert res[0]["n_raised_depth%d" % depth] == 1, res[0]


@pytest.mark.parametrize("table,world,ids", [(2, 2, "int32"), (1, 2, "int32"), (0, 2, "int32"), (2, 3, "int32"), (2, 2, "int64"), (1, 3, "int64")])
def test_checked_direct_path_reports_unserved_bags_peer_store_ranks(table, world, ids, tmp_path):
    """The same over peer stores with 2-3 processes on the one GPU: every shard counts the bags it serves of every requester's
    raw index array, the counts return in the tail of the requester's mailbox next to the "served" word, and the REQUESTING
    rank (rank 0 handed in the bad index) raises -- nobody else does, nobody hangs, the next batch is clean on every rank."""
    import uuid
    cfg = dict(rows=[7, 300, 5000, 64, 2000, 1500], dim=32, rep=64 * 32 * 4, split=3000 * 32 * 4, bags=29, max_len=1, fixed=True,
               expect_kinds=["replicated", "whole", "row_split"], peer=True, peer_tag="t" + uuid.uuid4().hex[:12], check=True,
               expect_direct=True, bad_index={"rank": 0, "table": table, "value": "beyond" if ids == "int32" else "negative"}, batches=3,
               good_after_bad=True, int64=ids == "int64")
    res = _run(cfg, world, tmp_path)
    for depth in (0, 2, 3):
        raised = [st["n_raised_depth%d" % depth] for st in res]
        assert raised == [1] + [0] * (world - 1), (depth, raised)
