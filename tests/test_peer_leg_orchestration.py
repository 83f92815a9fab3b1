"""CPU: `bench.py --exchange both` -- what happens to the RCCL leg's numbers when the peer-store leg succeeds, cannot come up,
hangs, or leaves wrong rows (dist_bench.peer_leg, the orchestration only: two gloo processes, the sharded leg itself replaced by
a stand-in that returns a canned line, raises or sleeps).  The rules: a peer leg that cannot come up is
recorded as {"skipped": reason} and does NOT fail the run; one that comes up and differs from the RCCL leg's bits, or from the
expected rows, does; every rank ends the leg the same way; a hang ends at the leg's deadline with the line printed, status 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import json, os, sys, time, types
sys.path.insert(0, %(root)r)
import torch
import torch.distributed as dist
from importlib import import_module
db = import_module("pim-embedding-lookup_amd.dist_bench")
rank, world, mode = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), sys.argv[1]
dist.init_process_group("gloo", rank=rank, world_size=world)
ctx = dict(rank=rank, world=world, dev=torch.device("cpu"), backend="gloo", stage_cpu=True)
result = None
if rank == 0:       # what the RCCL leg left behind
    result = {"value": 1.0, "verified": True, "config": {"last_step_sharded_outputs_sha1": "abc", "exchange": {}}, "roofline": {}}
printed = []
def emit(res):
    if rank == 0 and res is not None and not printed:
        printed.append(1)
        print(json.dumps(res), flush=True)
def finish(res):
    return res
def canned(digest):
    return {"value": 2.0, "ms_per_step": 0.05, "ms_per_step_event": 0.05, "steps": 4,
            "config": {"exchange_transport": "peer stores (HIP IPC, no RCCL in the data path; fine-grained arena)", "direct_one_hot_path": True,
                       "exchange": {"bytes_out_per_rank_per_step": 10, "host_us_per_step": 1.0}, "last_step_sharded_outputs_sha1": digest},
            "roofline": {"exchange": {"step_frac": 0.5}}}
def shard_leg(a, peak, c, rep, transport=None):
    assert transport == "peer" and os.environ.get("PIMEMB_PEER_WATCHDOG") == "report"
    if mode == "ok":
        return canned("abc") if rank == 0 else None
    if mode == "other-bits":
        return canned("xyz") if rank == 0 else None
    if mode == "one-rank-cannot":
        if rank == 1:
            raise RuntimeError("emb_peer_create: this runtime cannot allocate / export fine-grained device memory")
        return canned("abc") if rank == 0 else None
    if mode == "wrong-rows":
        if rank == 1:
            raise AssertionError("rank 1: timed step 3 table 2 (row_split) differs from the expected rows")
        return canned("abc") if rank == 0 else None
    if mode == "hang":
        time.sleep(3600)
    if mode.startswith("abort-rank"):            # what a GPU memory fault ends in: the runtime calls abort()
        if rank == int(mode[-1]):
            os.abort()
        time.sleep(1.0)
        return canned("abc") if rank == 0 else None
args = types.SimpleNamespace(steps=4, warmup=2)
db.peer_leg(args, 64 << 20, result, emit, finish, ctx, shard_leg, 8000.0, True)
emit(finish(result))
dist.barrier()
dist.destroy_process_group()
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(mode, tmp_path, env_extra=None):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT})
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), **(env_extra or {}))
        procs.append(subprocess.Popen([sys.executable, str(script), mode], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=180) for p in procs]
    lines = [l for l in outs[0][0].splitlines() if l.startswith("{")]
    return [p.returncode for p in procs], (json.loads(lines[-1]) if lines else None), outs


def test_peer_leg_success_is_reported_next_to_the_rccl_leg(tmp_path):
    rcs, d, outs = _run("ok", tmp_path)
    assert rcs == [0, 0], outs
    assert d["value"] == 1.0 and d["verified"] is True                      # the RCCL leg's numbers are untouched
    assert d["value_exchange_peer"] == 2.0 and d["ms_per_step_exchange_peer"] == 0.05 and d["exchange_same_bits"] is True
    assert d["exchange_peer"]["verified"] is True and d["config"]["exchange_peer"] == d["exchange_peer"]
    assert "peer stores" in d["exchange_transport_peer"] and d["roofline"]["exchange_peer"]["step_frac"] == 0.5


def test_peer_leg_with_other_bits_fails_the_run(tmp_path):
    rcs, d, outs = _run("other-bits", tmp_path, {"PIMEMB_TEARDOWN_TIMEOUT": "5"})
    assert rcs[0] == 1, outs
    assert d["verified"] is False and d["exchange_same_bits"] is False and d["value"] == 1.0


def test_peer_leg_that_one_rank_cannot_bring_up_is_skipped_on_all(tmp_path):
    rcs, d, outs = _run("one-rank-cannot", tmp_path)
    assert rcs == [0, 0], outs
    assert d["verified"] is True and d["value"] == 1.0 and "value_exchange_peer" not in d
    assert "skipped" in d["exchange_peer"] and "another rank" in d["exchange_peer"]["skipped"]
    assert "fine-grained" in outs[1][1]                                       # rank 1's stderr says why


def test_peer_leg_with_wrong_rows_fails_on_every_rank(tmp_path):
    rcs, d, outs = _run("wrong-rows", tmp_path)
    assert rcs == [1, 1], outs
    assert d["verified"] is False and "failed" in d["exchange_peer"] and d["value"] == 1.0


def test_peer_leg_that_hangs_ends_at_its_deadline_with_the_line_printed(tmp_path):
    rcs, d, outs = _run("hang", tmp_path, {"PIMEMB_PEER_LEG_TIMEOUT": "2"})
    assert rcs == [0, 0], outs
    assert d["verified"] is True and d["value"] == 1.0
    assert "no result within 2 s" in d["exchange_peer"]["skipped"]


@pytest.mark.parametrize("who", [0, 1])
def test_peer_leg_in_which_a_rank_dies_keeps_the_rccl_numbers(tmp_path, who):
    """A fatal signal inside the leg (abort(): the runtime's answer to a GPU memory fault) on rank 0 or on another rank: the
    line with the RCCL leg's numbers is still printed -- by rank 0's signal handler (emb_peer_last_words) or by rank 0 once its
    collective with the dead rank fails -- and EVERY rank ends with status 0, so the launcher does not tear the job down."""
    rcs, d, outs = _run("abort-rank%d" % who, tmp_path, {"PIMEMB_PEER_LEG_TIMEOUT": "60"})
    assert rcs == [0, 0], outs
    assert d["verified"] is True and d["value"] == 1.0 and "value_exchange_peer" not in d
    assert "skipped" in d["exchange_peer"] and ("fatal signal" if who == 0 else "left the job") in d["exchange_peer"]["skipped"]
