"""Soak of the sharded call with several PROCESSES on the one GPU: tests/shard_worker.py (every rank its own random ragged
batches, every table of every batch against the oracle) with many more and larger batches than the GPU tests use --
peer stores and RCCL (sockets over loopback), 2-4 ranks, depths 0 / 2 / 3, ragged / one index per bag / whole tables.
    python tests/soak_ranks.py [legs ...]      (lives under tests/: the oracle is the workers' checker)"""
import json
import os
import socket
import subprocess
import sys
import tempfile
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
BASE = dict(rows=[90, 40_000, 3_000, 90_000, 500, 25_000, 70_000, 12], dim=32, bags=2500, max_len=6, batches=80, depths=[0, 2, 3],
            kinds=["replicated", "row_split", "whole", "row_split", "replicated", "whole", "row_split", "replicated"])
LEGS = {
    "peer3-ragged": (3, dict(peer=True, arena_bytes=3 << 30)),
    "peer3-onehot-direct": (3, dict(peer=True, arena_bytes=3 << 30, max_len=1, fixed=True, check=False, expect_direct=True)),
    "peer4-pooled-fixed": (4, dict(peer=True, arena_bytes=3 << 30, max_len=4, fixed=True, batches=60)),
    "peer2-empty-rank": (2, dict(peer=True, arena_bytes=3 << 30, empty_rank=1)),
    "rccl3-ragged": (3, dict(batches=100)),
    "rccl2-onehot": (2, dict(max_len=1, fixed=True, batches=100)),
    "rccl4-ragged-self-via-comm": (4, dict(batches=60, self_via_comm=True)),
    # round 6: DLRM's int64 ids handed over in place
    "rccl3-ragged-int64": (3, dict(batches=80, int64=True)),
    "peer3-onehot-direct-int64-checked": (3, dict(peer=True, arena_bytes=3 << 30, max_len=1, fixed=True, check=True, expect_direct=True, int64=True)),
}


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def run(name, world, extra):
    t0 = time.time()
    with tempfile.TemporaryDirectory() as tmp:
        cfg = dict(BASE, **extra)
        cfg["out"] = os.path.join(tmp, "soak")
        if cfg.get("peer"):
            cfg["peer_tag"] = "soak-%d-%s" % (os.getpid(), name)
        port = free_port()
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                       HSA_ENABLE_IPC_MODE_LEGACY="0", PIMEMB_SHARD_TIMEOUT_S="120")
            procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "shard_worker.py"), json.dumps(cfg)], env=env,
                                          stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
        outs = []
        for p in procs:
            try:
                o, _ = p.communicate(timeout=900)
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                print("leg %s: TIMED OUT" % name, flush=True)
                return False
            outs.append(o)
        ok, counted = True, 0
        for r in range(world):
            path = cfg["out"] + ".rank%d.json" % r
            st = json.load(open(path)) if os.path.exists(path) else {"ok": False, "error": outs[r][-2000:]}
            counted += sum(int(v.get("n_batches", 0)) for k, v in st.items() if k.startswith("stats_depth"))
            if not st["ok"]:
                ok = False
                print("leg %s rank %d FAILED:\n%s" % (name, r, st.get("error")), flush=True)
        n = cfg["batches"] * len(cfg["depths"])
        print("leg %-26s %d ranks x %d batches (depths %s) of ~%d bags x %d tables -- the library counted %d batches through all stages: %s in %.0f s"
              % (name, world, n, cfg["depths"], cfg["bags"], len(cfg["rows"]), counted,
                 "every table of every batch equals the oracle's" if ok and counted >= world * n else "FAILED", time.time() - t0), flush=True)
        ok = ok and counted >= world * n
        return ok


if __name__ == "__main__":
    legs = sys.argv[1:] or list(LEGS)
    good = all([run(k, *LEGS[k]) for k in legs])
    sys.exit(0 if good else 1)
