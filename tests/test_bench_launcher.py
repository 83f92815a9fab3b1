"""bench.py's self-launch (`python bench.py --gpus N` with no launcher): one child per rank, rank 0's
stdout relayed, the worst child's exit status returned, a failed rank takes the job down.  CPU only:
the ranks here are a stand-in script that meets over gloo."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RANK_OK = textwrap.dedent("""
    import json, os, sys
    import torch, torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    assert os.environ["MASTER_ADDR"] == "127.0.0.1" and os.environ["LOCAL_RANK"] == str(rank)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t)
    print("noise from rank", rank, file=sys.stderr)
    print(json.dumps({"rank": rank, "sum": float(t.item()), "argv": sys.argv[1:]}) if rank == 0 else "not the json line")
    dist.destroy_process_group()
""")

RANK_FAIL = textwrap.dedent("""
    import os, sys, time
    rank = int(os.environ["RANK"])
    if rank == 1:
        sys.exit(7)
    time.sleep(600)          # a peer stuck in a collective that will never complete
""")


def _run(tmp_path, body, n, extra_env=None):
    script = tmp_path / "rank.py"
    script.write_text(body)
    driver = ("import sys; sys.path.insert(0, %r); import bench; "
              "sys.exit(bench.self_launch(%d, ['--x', '1'], script=%r))" % (ROOT, n, str(script)))
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, "-c", driver], capture_output=True, text=True, timeout=300, env=env)


def test_self_launch_relays_rank0_and_returns_zero(tmp_path):
    res = _run(tmp_path, RANK_OK, 3)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1                                   # ONE line on stdout: rank 0's
    d = json.loads(lines[0])
    assert d == {"rank": 0, "sum": 6.0, "argv": ["--x", "1"]}
    assert "[rank 1] not the json line" in res.stderr and "[rank 2] not the json line" in res.stderr


def test_self_launch_failed_rank_ends_the_job_nonzero(tmp_path):
    res = _run(tmp_path, RANK_FAIL, 2, {"PIMEMB_LAUNCH_GRACE": "1"})
    assert res.returncode == 143 or res.returncode == 7 or res.returncode == 137, (res.returncode, res.stderr[-2000:])
    assert res.returncode != 0 and "rank exit codes" in res.stderr


def test_bench_main_only_self_launches_without_a_launcher(monkeypatch):
    """--gpus N > 1 and no RANK / WORLD_SIZE in the environment -> self_launch; under torchrun
    (WORLD_SIZE set) the process is a rank and must not fan out again."""
    sys.path.insert(0, ROOT)
    import bench
    called = {}
    monkeypatch.setattr(bench, "self_launch", lambda n, argv, script=None: called.setdefault("n", n) and 0)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "1"])
    for k in ("RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    try:
        bench.main()
    except SystemExit as ex:
        assert ex.code in (0, 4)
    assert called == {"n": 4}
