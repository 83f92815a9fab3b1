"""CPU: the pieces of bench.py's JSON contract that need no GPU -- which committed PMC profile a command maps to,
how the roofline object is priced (`frac` on algorithmic bytes on EVERY line; the measured HBM-side rate of the committed
PMC profile beside it as `frac_measured`), and that every profile entry points at files that exist."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _args(**kw):
    base = dict(workload="c2", batch=None, tables=None, hot_rows=0, streams=1, pooling=None, index_dist=None, nbatch=8)
    base.update(kw)
    return argparse.Namespace(**base)


def test_profile_key_of_a_command():
    assert bench.profile_key(_args(), dict(L=1, dist="uniform")) == "c2"
    assert bench.profile_key(_args(index_dist="zipf"), dict(L=1, dist="zipf")) == "c2-zipf"
    assert bench.profile_key(_args(workload="c3"), dict(L=32, dist="zipf")) == "c3"
    assert bench.profile_key(_args(workload="c3"), dict(L=32, dist="uniform")) == "c3-uniform"
    assert bench.profile_key(_args(workload="c4"), dict(L=1, dist="uniform")) == "c4"
    assert bench.profile_key(_args(workload="c4", pooling=32), dict(L=32, dist="uniform")) == "c4-l32"
    assert bench.profile_key(_args(workload="c5"), dict(L=32, dist="mixed")) == "c5"
    # anything that changes the traffic has no profile: another batch, table count, hint, stream count or rotation length
    for kw in (dict(batch=2048), dict(tables=16, workload="c3"), dict(hot_rows=32, workload="c3"), dict(streams=2),
               dict(nbatch=2)):
        assert bench.profile_key(_args(**kw), dict(L=1, dist="uniform")) is None


def test_traffic_json_entries_are_backed_by_files():
    data = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    keys = [k for k in data if not k.startswith("_")]
    assert {"c2", "c2-zipf", "c3", "c3-uniform", "c4", "c4-l32", "c5"} <= set(keys)
    for k in keys:
        e = data[k]
        assert os.path.exists(os.path.join(ROOT, e["source"])), e["source"]
        assert e["traffic_bytes_per_launch"] == e["read_bytes"] + e["write_bytes"] > 0
        if "fetch_size_x2_bytes" in e:      # FETCH_SIZE tallies a 128-B request at 64 B on gfx950: 2x it ~ the RDREQ-derived bytes
            assert abs(e["fetch_size_x2_bytes"] - e["read_bytes"]) <= 0.02 * e["read_bytes"]
        if e.get("round") == "r02":
            stats = os.path.join(ROOT, "profiles", "r02", f"{k}_kernel_stats.csv")
            assert os.path.exists(stats) and e["kernel_avg_ns"] > 0


def test_roofline_object_pricing():
    alg, us = 138_936_512, 20.0
    # the metric's config: algorithmic basis (the contract's figure); the profile's bytes reported beside it
    r = bench.roofline_object(alg, us, dict(traffic_bytes_per_launch=114_309_676, read_bytes=48_869_523, source="x",
                                            tcc_hit=894_895, tcc_miss=892_590), None)
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["achieved"] - alg / 20e-6 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic"] == 114_309_676 and r["basis"] == "algorithmic bytes" and 0.3 < r["l2_frac"] < 0.35
    assert r["cache_served"] is False and abs(r["frac_measured"] - 114_309_676 / 20e-6 / 1e9 / 8000.0) < 1e-12
    # cache-served: `frac` stays algorithmic (and says why it exceeds 1); the measured HBM-side rate has its own keys
    alg3, us3 = 13_390_000_000, 668.0
    r3 = bench.roofline_object(alg3, us3, dict(traffic_bytes_per_launch=3_170_800_000, read_bytes=2_768_000_000, source="x",
                                               tcc_hit=40_428_845, tcc_miss=24_771_780), 1_624_000_000, meta_bytes=104_000_000)
    assert abs(r3["achieved"] - alg3 / 668e-6 / 1e9) < 1e-6 and r3["achieved"] == r3["achieved_algorithmic"] and r3["frac"] > 2
    assert r3["cache_served"] is True and abs(r3["achieved_measured"] - 3_170_800_000 / 668e-6 / 1e9) < 1e-6
    assert 0.55 < r3["frac_measured"] < 0.62 and 0.60 < r3["l2_hit_rate"] < 0.64 and "frac_measured" in r3["basis"]
    assert abs(r3["read_over_unique_rows"] - (2_768_000_000 - 104_000_000) / 1_624_000_000) < 1e-9
    # no profile for the command: algorithmic, and a fraction above 1 says what it is
    r0 = bench.roofline_object(alg3, us3, None, None)
    assert r0["traffic"] is None and r0["frac"] > 1 and "not an HBM utilisation" in r0["basis"] and "frac_measured" not in r0
    assert bench.roofline_object(alg, us, None, None)["basis"] == "algorithmic bytes"
