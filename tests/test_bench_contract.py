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
    assert {"c2", "c2-zipf", "c3", "c3-uniform", "c4", "c4-l32", "c5", "dist-c4-rows-l1", "dist-c4-rows-l32",
            "dist-c4-rows-l1-direct", "dist-c2-rows-l1-direct", "dist-c2-whole-l1"} <= set(keys)
    for k in keys:
        e = data[k]
        if e.get("round", "r00") >= "r04":          # entries collected from round 4 on are tied to a build and a kernel
            assert len(e["lib_sha256"]) == 64 and len(e["src_sha256"]) == 64 and "bag_sum" in e["kernel"], k
        one_launch = k.endswith("-direct") or "-whole-" in k     # direct one-hot path / whole tables: the step IS one lookup launch
        if one_launch:
            assert "router" not in e and "unrouter" not in e, k
        elif k.startswith("dist-"):                 # the routed sharded step: lookup entry + router / un-router sub-entries, per step
            for sub in ("router", "unrouter"):
                assert e[sub]["traffic_bytes_per_launch"] == e[sub]["read_bytes"] + e[sub]["write_bytes"] > 0
                assert e[sub]["kernel_avg_ns"] > 0
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


def test_traffic_entry_is_dropped_when_the_library_or_the_kernel_differs():
    """an entry of profiles/traffic.json is used only for the build it was collected with (the library's
    sha256 or the kernel sources' sha256 must match the loaded one) and for the kernel it was summed over; otherwise the
    line carries no traffic / frac_measured and says why."""
    ident = dict(lib_sha256="a" * 64, src_sha256="b" * 64)
    good = dict(traffic_bytes_per_launch=100, read_bytes=40, write_bytes=60, source="profiles/r04/x.txt", lib_sha256="a" * 64,
                src_sha256="c" * 64, kernel="void pimemb::bag_sum_wavebatch_kernel<unsigned int, 0, 4, ...>(...)")
    assert bench.traffic_entry_status(good, "bag_sum_wavebatch_kernel", ident) is None          # same library
    assert bench.traffic_entry_status(dict(good, lib_sha256="d" * 64, src_sha256="b" * 64), "bag_sum_wavebatch_kernel", ident) is None   # a rebuild of the same sources
    assert "another build" in bench.traffic_entry_status(dict(good, lib_sha256="d" * 64), "bag_sum_wavebatch_kernel", ident)
    assert "summed over" in bench.traffic_entry_status(good, "bag_sum_group_kernel", ident)
    old = {k: v for k, v in good.items() if k not in ("lib_sha256", "src_sha256")}
    assert "no library / source hash" in bench.traffic_entry_status(old, "bag_sum_wavebatch_kernel", ident)
    assert bench.traffic_entry_status(None) == "no entry"
    # what the JSON line then looks like: algorithmic figures only, and the reason
    r = bench.roofline_object(138_936_512, 20.0, {"dropped": "collected with another build (...)", "source": "profiles/r03/c2_pmc_summary.txt"}, None)
    assert r["traffic"] is None and "frac_measured" not in r and "another build" in r["traffic_dropped"]
    # the identity of THIS tree is computable without a GPU, and the committed round-4 entries carry hashes of the same shape
    me = bench.library_identity()
    assert me["src_sha256"] and len(me["src_sha256"]) == 64
