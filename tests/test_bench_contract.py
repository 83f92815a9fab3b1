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
    for kw in (dict(batch=2048), dict(tables=16, workload="c3"), dict(streams=2), dict(nbatch=2)):
        assert bench.profile_key(_args(**kw), dict(L=1, dist="uniform")) is None
    assert bench.profile_key(_args(workload="c3", hot_rows=32), dict(L=32, dist="zipf")) == "c3-hot32"      # the LDS kernel: a profile of its own


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


def test_traffic_entry_is_tied_to_the_kernels_code_and_the_launch_not_to_host_sources():
    """round 5 re-collected its evidence chain four times because every entry was tied to hashes that include
    pimemb_engine.cpp / pimemb_shard.cpp -- host-only edits orphaned the PMC bytes of kernels they cannot change.  Entries
    collected from round 6 on carry the sha256 of the dominant kernel's machine code and the launch's signature (kernel kind, grid,
    XCD map, counts); sharded (dist-*) entries the code of the step's kernel families + pimemb_shard.cpp."""
    entry = dict(traffic_bytes_per_launch=100, read_bytes=40, write_bytes=60, source="profiles/r06/c2_pmc_summary.txt", round="r06",
                 lib_sha256="a" * 64, src_sha256="b" * 64, kernel="void pimemb::bag_sum_wavebatch_kernel<unsigned int, 0, 4, ...>(...)",
                 kernel_symbol="_ZN6pimemb24bag_sum_wavebatch_kernelIjLi0ELi4E...", kernel_sha256="c" * 64, launch_signature="00ff" * 4)
    same_launch = dict(kernel_sha256="c" * 64, launch_signature="00ff" * 4)
    # a HOST-ONLY edit of pimemb_engine.cpp: another library, other source hashes -- the same kernel code, the same launch: kept
    edited_host = dict(lib_sha256="d" * 64, src_sha256="e" * 64)
    assert bench.traffic_entry_status(entry, "bag_sum_wavebatch_kernel", edited_host, same_launch) is None
    # the kernel was recompiled to other code: dropped, whatever the source hashes say
    why = bench.traffic_entry_status(entry, "bag_sum_wavebatch_kernel", dict(lib_sha256="a" * 64, src_sha256="b" * 64),
                                     dict(kernel_sha256="f" * 64, launch_signature="00ff" * 4))
    assert "machine code differs" in why
    # the same code launched over another grid / map (a host edit that DOES change the launch): dropped
    assert "the launch" in bench.traffic_entry_status(entry, "bag_sum_wavebatch_kernel", edited_host, dict(kernel_sha256="c" * 64, launch_signature="1234" * 4))
    assert "no launch identity" in bench.traffic_entry_status(entry, "bag_sum_wavebatch_kernel", edited_host, None)
    assert "summed over" in bench.traffic_entry_status(entry, "bag_sum_group_kernel", edited_host, same_launch)
    # a sharded (dist-*) entry: its kernels' code + pimemb_shard.cpp; an edit of pimemb_engine.cpp alone keeps it
    dist = dict(traffic_bytes_per_launch=10, source="x", kernel="bag_sum_wavebatch_kernel<...>", shard_kernels_sha256="1" * 64, shard_src_sha256="2" * 64,
                lib_sha256="a" * 64, src_sha256="b" * 64)
    assert bench.traffic_entry_status(dist, "bag_sum", edited_host, dict(shard_kernels_sha256="1" * 64, shard_src_sha256="2" * 64)) is None
    assert "pimemb_shard.cpp" in bench.traffic_entry_status(dist, "bag_sum", edited_host, dict(shard_kernels_sha256="1" * 64, shard_src_sha256="3" * 64))
    # THIS tree: the identities are computable without a GPU (the code object is read out of the built library), an edit of
    # pimemb_engine.cpp's text changes the round-5 style source hash and neither of the round-6 ones
    import shutil
    import tempfile
    me, shard = bench.library_identity(), bench.shard_identity()
    code = bench.launch_identity()["device_code_sha256"]
    assert len(shard["shard_kernels_sha256"]) == 64 and len(code) == 64
    pkg = os.path.join(ROOT, "pim-embedding-lookup_amd")
    with tempfile.TemporaryDirectory() as tmp:
        shutil.copytree(os.path.join(pkg, "csrc"), os.path.join(tmp, "pim-embedding-lookup_amd", "csrc"))
        os.makedirs(os.path.join(tmp, "pim-embedding-lookup_amd", "lib"))
        shutil.copy(os.path.join(pkg, "lib", "libpimemb.so"), os.path.join(tmp, "pim-embedding-lookup_amd", "lib"))
        with open(os.path.join(tmp, "pim-embedding-lookup_amd", "csrc", "pimemb_engine.cpp"), "a") as f:
            f.write("\n// a host-only edit\n")
        root = bench.ROOT
        try:
            bench.ROOT = tmp
            assert bench.library_identity()["src_sha256"] != me["src_sha256"]
            assert bench.shard_identity() == shard and bench.launch_identity()["device_code_sha256"] == code
        finally:
            bench.ROOT = root


def test_code_object_reader_names_every_kernel_of_a_launch():
    """pim-embedding-lookup_amd/codeobj.py reads the gfx950 code object out of libpimemb.so without any tool: every kernel has a
    code hash and register figures; a Plan.describe() record picks exactly one instantiation; the hash does not move with the
    kernel's position in the library (the descriptor's code-entry offset is masked)."""
    from importlib import import_module
    codeobj = import_module("pim-embedding-lookup_amd.codeobj")
    lib = os.path.join(ROOT, "pim-embedding-lookup_amd", "lib", "libpimemb.so")
    hashes, res = codeobj.kernel_hashes(lib), codeobj.kernel_resources(lib)
    assert len(hashes) > 150 and set(hashes) == set(res)
    c2 = dict(kind=2, dtype=0, itype=0, lanes_per_row=4, ranged=0)            # the metric's launch: two-batch wave-batch, dim 16 fp32, u32
    sym, sha = codeobj.kernel_of_launch(lib, c2)
    assert "bag_sum_wavebatch_kernelIjLi0ELi4E" in sym and sha == hashes[sym]
    assert res[sym]["vgpr"] <= 64 and res[sym]["vgpr_spill"] == 0 and res[sym]["lds"] == 0       # 8 waves / SIMD, no spills
    seen = set()
    for kind in (0, 1, 2, 4):
        for itype in (0, 1):
            for dtype in (0, 1, 2):
                for ranged in ((0, 1) if kind in (0, 2) else (0,)):
                    for lpr in ((1, 2, 4) if kind == 2 else (1, 2, 4, 8, 16, 32, 64)):
                        s, _ = codeobj.kernel_of_launch(lib, dict(kind=kind, dtype=dtype, itype=itype, lanes_per_row=lpr, ranged=ranged))
                        assert s not in seen
                        seen.add(s)
    for itype in (0, 1):
        for dtype in (0, 1, 2):
            for vec in (0, 1):
                seen.add(codeobj.kernel_of_launch(lib, dict(kind=3, dtype=dtype, itype=itype, lanes_per_row=0, anydim_vec=vec, ranged=0))[0])
    assert len(seen) > 200
    # int64 ranged instantiations exist (the sharded call takes DLRM's int64 tensors in place)
    assert "IlLi0ELi32E" in codeobj.kernel_of_launch(lib, dict(kind=0, dtype=0, itype=1, lanes_per_row=32, ranged=1))[0]


def test_roofline_object_names_the_roof_that_binds():
    """every line carries `binding` and a `frac_binding` <= 1 on THAT roof; the algorithmic HBM `frac` stays."""
    # c2: HBM-side traffic 114 MB in 20 us = 0.71 of 8 TB/s; L2 0.33: HBM binds
    r = bench.roofline_object(138_936_512, 20.0, dict(traffic_bytes_per_launch=114_309_676, read_bytes=48_869_523, source="x", tcc_hit=894_895,
                                                     tcc_miss=892_590, ta_busy_frac=0.41), None, alg_read_bytes=73_554_624)
    assert r["binding"] == "hbm" and abs(r["frac_binding"] - r["frac_measured"]) < 1e-12 and r["binding_peak"] == 8000.0 and r["frac_binding"] <= 1
    # c3 (Zipf, cache-served): algorithmic frac 2.5, HBM-side 0.60, L2 0.37, TA busy 0.86 -> the L1 / TA path binds
    r3 = bench.roofline_object(13_390_000_000, 668.0, dict(traffic_bytes_per_launch=3_170_800_000, read_bytes=2_768_000_000, source="x", tcc_hit=40_428_845,
                                                          tcc_miss=24_771_780, ta_busy_frac=0.86), 1_624_000_000, meta_bytes=104_000_000,
                               alg_read_bytes=12_987_000_000)
    assert r3["frac"] > 2 and r3["bound"] == "hbm"                      # the contract's figure is untouched
    assert r3["binding"] == "l1_ta" and abs(r3["frac_binding"] - 0.86) < 1e-9 and r3["ta_busy"] == 0.86
    assert abs(r3["binding_peak"] - 64 * 256 * 2.4) < 1e-6 and "TA_BUSY" in r3["binding_basis"] and "binding" in r3["basis"]
    assert set(r3["roofs"]) == {"hbm", "l2", "l1_ta", "launch"} and all(0 <= v <= 1 for v in r3["roofs"].values())
    # c3 with uniform indices: TA busy 0.85 as well -- but 13.4 GB of HBM-side traffic in 2.29 ms = 5.86 TB/s = 0.73 of the spec, which is
    # what this chip sustains for gathered rows: the HBM binds (the busy counter says requests are in flight, not that the TA is the limit)
    r3u = bench.roofline_object(13_390_000_000, 2288.0, dict(traffic_bytes_per_launch=13_402_323_942, read_bytes=12_999_670_758, source="x", tcc_hit=3_000_000,
                                                            tcc_miss=101_000_000, ta_busy_frac=0.85), None, alg_read_bytes=12_987_000_000)
    assert r3u["binding"] == "hbm" and 0.72 < r3u["frac_binding"] < 0.75 and r3u["roofs"]["l1_ta"] == 0.85
    # the same entry without a TA counter: the bytes the lanes were handed against 64 B/clk/CU (0.49) -- HBM (0.59) then binds
    r3b = bench.roofline_object(13_390_000_000, 668.0, dict(traffic_bytes_per_launch=3_170_800_000, read_bytes=2_768_000_000, source="x", tcc_hit=40_428_845,
                                                           tcc_miss=24_771_780), None, alg_read_bytes=12_987_000_000)
    assert r3b["binding"] == "hbm" and 0.45 < r3b["roofs"]["l1_ta"] < 0.55
    # cache-served and no profile: not attributed -- never a fraction above 1 printed as a utilisation
    r0 = bench.roofline_object(13_390_000_000, 668.0, None, None)
    assert r0["frac"] > 1 and r0["binding"] is None and r0["frac_binding"] is None and "not attributed" in r0["binding_note"]
    # a 3.6-us launch of 26 bags (c1): the launch itself binds
    r1 = bench.roofline_object(26 * 136, 3.6, None, None)
    assert r1["binding"] == "launch" and abs(r1["frac_binding"] - 2.7 / 3.6) < 1e-9
    # no profile, not cache-served: the algorithmic figure is the HBM fraction
    r2 = bench.roofline_object(138_936_512, 20.0, None, None)
    assert r2["binding"] == "hbm" and abs(r2["frac_binding"] - r2["frac"]) < 1e-12


def test_cpu_baseline_states_its_conditions():
    """cpu_baseline carries what the host granted (cgroup quota, affinity, load) as SCALAR members, and every
    figure is the best of three short legs."""
    c = bench.cpu_conditions()
    assert set(c) >= {"host_cpus", "cpu_quota", "affinity_cpus", "loadavg_1m"} and c["host_cpus"] >= 1
    assert 1 <= bench.usable_cores() <= 16
    calls = []
    best, n, el, legs = bench.best_of_legs(lambda: calls.append(1), 0.03)
    assert len(legs) == 3 and best == max(legs) and n == len(calls) and el > 0
    import numpy as np
    import pim_embedding_lookup_amd as pel
    rng = np.random.default_rng(0)
    tabs = [rng.standard_normal((100, 16)).astype(np.float32) for _ in range(2)]
    idx = [rng.integers(0, 100, size=64).astype(np.uint32) for _ in range(2)]
    off = [np.arange(64, dtype=np.uint32) for _ in range(2)]
    b = bench.cpu_baseline(pel, tabs, (idx, off), 0.2)
    scalars = {k for k, v in b.items() if isinstance(v, (int, float, str, bool, type(None)))}
    assert {"value", "unit", "cores", "kind", "cpu_quota", "loadavg_1m", "threads_granted", "affinity_cpus", "host_cpus", "one_thread",
            "all_threads_value", "torch_value", "legs_spread", "sample"} <= scalars
    assert b["kind"] == "port" and b["value"] > 0 and "best of 3 legs" in b["sample"]


def _recorded_legs():
    """The legs of round 5's six-rank run of the driver's command (profiles/r05/six_ranks/driver_default.json), as dist_bench
    hands them to the line assembly: the replica leg's dict and the sharded leg's."""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05", "six_ranks", "driver_default.json")))
    replica = {k: d[k] for k in ("metric", "unit", "value", "ms_per_step", "clock", "ms_per_step_event", "ms_per_step_sync", "value_event", "value_sync",
                                 "n_gpus", "steps", "warmup", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")}
    replica["config"] = {k: v for k, v in d["config"].items() if k not in ("exchange", "exchange_peer")}
    replica["roofline"] = {k: v for k, v in d["roofline"].items() if k not in ("exchange", "exchange_peer")}
    se = d["sharded_exchange"]
    sec = dict(replica, value=se["value"], ms_per_step=se["ms_per_step"], ms_per_step_event=se["ms_per_step_event"], ms_per_step_sync=se["ms_per_step_sync"],
               value_sync=se["value"], value_event=se["value"])
    sec["config"] = {"workload": se["config"], "tables": 26, "dim": 16, "shard_mode": "whole", "exchange_transport": d["exchange_transport"],
                     "placement": {"replicated": 21, "whole": 5, "row_split": 0, "rules": ["x"]}, "last_step_sharded_outputs_sha1": d["config"]["exchange"]["last_step_sharded_outputs_sha1"],
                     "exchange": {k: v for k, v in d["config"]["exchange"].items() if k not in ("what",)}, "parallelism": "ONE library call per batch"}
    sec["roofline"] = dict(se["roofline"])
    return d, replica, sec


def test_the_n_gt_1_line_survives_the_drivers_record():
    """round 5's N > 1 line kept the all-to-all curve in nested objects and extra top-level keys; a BENCH /
    SCALE record keeps the scalar members of config / roofline / cpu_baseline, the NAMES of other keys and the last ~2.3 kB of
    the text -- `value` (the replica curve: N x by construction) survived, value_exchange did not.  Now: (a) `value` IS the sharded
    RCCL leg, the replica figure rides in config.replica_value; (b) every sharded-leg figure -- RCCL and peer stores -- is a scalar
    member of config / roofline; (c) the contract's keys close the line.  Recorded legs of round 5's six-rank run through the new
    assembly (dist_bench.headline_from_legs / driver_proof), then through a stand-in of the driver's parser."""
    from importlib import import_module
    db = import_module("pim-embedding-lookup_amd.dist_bench")
    d, replica, sec = _recorded_legs()
    res = db.headline_from_legs(replica, sec, "whole")
    # what peer_leg records on the same line
    res["exchange_peer"] = res["config"]["exchange_peer"] = dict(d["exchange_peer"])
    res["roofline"]["exchange_peer"] = dict(d["roofline"]["exchange_peer"])
    res["value_exchange_peer"], res["ms_per_step_exchange_peer"] = d["value_exchange_peer"], d["ms_per_step_exchange_peer"]
    res["exchange_transport_peer"], res["exchange_same_bits"], res["verified"] = d["exchange_transport_peer"], True, True
    line = json.dumps(db.driver_proof(res))
    rec = db.driver_record_stand_in(line)
    p, c, r = rec["parsed"], rec["parsed"]["config"], rec["parsed"]["roofline"]
    se, pe = d["sharded_exchange"], d["exchange_peer"]
    # (a) the headline is the sharded RCCL leg; the replica curve is a scalar beside it
    assert p["value"] == se["value"] and p["ms_per_step"] == se["ms_per_step"] and p["n_gpus"] == 6 and p["scaling"] == "weak"
    assert c["replica_value"] == d["value"] and c["replica_ms_per_step"] == d["ms_per_step"] and r["replica_frac"] == d["roofline"]["frac"]
    assert "HEADLINE" in c["workload"] and "replica" in c["workload"] and "RCCL" in c["workload"]
    # (b) every figure of both sharded legs, as scalars the record keeps
    assert c["exchange_value"] == se["value"] and c["exchange_ms_per_step"] == se["ms_per_step"] and c["exchange_mode"] == "whole"
    assert c["exchange_transport"] == "RCCL groups issued from C" and c["exchange_verified"] is True and c["exchange_steps"] == 20
    assert c["exchange_bytes_out_per_rank_per_step"] == 13202304 and c["exchange_host_us_per_step"] > 0 and len(c["exchange_sha1"]) == 40
    assert c["exchange_peer_value"] == pe["value"] and c["exchange_peer_ms_per_step"] == pe["ms_per_step"] and c["exchange_same_bits"] is True
    assert "peer stores" in c["exchange_peer_transport"] and c["exchange_peer_verified"] is True
    rx, rp = d["roofline"]["exchange"], d["roofline"]["exchange_peer"]
    for k in ("step_frac", "step_GBps", "xgmi_GBps", "xgmi_frac", "host_us_per_step", "host_wait_counts_us_per_step"):
        assert r["exchange_" + k] == rx[k] and r["exchange_peer_" + k] == rp[k], k
    assert r["kernel_lookup_us"] == se["roofline"]["kernels"]["lookup_us"]
    assert not any(isinstance(v, (dict, list)) for v in list(c.values()) + list(r.values()))        # (the stand-in dropped them, as the driver does)
    # (c) the contract's keys and the roofline's scalar block close the line: they are inside the kept tail
    tail = rec["tail"]
    for k in ("metric", "value", "unit", "n_gpus", "ms_per_step", "scaling", "exchange_xgmi_GBps", "exchange_peer_xgmi_GBps", "exchange_step_frac", "replica_frac"):
        assert '"%s"' % k in tail, k
    assert line.rstrip().endswith('"verified": true}')
    # a peer leg that could not come up is a scalar too
    res2 = db.headline_from_legs(replica, sec, "whole")
    res2["config"]["exchange_peer"] = {"skipped": "no fine-grained arena"}
    c2 = db.driver_record_stand_in(json.dumps(db.driver_proof(res2)))["parsed"]["config"]
    assert c2["exchange_peer_skipped"] == "no fine-grained arena" and "exchange_peer_value" not in c2
