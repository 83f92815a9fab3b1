#!/usr/bin/env python3
"""bench.py -- pooled-lookups/sec + achieved HBM GB/s of the embedding-lookup hot path.

    python bench.py --gpus 1 --steps K --warmup W          (default: N=1, finishes in minutes)
    python bench.py --gpus N ...                           (self-launching: this process touches no GPU, starts one
                                                            child per rank, relays rank 0's JSON line, returns the worst
                                                            child's exit status -- the reference's lookup() likewise
                                                            fans out to every device from one call, emb_host.h:258-321)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...   (launched ranks: RANK /
                                                            WORLD_SIZE in the environment, no further fan-out)

A "step" is one pass of the hot path over one batch of synthetic input: the fused multi-table
EmbeddingBag(sum) lookup (libpimemb.so, HIP) for the 26 Criteo-Kaggle tables, dim 16 fp32,
B = 39292 bags per table, one index per bag (BASELINE.json configs[1], SURVEY.md section 8 row D "C2").
Inputs are resident in HBM before the timed region; NBATCH distinct index batches and output
buffers are rotated so a step never re-reads the previous step's rows out of the 256 MiB
Infinity Cache (working set per rotation > 1 GB).

N > 1 (one process per GPU, torch.distributed over RCCL; pim-embedding-lookup_amd/dist_bench.py): each
rank owns B bags per table (weak scaling, global batch N * B).  Tables are replicated while the whole
set fits a quarter of one GPU's HBM -- the Kaggle tables do, so the metric's config needs no exchange --
and sharded by table id / row range with indices in / pooled rows out otherwise (ONE
library call per batch, emb_shard_*); the sharded exchange is measured in the same run as a secondary leg -- over BOTH
transports by default (--exchange both): grouped ncclSend / ncclRecv issued from C (value_exchange), then the
collective-free peer-store exchange (value_exchange_peer, or exchange_peer = {"skipped": reason} if it cannot come up),
same expected rows, same bits.  Before a sharded leg allocates anything it adds up the fullest rank's HBM
(sharding.hbm_budget) and shrinks a layout that does not fit (config.rows_scale_to_fit).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # CPU baselines: idle OpenMP threads must not spin

import numpy as np  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--batch", type=int, default=None, help="bags per table per rank (default 39292)")
    ap.add_argument("--nbatch", type=int, default=8, help="distinct index batches rotated through")
    ap.add_argument("--index-dist", choices=["uniform", "zipf"], default=None,
                    help="default: uniform for c2, zipf(1.2) for c3")
    ap.add_argument("--index-order", choices=["drawn", "sorted"], default="drawn",
                    help="N=1 experiment (REJECTED_EXPERIMENTS, round 5): 'sorted' hands every table's indices over in ascending row "
                         "order -- the IDEAL outcome of any locality bucketing of a one-index-per-bag launch (every XCD then sees a "
                         "disjoint row range, every distinct row is fetched by one L2 only) without paying for the bucketing")
    ap.add_argument("--workload", choices=["c1", "c2", "c3", "c4", "c5"], default="c2",
                    help="c1 = configs[0]'s shape on the GPU (the 26 Kaggle tables, mini-batch 1: launch latency); "
                         "c2 = BASELINE configs[1] (the metric's config, default); c3 = configs[2] scaled "
                         "to fit one GPU: 48 tables x 10M rows x dim 128 fp32, B=16384, pooling 32; c5 = one GPU's "
                         "share of configs[4]: 64 tables x 30M rows x dim 64 fp16, mixed Zipf/uniform, pooling 32; "
                         "c4 = configs[3], the Terabyte-shaped 26 tables at dim 128: with --gpus N the row-range "
                         "sharded exchange, at N=1 the lookups one of 8 ranks serves (its row shards + small tables)")
    ap.add_argument("--rows-scale", type=float, default=1.0,
                    help="N>1: shrink every table of the set by this factor (rehearse an 8-rank layout on fewer GPUs)")
    ap.add_argument("--pooling", type=int, default=None, help="c4 at N=1, and the N>1 legs (data-parallel / --shard-mode whole): indices per bag "
                         "(default 1; the exchange is link-bound for 1 and index-volume-bound for 32)")
    ap.add_argument("--tables", type=int, default=None, help="c3: number of tables (default 48)")
    ap.add_argument("--replicate-mb", type=int, default=None,
                    help="N>1: tables up to this size are replicated on every rank, larger ones are sharded. "
                         "Default (auto): replicate everything when the whole table set fits a quarter of one "
                         "GPU's HBM (the Kaggle config does), else 64")
    ap.add_argument("--no-exchange-leg", action="store_true",
                    help="N>1, auto policy: skip the secondary sharded-exchange measurement")
    ap.add_argument("--shard-mode", choices=["whole", "rows", "plan"], default=None,
                    help="N>1: big tables placed whole on owner ranks (default for c2: no routing kernel, no copy), split "
                         "by row range over all ranks with GPU-side request routing (balanced xGMI egress; default for c4), or "
                         "whatever the shard planner decides (plan)")
    ap.add_argument("--checked", action="store_true",
                    help="sharded legs: create the shard with EMB_SHARD_CHECK_SERVED (one-index batches keep the direct path and count "
                         "what every shard serves; routed batches validate what they serve)")
    ap.add_argument("--ids", choices=["uint32", "int64"], default="uint32",
                    help="sharded legs: index dtype handed to the library -- uint32 (the reference's width, emb_host.h:234) or int64 "
                         "(DLRM's tensors, used in place: emb_shard_input.index_type)")
    ap.add_argument("--exchange", choices=["rccl", "peer", "both"], default=None,
                    help="N>1 sharded legs: how pieces travel between ranks -- grouped ncclSend/ncclRecv issued from C (rccl), "
                         "the collective-free exchange (peer: HIP IPC mappings, the owner gathers a requester's indices in place "
                         "and stores pooled rows straight into its HBM; handshake through a shared-memory segment, no RCCL in the data path), "
                         "or BOTH in one run (the default for N > 1): the RCCL leg is value_exchange, the peer-store leg is brought up under a "
                         "deadline afterwards and reported as value_exchange_peer -- or {skipped: reason} if it cannot come up, which does not "
                         "fail the run; both legs are verified against the same expected rows and must leave the same bits.  N = 1: rccl")
    ap.add_argument("--collective", choices=["native"], default="native",
                    help="kept for command-line compatibility: the sharded legs' transfers are always grouped ncclSend/ncclRecv "
                         "issued from the C side (emb_comm_exchange inside emb_shard_*)")
    ap.add_argument("--streams", type=int, default=1,
                    help="N=1: round-robin the independent steps over this many HIP streams (default 1: every "
                         "step on one stream, which is what roofline.kernel_us assumes)")
    ap.add_argument("--hot-rows", type=int, default=0,
                    help="pooled workloads (c3/c5/c4 --pooling): hint each table's K most frequent rows of the first "
                         "batch to the engine (emb_set_hot_rows: served from LDS) when they cover >= 5 %% of "
                         "that table's accesses; 0 = no hint")
    ap.add_argument("--prewarm-ms", type=float, default=250.0,
                    help="untimed device pre-warm before the W warm-up steps (clock ramp of a fresh process; N=1 and the data-parallel N>1 leg); 0 = off")
    ap.add_argument("--graph", action="store_true",
                    help="N=1: capture the K timed launches in one hipGraph and replay it (tried in round 3: see DESIGN.md section 5)")
    ap.add_argument("--coalesce", type=int, default=0,
                    help="N=1: serve the workload's batch as R independent REQUESTS per step through the request queue "
                         "(emb_queue_*: R x add + ONE flush = one fused launch) instead of one prepared plan per step -- the "
                         "reference's serving shapes (--workload c1: mini-batch 1; --batch 32 / 512) are launch-bound one by one; "
                         "0 = off.  The line also carries the back-to-back figure (R separate launches) it is to be compared with")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=10.0)
    return ap.parse_args()


def make_tables_on_gpu(torch, eng, rows_list, dim, device, seed=0, keep_host=False, dtype="f32"):
    """W_t ~ U(-sqrt(1/N_t), sqrt(1/N_t)), generated on the GPU and handed to the engine
    device-to-device (no 2 GB host round trip)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    host = []
    for t, n in enumerate(rows_list):
        a = float(np.sqrt(1.0 / n))
        w = torch.empty((n, dim), dtype=torch.float32, device=device)
        w.uniform_(-a, a, generator=g)
        if dtype == "f16":
            w = w.to(torch.float16)
        eng.load_table(t, w)
        if t < int(keep_host):
            host.append(w.float().cpu().numpy())   # the oracle's fp16 path is exercised by tests; baseline in fp32
        del w
    torch.cuda.empty_cache()
    return host


L2_PEAK_GBS = 34500.0    # MI355X_MICROARCH.md "L2 (per XCD)": ~34.5 TB/s aggregate
PROFILED_NBATCH = 8      # the --nbatch every entry of profiles/traffic.json was collected with


def profile_key(args, spec):
    """Key of this run in profiles/traffic.json, or None when the run is not one of the profiled commands
    (another batch size / table count / hint changes the traffic)."""
    if args.batch is not None or args.tables is not None or args.streams != 1 \
            or getattr(args, "nbatch", PROFILED_NBATCH) != PROFILED_NBATCH:     # (another rotation length = another index reuse)
        return None
    key = args.workload
    if args.workload == "c4" and spec["L"] != 1:
        key += "-l%d" % spec["L"]
    default_dist = {"c1": "uniform", "c2": "uniform", "c3": "zipf", "c4": "uniform", "c5": "mixed"}[args.workload]
    if spec["dist"] != default_dist:
        key += "-" + spec["dist"]
    if getattr(args, "index_order", "drawn") != "drawn":
        key += "-" + args.index_order
    if args.hot_rows:                  # rows served from LDS: another kernel, other traffic (profiled for c3 --hot-rows 32)
        key += "-hot%d" % args.hot_rows
    return key


KERNEL_OF_KIND = ["bag_sum_wavebatch_kernel", "bag_sum_group_kernel", "bag_sum_wavebatch_kernel", "bag_sum_anydim", "bag_sum_hot_kernel"]
# rounds 4-5: an entry was tied to these sources (and to the library's bytes).  Host-only edits of the last two orphaned every
# entry -- four re-collections in round 5.  Entries collected from round 6 on are tied to the DEVICE CODE and the LAUNCH instead
# (launch_identity below); the source hash is still recorded, and still honoured for the old entries.
KERNEL_SOURCES = ["pimemb_bag_kernels.h", "pimemb_kernels.hip", "pimemb_xcd_map.h", "pimemb_internal.h", "pimemb_engine.cpp", "pimemb_shard.cpp"]
SHARD_SOURCES = ["pimemb_shard.cpp"]          # what a dist-* entry is tied to besides its kernels' code (the sharded step's launch logic)
SHARD_KERNEL_FAMILIES = ("bag_sum_wavebatch_kernel", "bag_sum_group_kernel", "route_bags", "route_onehot", "unroute_bags", "served_counts_kernel",
                         "publish_words_kernel", "peer_post_kernel", "peer_done_kernel")


def _pkg_path(*parts):
    return os.path.join(ROOT, "pim-embedding-lookup_amd", *parts)


def library_identity():
    """sha256 of the loaded libpimemb.so and of the sources its kernels and launch logic are built from -- what entries of
    rounds 4-5 were tied to (a rebuild of unchanged sources may differ in bytes; changed sources never match)."""
    import hashlib
    out = {"lib_sha256": None, "src_sha256": None}
    try:
        with open(_pkg_path("lib", "libpimemb.so"), "rb") as f:
            out["lib_sha256"] = hashlib.sha256(f.read()).hexdigest()
    except OSError:
        pass
    try:
        h = hashlib.sha256()
        for name in KERNEL_SOURCES:
            with open(_pkg_path("csrc", name), "rb") as f:
                h.update(name.encode() + b"\0" + f.read())
        out["src_sha256"] = h.hexdigest()
    except OSError:
        pass
    return out


def launch_identity(plan=None, lib_path=None):
    """What a profile of a single-GPU launch is tied to, by construction: the machine code of the kernel instantiation the launch
    runs (sha256 of its text + kernel descriptor out of the library's gfx950 code object: pim-embedding-lookup_amd/codeobj.py)
    and the launch itself (emb_plan_signature: kernel kind, grid, XCD map word for word, per-descriptor counts -- no address).
    A host-only edit of the engine leaves both unchanged (the profile stays valid); a changed kernel, grid or map changes one of
    them (the profile is dropped, with the reason on the line).  plan = None: the code-object half only."""
    from importlib import import_module
    codeobj = import_module("pim-embedding-lookup_amd.codeobj")
    lib_path = lib_path or _pkg_path("lib", "libpimemb.so")
    out = {"device_code_sha256": codeobj.device_code_sha256(lib_path)}
    if plan is not None:
        launches = plan.describe()
        top = max(launches, key=lambda g: g["grid"])          # the dominant launch of the plan (one launch for every BASELINE shape)
        sym, sha = codeobj.kernel_of_launch(lib_path, top)
        out.update(kernel_symbol=sym, kernel_sha256=sha, launch_signature="%016x" % plan.signature(), launches=len(launches))
    return out


def _demangled_symbols(hashes):
    """{demangled name: mangled symbol} of the library's kernels (c++filt: binutils, on this image and on the GPU boxes), or None."""
    import subprocess
    syms = sorted(hashes)
    try:
        out = subprocess.run(["c++filt"], input="\n".join(syms), capture_output=True, text=True, timeout=30)
        names = out.stdout.splitlines()
        return dict(zip(names, syms)) if out.returncode == 0 and len(names) == len(syms) else None
    except (OSError, subprocess.SubprocessError):
        return None


def shard_identity(kernel_names=None, lib_path=None):
    """What a dist-* entry (the sharded step at world 1: several kernels, launched by pimemb_shard.cpp) is tied to: one sha256 over
    the code of the kernels the profiled step RAN -- `kernel_names`: their demangled names as rocprofv3's kernel trace lists them
    (the entry keeps the list) -- and the sha256 of pimemb_shard.cpp.  Without names (or without c++filt): over every kernel of the
    families a sharded step can run, which also moves when an instantiation the step never launches changes."""
    import hashlib
    from importlib import import_module
    codeobj = import_module("pim-embedding-lookup_amd.codeobj")
    hashes = codeobj.kernel_hashes(lib_path or _pkg_path("lib", "libpimemb.so"))
    picked, basis = None, "families"
    if kernel_names:
        by_name = _demangled_symbols(hashes)
        if by_name is not None:
            norm = lambda n: n.replace(" ", "")
            table = {norm(k): v for k, v in by_name.items()}
            found = [table.get(norm(n)) for n in kernel_names]
            if all(found):
                picked, basis = sorted(found), "kernels of the profiled step"
    if picked is None:
        picked = sorted(sym for sym in hashes if any(f in sym for f in SHARD_KERNEL_FAMILIES))
    h = hashlib.sha256()
    for sym in picked:
        h.update(sym.encode() + b"\0" + hashes[sym].encode())
    hs = hashlib.sha256()
    for name in SHARD_SOURCES:
        with open(_pkg_path("csrc", name), "rb") as f:
            hs.update(name.encode() + b"\0" + f.read())
    return {"shard_kernels_sha256": h.hexdigest(), "shard_src_sha256": hs.hexdigest(), "shard_kernels_basis": basis, "shard_kernels_hashed": len(picked)}


def traffic_entry_status(entry, kernel_name=None, ident=None, launch=None):
    """None when the committed PMC entry may be used for this run, else the reason it must not be.
    Round-6 entries (they carry kernel_sha256 / launch_signature, or shard_kernels_sha256 for the sharded legs): the kernel's CODE
    and the LAUNCH of this run (`launch` = launch_identity(plan) / shard_identity()) must be the ones the counters were collected
    on -- whatever the host sources or the library's bytes are now.  Older entries: the library's or the kernel sources' hash must
    match (an entry keyed by workload name only mis-prices frac_measured silently after a kernel change).  Either way
    the launch's dominant kernel must be the family the counters were summed over."""
    if not entry:
        return "no entry"
    if kernel_name and entry.get("kernel") and kernel_name not in entry["kernel"]:
        return "the counters were summed over %s, this launch runs %s" % (entry["kernel"][:60], kernel_name)
    if entry.get("kernel_sha256") or entry.get("shard_kernels_sha256"):
        if launch is None:
            return "no launch identity for this run (the entry is tied to the kernel's code and the launch signature)"
        for k, what in (("kernel_sha256", "the kernel's machine code"), ("launch_signature", "the launch (grid / XCD map / counts)"),
                        ("shard_kernels_sha256", "the sharded step's kernels"), ("shard_src_sha256", "pimemb_shard.cpp")):
            if entry.get(k) and entry[k] != launch.get(k):
                return "%s differs from the profiled one (%s.. vs %s..)" % (what, str(entry[k])[:12], str(launch.get(k))[:12])
        return None
    ident = ident or library_identity()
    have = [entry.get("lib_sha256"), entry.get("src_sha256")]
    if not any(have):
        return "the entry records no library / source hash (collected before round 4): re-collect it"
    if not ((entry.get("lib_sha256") and entry["lib_sha256"] == ident["lib_sha256"]) or
            (entry.get("src_sha256") and entry["src_sha256"] == ident["src_sha256"])):
        return "collected with another build (library %s.., sources %s..; loaded: %s.., %s..)" % (
            str(entry.get("lib_sha256"))[:10], str(entry.get("src_sha256"))[:10], str(ident["lib_sha256"])[:10], str(ident["src_sha256"])[:10])
    return None


def measured_traffic(key, kernel_name=None, launch=None):
    """This command's entry of profiles/traffic.json: HBM-side bytes per launch of the dominant kernel from the
    committed rocprofv3 PMC passes (collected and corrected as MI355X_MICROARCH.md "HBM" prescribes), L2 hit / miss
    requests per launch.  None when no profile exists for the command being run; {"dropped": reason} when one exists but
    belongs to another kernel, another launch or another build (traffic_entry_status)."""
    if key is None:
        return None
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            entry = json.load(f).get(key)
    except (OSError, ValueError):
        return None
    if entry is None:
        return None
    if callable(launch):          # (a sharded entry names the kernels its step ran: the identity is computed over those)
        launch = launch(entry)
    why = traffic_entry_status(entry, kernel_name, launch=launch)
    return entry if why is None else {"dropped": why, "source": entry.get("source")}


def unique_row_bytes(batch, row_bytes):
    """Compulsory table bytes of one launch: every DISTINCT row of every table once (host-side count).  Read
    traffic well above this means rows fetched more than once (several private L2s, or evicted and re-read)."""
    idx, _off = batch
    return int(sum(np.unique(i).shape[0] for i in idx)) * row_bytes


def unique_line_bytes(batch, row_bytes, line=128):
    """The same at the granularity the memory system fetches at: every DISTINCT 128-byte line that holds a piece of a named
    row, once (gfx950 has no 32 / 64-byte sector fetch: tools/sector_probe.hip).  For rows narrower than a line -- dim 16 fp32 =
    64 B -- this, not unique_row_bytes, is the floor of the table traffic: two rows share a line only when they are neighbours."""
    idx, _off = batch
    total = 0
    for i in idx:
        if row_bytes >= line:        # (rows of k x 128 B are line-aligned: tables are 256-byte aligned allocations)
            total += np.unique(i).shape[0] * -(-row_bytes // line)
        else:                        # the line holding the row's first byte (64-byte rows never straddle)
            total += np.unique(i.astype(np.int64) * row_bytes // line).shape[0]
    return int(total) * line


L1_TA_PEAK_GBS = 64 * 256 * 2.4      # MI355X_MICROARCH.md: a CU's vector L1 / texture-address path delivers 64 B/clk; 256 CUs, ~2.4 GHz
HBM_SUSTAINED_FRAC = 0.70            # MI355X_MICROARCH.md "HBM" / "Indexed rows": ~6.3 TB/s streaming, 5.5-5.8 TB/s for random rows fetched
                                     # once, of the 8 TB/s spec: a launch whose MEASURED HBM-side rate is there is HBM-bound, whatever else is busy
LAUNCH_FLOOR_US = 2.7                # an empty kernel of the same grid, back to back on one stream (2.5-2.9 us)


def binding_roof(r, entry, alg_read_bytes, kernel_us):
    """Which roof binds the launch, from the committed counters -- so that every line carries a fraction <= 1 on the roof it is
    quoted against (round 5 printed bound "hbm" with frac 2.53 for the cache-served C3 launch).  Candidates, each as a fraction
    of its own peak: `hbm` -- the MEASURED HBM-side bytes against 8 TB/s; `l2` -- the L2's requests x 128 B against 34.5 TB/s;
    `l1_ta` -- the texture-address / vector-L1 path: its own busy counter (TA_BUSY_avr / kernel cycles) where the profile has one,
    else the bytes the lanes were handed against 64 B/clk/CU; `launch` -- an empty kernel of the same grid takes ~2.7 us.  A
    launch whose measured HBM-side rate has reached what this chip sustains for gathered rows (>= 0.70 of the 8 TB/s spec: 5.6 TB/s)
    is HBM-bound whatever else is busy -- the TA's busy counter says that requests are in flight, not that it is the limit: the
    uniform C3 launch shows TA busy 0.85 at 5.86 TB/s of HBM traffic, and its pace is the HBM's -- otherwise the
    largest fraction binds.  Without a counter profile only `launch` and the algorithmic HBM figure exist: a fraction above 1 is then
    marked as unattributed, never printed as a utilisation."""
    t = kernel_us * 1e-6
    cand = {}
    if r.get("frac_measured") is not None:
        cand["hbm"] = (r["frac_measured"], r["achieved_measured"], HBM_PEAK_GBS, "measured HBM-side bytes (profiles/traffic.json) / launch time")
    elif r["frac"] <= 1.0:
        cand["hbm"] = (r["frac"], r["achieved"], HBM_PEAK_GBS, "algorithmic bytes / launch time (no counter profile of this command)")
    if r.get("l2_frac") is not None:
        cand["l2"] = (r["l2_frac"], r["l2_frac"] * L2_PEAK_GBS, L2_PEAK_GBS, "(TCC_HIT + TCC_MISS) x 128 B / launch time")
    if entry and entry.get("ta_busy_frac") is not None:
        cand["l1_ta"] = (min(entry["ta_busy_frac"], 1.0), alg_read_bytes / t / 1e9, L1_TA_PEAK_GBS, "TA_BUSY_avr / kernel cycles (GRBM_GUI_ACTIVE / 8)")
    elif entry and alg_read_bytes:
        cand["l1_ta"] = (alg_read_bytes / t / 1e9 / L1_TA_PEAK_GBS, alg_read_bytes / t / 1e9, L1_TA_PEAK_GBS, "bytes gathered by the lanes / launch time against 64 B/clk/CU")
    cand["launch"] = (min(LAUNCH_FLOOR_US / max(kernel_us, 1e-9), 1.0), None, None, "an empty kernel of the same grid: ~%.1f us" % LAUNCH_FLOOR_US)
    if "hbm" not in cand and len(cand) == 1 and cand["launch"][0] < 0.5:
        r["binding"], r["frac_binding"] = None, None
        r["binding_note"] = "cache-served (algorithmic frac > 1) and no counter profile of this exact command: the binding roof is not attributed"
        return
    name = "hbm" if r.get("frac_measured") is not None and r["frac_measured"] >= HBM_SUSTAINED_FRAC else max(cand, key=lambda k: cand[k][0])
    frac, ach, peak, basis = cand[name]
    r["binding"], r["frac_binding"], r["binding_basis"] = name, frac, basis
    if peak is not None:
        r["binding_achieved"], r["binding_peak"] = ach, peak
    r["roofs"] = {k: v[0] for k, v in cand.items()}


def roofline_object(alg_bytes, kernel_us, entry, uniq_bytes, meta_bytes=0, uniq_line_bytes=None, alg_read_bytes=None):
    """HBM roofline of the dominant kernel, ONE basis on every line: `achieved` / `frac` = ALGORITHMIC bytes per launch
    (cache hits still count) / the launch's duration / the 8 TB/s peak.  A launch that is
    mostly served by L2 / Infinity Cache gathers more than the HBM delivers, so its `frac` can exceed 1: `cache_served`
    says so, and the MEASURED HBM-side rate (committed rocprofv3 PMC passes of the same command, profiles/traffic.json)
    is reported next to it as `achieved_measured` / `frac_measured`, with `l2_frac` pricing the L2 requests against
    the L2's own peak."""
    t = kernel_us * 1e-6
    alg = alg_bytes / t / 1e9
    r = {"bound": "hbm", "achieved": alg, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": alg / HBM_PEAK_GBS,
         "traffic": None, "basis": "algorithmic bytes", "kernel_us": kernel_us, "algorithmic_bytes": alg_bytes,
         "achieved_algorithmic": alg}
    if entry and entry.get("dropped"):
        r["traffic_dropped"] = "profiles/traffic.json entry (%s) not used: %s" % (entry.get("source"), entry["dropped"])
        entry = None
    if entry:
        traffic = entry.get("traffic_bytes_per_launch")
        r["traffic"] = traffic
        r["traffic_source"] = entry.get("source")
        if traffic:
            r["achieved_measured"] = traffic / t / 1e9      # committed profile's bytes / THIS run's launch time
            r["frac_measured"] = r["achieved_measured"] / HBM_PEAK_GBS
            r["cache_served"] = bool(traffic < 0.75 * alg_bytes)
        if entry.get("tcc_hit") is not None and entry.get("tcc_miss") is not None:
            req = entry["tcc_hit"] + entry["tcc_miss"]
            r["l2_hit_rate"] = entry["tcc_hit"] / max(req, 1)
            r["l2_frac"] = req * 128 / t / 1e9 / L2_PEAK_GBS
        if uniq_bytes and entry.get("read_bytes"):     # table bytes read (indices / offsets taken out) per distinct-row byte
            r["read_over_unique_rows"] = max(entry["read_bytes"] - meta_bytes, 0) / uniq_bytes
    if uniq_bytes:
        r["unique_row_bytes"] = uniq_bytes
    if uniq_line_bytes:            # the floor of the table traffic at line granularity; read / it = what bucketing could remove
        r["unique_line_bytes"] = uniq_line_bytes
        if entry and entry.get("read_bytes"):
            r["read_over_unique_lines"] = max(entry["read_bytes"] - meta_bytes, 0) / uniq_line_bytes
    if r["frac"] > 1.0:
        r["basis"] = ("algorithmic bytes; above 1 because most rows were served by L2 / Infinity Cache -- not an HBM "
                      "utilisation" + ("; the HBM-side rate is frac_measured" if r.get("frac_measured") is not None else
                                       "; no PMC profile of this exact command in profiles/traffic.json")
                      + "; the roof that binds the launch and the fraction of IT are `binding` / `frac_binding`")
    if entry and entry.get("ta_busy_frac") is not None:
        r["ta_busy"] = entry["ta_busy_frac"]
    binding_roof(r, entry, alg_read_bytes if alg_read_bytes is not None else alg_bytes, kernel_us)
    return r


def workload_spec(pel, args):
    """rows per table, dim, bags per table, pooling, index distribution, description."""
    if args.workload in ("c1", "c2"):
        B = args.batch or (1 if args.workload == "c1" else pel.workloads.KAGGLE_BATCH)
        dist = args.index_dist or "uniform"
        return dict(rows=pel.workloads.KAGGLE_ROWS, dim=pel.workloads.KAGGLE_DIM, B=B, L=1, dist=dist,
                    name="%s: 26 Criteo-Kaggle tables, dim 16 fp32, B=%d bags/table, L=1, u32 indices+offsets, "
                         "%s indices%s" % (args.workload.upper(), B, dist, ", SORTED per table (ideal bucketing)" if getattr(args, "index_order", "drawn") == "sorted" else ""))
    if args.workload == "c5":
        T = args.tables or 64
        B = args.batch or 16384
        return dict(rows=[30_000_000] * T, dim=64, B=B, L=32, dist="mixed", dtype="f16",
                    name="C5, one GPU's share (512 tables x 50M rows = 3.28 TB does not fit 8 x 288 GB; scaled to "
                         "512 x 30M = 1.97 TB, 64 tables per GPU): %d tables x 30M rows, dim 64 fp16 rows / fp32 "
                         "accumulate, B=%d bags/table, L=32, Zipf(1.2) on even tables, uniform on odd ones" % (T, B))
    if args.workload == "c4":
        # what ONE of 8 ranks serves per step: its 1/8 row range of every table above 64 MiB (requests of
        # all 8 ranks that fall in the range: ~B per table) and the small tables replicated (its own B bags)
        B = args.batch or pel.workloads.TERABYTE_BATCH
        L = args.pooling or 1
        dim = pel.workloads.TERABYTE_DIM
        rows = [(-(-n // 8) if n * dim * 4 > (64 << 20) else n) for n in pel.workloads.TERABYTE_ROWS]
        dist = args.index_dist or "uniform"
        return dict(rows=rows, dim=dim, B=B, L=L, dist=dist,
                    name="C4, one of 8 ranks' share (26 Criteo-Terabyte-shaped tables, 452 GB as written): row "
                         "shards 1/8 of the %d tables above 64 MiB + %d small tables whole, %.1f GB, dim 128 fp32, "
                         "B=%d bags/table, L=%d, %s indices"
                         % (sum(n * dim * 4 > (64 << 20) for n in pel.workloads.TERABYTE_ROWS),
                            sum(n * dim * 4 <= (64 << 20) for n in pel.workloads.TERABYTE_ROWS),
                            sum(rows) * dim * 4 / 1e9, B, L, dist))
    T = args.tables or 48
    B = args.batch or 16384
    dist = args.index_dist or "zipf"
    return dict(rows=[10_000_000] * T, dim=128, B=B, L=32, dist=dist,
                name="C3 scaled to fit 288 GB (as written it needs 327.7 GB): %d tables x 10M rows, dim 128 fp32, "
                     "B=%d bags/table, L=32, u32 indices+offsets, %s indices" % (T, B, dist))


def make_batches(pel, spec, nbatch, seed=1, order="drawn"):
    rng = np.random.default_rng(seed)
    def gen(t):
        if spec["dist"] == "uniform" or (spec["dist"] == "mixed" and t % 2 == 1):
            return pel.workloads.uniform_indices
        return pel.workloads.zipf_indices
    batches = []
    off = pel.workloads.fixed_offsets(spec["B"], spec["L"])
    for _ in range(nbatch):
        idx = [gen(t)(rng, n, spec["B"] * spec["L"]) for t, n in enumerate(spec["rows"])]
        if order == "sorted":      # the ideal outcome of a locality bucketing pre-pass (outputs permute with the bags)
            idx = [np.sort(i) for i in idx]
        batches.append((idx, [off] * len(spec["rows"])))
    return batches


def cpu_conditions():
    """What the host granted this process -- a stated baseline states its conditions (round 4 and round 5 quoted 1.05e9 and 5.2e8
    lookups/s for the same code on "16 of 256 CPUs"; nothing on the line said why): the cgroup CPU quota, the load of the box,
    the affinity mask."""
    out = {"host_cpus": os.cpu_count(), "cpu_quota": None, "affinity_cpus": None, "loadavg_1m": None}
    try:
        out["affinity_cpus"] = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        out["cpu_quota"] = None if quota == "max" else round(int(quota) / int(period), 2)
        out["cpu_quota_raw"] = "%s %s" % (quota, period)
    except (OSError, ValueError):
        pass
    try:
        out["loadavg_1m"] = round(os.getloadavg()[0], 2)
    except OSError:
        pass
    return out


def usable_cores(cap=16):
    """Threads worth starting: the affinity mask, cut to the cgroup CPU quota when there is one and to
    `cap` (a one-GPU box exposes all 256 host CPUs but grants a share of about 16)."""
    c = cpu_conditions()
    n = c["affinity_cpus"] or c["host_cpus"] or 1
    if c["cpu_quota"]:
        n = min(n, max(1, int(c["cpu_quota"])))
    return max(1, min(n, cap))


def best_of_legs(run_once, budget, legs=3):
    """`legs` short timed legs of whole batches within `budget` seconds in all; returns (best rate in calls/s, calls in all,
    seconds in all, rates of every leg) -- the best leg is the baseline (a neighbour's burst on a shared host slows a leg, never
    speeds one up), the spread says how quiet the box was."""
    rates, n_all, el_all = [], 0, 0.0
    for _ in range(legs):
        n, t0 = 0, time.perf_counter()
        while True:
            run_once()
            n += 1
            el = time.perf_counter() - t0
            if el >= budget / legs or n >= 2000:
                break
        rates.append(n / el)
        n_all += n
        el_all += el
    return max(rates), n_all, el_all, rates


def cpu_baseline(pel, host_tables, batch, seconds):
    """The oracle (kind "port": C restatement of the reference loop) timed on this host on a bounded
    sample of the same workload: whole batches for ~`seconds` in all, half of the budget with one thread and half with the bags
    split over the cores this process may use (OpenMP, at most 16), each as the BEST of three short legs.  `value` / `cores` are
    the faster of the two; both are kept in the object, with what the host granted (cpu_quota, affinity, load)."""
    from oracle import oracle
    idx, off = batch
    idx, off = idx[:len(host_tables)], off[:len(host_tables)]
    per_call = sum(o.shape[0] for o in off)

    outs = [np.empty((o.shape[0], host_tables[0].shape[1]), dtype=np.float32) for o in off]   # reused: no page faults in the loop
    cond = cpu_conditions()
    ncores = usable_cores()

    def leg(threads, budget):
        oracle.c_lookup_tables(host_tables, idx, off, threads, outs)      # warm
        return best_of_legs(lambda: oracle.c_lookup_tables(host_tables, idx, off, threads, outs), budget)

    r1, n1, el1, legs1 = leg(1, seconds / 2)
    v1 = r1 * per_call
    rm, nm, elm, legsm = leg(ncores, seconds / 2) if ncores > 1 else (r1, n1, el1, legs1)
    vm = rm * per_call
    best_v, best_c = (vm, ncores) if vm > v1 else (v1, 1)
    res = {"all_threads": {"threads": ncores, "value": vm, "legs": [x * per_call for x in legsm]},
           "one_thread_legs": [x * per_call for x in legs1],
           "value": best_v, "unit": "pooled-lookups/s", "cores": best_c, "kind": "port",
           "one_thread": v1, "all_threads_value": vm, "threads_granted": ncores,
           "cpu_quota": cond["cpu_quota"], "affinity_cpus": cond["affinity_cpus"], "host_cpus": cond["host_cpus"],
           "loadavg_1m": cond["loadavg_1m"], "loadavg_1m_after": cpu_conditions()["loadavg_1m"],
           "legs_spread": (max(legsm) - min(legsm)) / max(legsm) if legsm else None,
           "omp_wait_policy": os.environ.get("OMP_WAIT_POLICY", ""),
           "sample": f"best of 3 legs each: {n1} + {nm} batches of the bench workload restricted to its first {len(host_tables)} tables "
                     f"({off[0].shape[0]} bags/table, {idx[0].shape[0] // max(off[0].shape[0], 1)} indices/bag) "
                     f"through oracle/emb_oracle.c in {el1:.1f} s (1 thread) + {elm:.1f} s ({ncores} OpenMP "
                     f"threads over bags); host has {cond['host_cpus']} cpus, cgroup quota {cond['cpu_quota']}, load {cond['loadavg_1m']}"}
    res["torch"] = cpu_baseline_torch(host_tables, (idx, off), max(seconds * 0.6, 2.0), ncores)
    res["torch_value"], res["torch_threads"] = res["torch"]["value"], res["torch"]["threads"]
    return res


def cpu_baseline_torch(host_tables, batch, seconds, ncores):
    """The CPU path north_star names: torch CPU nn.EmbeddingBag(mode='sum') -- F.embedding_bag looped over
    the tables, as dlrm_s_pytorch.py::apply_emb does (README.md:6,10,14 of the reference; the file itself
    is an empty submodule).  Same tables / indices / offsets as the "port" legs, int64 indices (torch's
    type), output tensors allocated by torch per call; 1 thread and `ncores` threads, OMP_WAIT_POLICY=passive."""
    import torch
    import torch.nn.functional as F
    idx, off = batch
    ws = [torch.from_numpy(w) for w in host_tables]
    ii = [torch.from_numpy(i.astype(np.int64)) for i in idx]
    oo = [torch.from_numpy(o.astype(np.int64)) for o in off]
    per_call = sum(o.shape[0] for o in off)

    def one():
        return [F.embedding_bag(i, w, o, mode="sum") for w, i, o in zip(ws, ii, oo)]

    def leg(threads, budget):
        torch.set_num_threads(threads)
        with torch.no_grad():
            one()
            r, n, el, _legs = best_of_legs(one, budget)
            return r, n, el

    prev = torch.get_num_threads()
    r1, n1, el1 = leg(1, seconds / 2)
    rm, nm, elm = leg(ncores, seconds / 2) if ncores > 1 else (r1, n1, el1)
    torch.set_num_threads(prev)
    v1, vm = r1 * per_call, rm * per_call
    return {"value": max(v1, vm), "unit": "pooled-lookups/s", "threads": ncores if vm > v1 else 1,
            "one_thread": v1, "all_threads": {"threads": ncores, "value": vm},
            "torch_version": torch.__version__, "omp_wait_policy": os.environ.get("OMP_WAIT_POLICY", ""),
            "sample": f"best of 3 legs each: {n1} + {nm} batches, F.embedding_bag(mode='sum') looped over {len(ws)} tables, "
                      f"{el1:.1f} s (1 thread) + {elm:.1f} s ({ncores} threads)"}


def verify_last_batch(torch, eng, plan, d_idx, L, n_sample=32, seed=7):
    """What the timed loop left in the output buffers of its LAST batch, checked after the timed region:
    (1) every bag of every table against an independent torch gather on the GPU, summed in index order
        in fp32 exactly as the kernel does -- bit for bit (L = 1: the pooled row IS the table row);
    (2) a sample of bags per table against the oracle (oracle/emb_oracle.c) on rows copied back from HBM.
    Returns the number of bags compared in (1); raises AssertionError on the first difference."""
    from oracle import oracle
    rng = np.random.default_rng(seed)
    dev = plan.outputs[0].device
    checked = 0
    for k, (out, idx) in enumerate(zip(plan.outputs, d_idx)):
        w = eng.table_tensor(k)
        B, dim = out.shape
        rows = w[idx.long()].float().view(B, L, dim)
        acc = rows[:, 0, :] + 0.0
        for j in range(1, L):
            acc = acc + rows[:, j, :]
        if not torch.equal(out, acc):
            raise AssertionError(f"table {k}: timed output differs from the in-order torch gather-sum")
        checked += B
        del rows, acc
        sel = np.unique(rng.integers(0, B, size=min(n_sample, B)))
        pos = (sel[:, None] * L + np.arange(L)[None, :]).reshape(-1)
        idx_s = idx[torch.from_numpy(pos).to(dev)].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
        uniq, inv = np.unique(idx_s, return_inverse=True)
        small = w[torch.from_numpy(uniq).to(dev)].cpu().numpy()
        want = oracle.c_bag_sum(small, inv.astype(np.int64), np.arange(sel.shape[0], dtype=np.int64) * L)
        if not np.array_equal(out[torch.from_numpy(sel).to(dev)].cpu().numpy(), want):
            raise AssertionError(f"table {k}: timed output differs from the oracle on the sampled bags")
    return checked


def run_single(args):
    import torch
    import pim_embedding_lookup_amd as pel

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    spec = workload_spec(pel, args)
    rows_list, dim, B = spec["rows"], spec["dim"], spec["B"]
    T = len(rows_list)
    eng = pel.EmbeddingEngine(device=0, max_tables=T)
    want_cpu = not args.no_cpu_baseline
    n_host = T if args.workload in ("c1", "c2") else 2        # c3: 5 GB per table, sample two on the host
    host_tables = make_tables_on_gpu(torch, eng, rows_list, dim, dev, keep_host=n_host if want_cpu else 0,
                                     dtype=spec.get("dtype", "f32"))
    batches = make_batches(pel, spec, args.nbatch, order=args.index_order)
    hot_learnt = None
    if args.hot_rows > 0:          # the ENGINE picks them from the first batch's indices (emb_learn_hot_rows; round 5: host-side numpy)
        hot_learnt = [eng.learn_hot_rows(t, batches[0][0][t], args.hot_rows, min_share=0.05) for t in range(T)]

    plans, plan_idx = [], []
    for idx, off in batches:
        d_idx = [torch.from_numpy(i.view(np.int32)).to(dev) for i in idx]
        d_off = [torch.from_numpy(o.view(np.int32)).to(dev) for o in off]
        d_out = [torch.empty((B, dim), dtype=torch.float32, device=dev) for _ in range(T)]
        plans.append(eng.plan(list(range(T)), d_idx, d_off, d_out))
        plan_idx.append(d_idx)
    alg_bytes, n_bags, n_idx = plans[0].bytes()

    stream = torch.cuda.current_stream(dev)
    if args.graph:                       # graph capture needs a non-default stream; everything below runs on it
        stream = torch.cuda.Stream(dev)
        torch.cuda.set_stream(stream)
    sh = stream.cuda_stream
    extra = [torch.cuda.Stream(dev) for _ in range(max(args.streams, 1) - 1)]
    handles = [sh] + [x.cuda_stream for x in extra]

    def timed_region(replay=None):
        """W untimed warm-up steps, then EXACTLY K timed steps between two device-wide synchronizes.  Returns
        (sync-clock seconds, event-clock seconds, HIP-event ms over the K launches).  replay: a captured graph holding
        the K launches (one graph launch instead of K kernel enqueues)."""
        for i in range(args.warmup):
            plans[i % len(plans)].launch(handles[i % len(handles)])
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        if replay is not None:
            replay.replay()
        else:
            for i in range(args.steps):
                plans[i % len(plans)].launch(handles[i % len(handles)])
        for x in extra:
            stream.wait_stream(x)
        ev1.record(stream)
        while not ev1.query():               # the event clock: the K-th launch has retired (polled)
            pass
        t_event = time.perf_counter() - t0
        torch.cuda.synchronize()             # the contract's closing bracket, inside the primary clock
        t_sync = time.perf_counter() - t0
        return t_sync, t_event, ev0.elapsed_time(ev1)

    # The same W / K shape on the chip as a fresh process finds it (idle clocks), reported next to the headline number:
    # the driver's default run is W = 5 / K = 20, 0.5 ms in all.
    cold_sync, _cold_event, cold_ms = timed_region() if args.prewarm_ms > 0 else (None, None, None)
    # Device pre-warm (untimed, before the W warm-up steps): ~0.25 s of the same launches brings the chip to the state a
    # serving loop is in; the W warm-up steps and the EXACTLY K timed steps follow unchanged.
    t_pre = time.perf_counter()
    n_pre = 0
    while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3:
        for _ in range(64):
            plans[n_pre % len(plans)].launch(sh)
            n_pre += 1
        torch.cuda.synchronize()
    graph = None
    if args.graph:                       # the K launches of the timed region captured once, replayed as ONE graph launch
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.cuda.graph(graph, stream=stream):
            for i in range(args.steps):
                plans[i % len(plans)].launch(torch.cuda.current_stream(dev).cuda_stream)
        graph.replay()
        torch.cuda.synchronize()
    wall, wall_event, dev_ms = timed_region(graph)
    kernel_us = dev_ms * 1000.0 / args.steps          # avg launch duration on the launch stream

    # what was just timed, checked outside the timed region: every bag of EVERY rotating batch the timed loop wrote (all
    # NBATCH plans' output buffers, not only the last one's) bit for bit against an in-order torch gather-sum, plus the
    # oracle on a sample of bags per table
    touched = sorted({(args.steps - 1 - j) % len(plans) for j in range(min(args.steps, len(plans)))}) if args.steps > 0 else [0]
    n_checked = 0
    try:
        for q in touched:
            n_checked += verify_last_batch(torch, eng, plans[q], plan_idx[q], spec["L"])
    except AssertionError as ex:
        print(f"bench.py: VERIFICATION FAILED: {ex}", file=sys.stderr, flush=True)
        raise SystemExit(1)
    kinds = eng.stats()["n_launches_by_kind"]
    dominant_kernel = KERNEL_OF_KIND[max(range(len(kinds)), key=lambda k: kinds[k])]
    try:                        # the kernel's own code bytes + the launch's signature: what a counter profile is tied to
        launch_id = launch_identity(plans[0])
    except Exception as ex:  # noqa: BLE001 -- an unreadable code object drops the profile (with the reason), never the run
        launch_id = None
        print(f"bench.py: no launch identity ({type(ex).__name__}: {ex}): counter profiles are not used", file=sys.stderr)
    elem_b = 2 if spec.get("dtype") == "f16" else 4
    result = {
        "metric": "pooled-lookups/sec + achieved HBM GB/s, 26-table dim-16 Kaggle, 1/2/4/8 GPU",
        "value": args.steps * n_bags / wall,
        "unit": "pooled-lookups/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1000.0 / args.steps,
        # both closing brackets, as on the N > 1 lines: `sync` (primary, the contract's: the clock stops after the
        # device-wide synchronize) and `event` (the event behind the K-th launch has fired)
        "clock": "sync", "ms_per_step_sync": wall * 1000.0 / args.steps, "ms_per_step_event": wall_event * 1000.0 / args.steps,
        "value_sync": args.steps * n_bags / wall, "value_event": args.steps * n_bags / wall_event,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": spec.get("dtype", "f32"), "data": "synthetic",
        "verified": True,
        "verify": {"bags_bit_exact_vs_torch_gather": n_checked, "oracle_sample_bags_per_table": 32, "batches_checked": len(touched),
                   "what": "outputs of every rotating batch the timed loop wrote (%d of %d plans), all %d tables, after the timed region"
                           % (len(touched), len(plans), T)},
        # the same W warm-up + K timed steps measured BEFORE the device pre-warm (a fresh process, idle clocks): `value` is the
        # warmed figure, this the cold one
        "value_cold": None if cold_sync is None else args.steps * n_bags / cold_sync,
        "ms_per_step_cold": None if cold_sync is None else cold_sync * 1000.0 / args.steps,
        "config": {"workload": "%s, %d rotating batches" % (spec["name"], len(plans)),
                   "tables": T, "dim": dim, "bags_per_table": B, "pooling": spec["L"],
                   "table_bytes": eng.stats()["table_bytes"], "hot_rows_hint": args.hot_rows,
                   "hot_rows_staged_per_table": None if hot_learnt is None else [n for n, _ in hot_learnt],
                   "hot_rows_mean_share": None if hot_learnt is None else float(np.mean([sh_ for _, sh_ in hot_learnt])),
                   "prewarm_ms": args.prewarm_ms, "prewarm_launches": n_pre,
                   # the same W warm-up + K timed steps BEFORE the pre-warm (a fresh process, idle clocks): us per step on
                   # the sync clock / per launch by HIP events
                   "timed_without_prewarm_us": None if cold_sync is None else cold_sync * 1e6 / args.steps,
                   "kernel_without_prewarm_us": None if cold_ms is None else cold_ms * 1000.0 / args.steps,
                   "launch_mode": "one hipGraph holding the K launches" if args.graph else "K eager kernel enqueues",
                   "launches_by_kind": kinds, "dominant_kernel": dominant_kernel,
                   "parallelism": "single" if len(handles) == 1 else "single GPU, %d streams" % len(handles)},
        "roofline": roofline_object(alg_bytes, kernel_us, measured_traffic(profile_key(args, spec), dominant_kernel, launch_id),
                                    unique_row_bytes(batches[0], dim * elem_b)
                                    if spec["L"] > 1 or spec["dist"] != "uniform" else None,
                                    meta_bytes=4 * (n_idx + n_bags),
                                    uniq_line_bytes=unique_line_bytes(batches[0], dim * elem_b)
                                    if spec["L"] == 1 and spec["dist"] != "uniform" else None,
                                    alg_read_bytes=n_idx * (dim * elem_b + 4) + 4 * n_bags),
    }
    if launch_id:               # (scalars: a record that keeps only scalar members of `roofline` keeps them)
        result["roofline"].update(kernel_symbol=launch_id["kernel_symbol"], kernel_sha256=launch_id["kernel_sha256"],
                                  launch_signature=launch_id["launch_signature"], device_code_sha256=launch_id["device_code_sha256"])
    if want_cpu:
        result["cpu_baseline"] = cpu_baseline(pel, host_tables, batches[0], args.cpu_seconds)
    # key order as on the N > 1 lines: objects first, scalars after them, the contract's keys last (a record that keeps only the
    # tail of the text still holds them)
    from importlib import import_module
    result = import_module("pim-embedding-lookup_amd.dist_bench").driver_proof(result)
    print(json.dumps(result))
    for p in plans:
        p.destroy()
    eng.close()


def run_coalesced(args):
    """--coalesce R: a step = R independent requests (each one lookup() call's worth: all T tables, B bags) added to the
    request queue and ONE flush -- one fused launch -- against the same R requests issued back to back as R prepared-plan
    launches (the fastest one-by-one form there is).  Inputs and outputs resident in HBM; every request of every rotating
    slot is checked against the in-order torch gather-sum and the oracle after the timed region."""
    import torch
    import pim_embedding_lookup_amd as pel

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    spec = workload_spec(pel, args)
    rows_list, dim, B, R = spec["rows"], spec["dim"], spec["B"], args.coalesce
    T = len(rows_list)
    eng = pel.EmbeddingEngine(device=0, max_tables=T)
    make_tables_on_gpu(torch, eng, rows_list, dim, dev, dtype=spec.get("dtype", "f32"))
    nslots = max(2, min(args.nbatch, 4))
    batches = make_batches(pel, spec, nslots * R)
    q = pel.RequestQueue(eng, pel.EMB_IDX_U32, pel.EMB_MEM_DEVICE)
    slots = []
    for j in range(nslots):
        reqs = []
        for r in range(R):
            idx, off = batches[j * R + r]
            d_idx = [torch.from_numpy(i.view(np.int32)).to(dev) for i in idx]
            d_off = [torch.from_numpy(o.view(np.int32)).to(dev) for o in off]
            d_out = [torch.empty((B, dim), dtype=torch.float32, device=dev) for _ in range(T)]
            arr, n, _res, keep = q.descriptors(list(range(T)), d_idx, d_off, d_out)
            reqs.append(dict(arr=arr, n=n, keep=keep, idx=d_idx, plan=eng.plan(list(range(T)), d_idx, d_off, d_out)))
        slots.append(reqs)
    import ctypes as C
    many = []                         # per slot: the R requests' descriptors back to back + descriptors per request (emb_queue_add_many)
    for reqs in slots:
        arr = (pel.lib.EmbLookupDesc * sum(rq["n"] for rq in reqs))()
        at = 0
        for rq in reqs:
            C.memmove(C.byref(arr, at * C.sizeof(pel.lib.EmbLookupDesc)), rq["arr"], rq["n"] * C.sizeof(pel.lib.EmbLookupDesc))
            at += rq["n"]
        many.append((arr, (C.c_uint32 * len(reqs))(*[rq["n"] for rq in reqs])))
    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream
    one_add_per_request = os.environ.get("PIMEMB_QUEUE_ADD", "many") == "each"

    def step_queue(i):
        if one_add_per_request:          # R calls (R client threads would make them): ~1 us of Python + ctypes each
            for rq in slots[i % nslots]:
                q.add_descriptors(rq["arr"], rq["n"])
        else:                            # a front end that drained R requests hands them over in one call
            q.add_many(*many[i % nslots])
        q.flush(sh)

    def step_one_by_one(i):
        for rq in slots[i % nslots]:
            rq["plan"].launch(sh)

    def timed(step):
        for i in range(args.warmup):
            step(i)
        torch.cuda.synchronize()
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        ev0.record(stream)
        for i in range(args.steps):
            step(i)
        ev1.record(stream)
        torch.cuda.synchronize()
        return time.perf_counter() - t0, ev0.elapsed_time(ev1) * 1e-3

    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < args.prewarm_ms * 1e-3:
        for i in range(16):
            step_one_by_one(i)
        torch.cuda.synchronize()
    for rq in slots[0]:
        for o in rq["plan"].outputs:
            o.zero_()
    wall_1, dev_1 = timed(step_one_by_one)
    for reqs in slots:                      # the queue must write every output itself
        for rq in reqs:
            for o in rq["plan"].outputs:
                o.fill_(float("nan"))
    wall_q, dev_q = timed(step_queue)
    n_checked = 0
    try:
        for reqs in slots[:min(nslots, args.steps)]:
            for rq in reqs:
                n_checked += verify_last_batch(torch, eng, rq["plan"], rq["idx"], spec["L"], n_sample=8)
    except AssertionError as ex:
        print(f"bench.py: VERIFICATION FAILED: {ex}", file=sys.stderr, flush=True)
        raise SystemExit(1)
    alg_bytes, n_bags, _n_idx = slots[0][0]["plan"].bytes()
    lookups = args.steps * R * n_bags
    result = {
        "metric": "pooled-lookups/sec + achieved HBM GB/s, 26-table dim-16 Kaggle, 1/2/4/8 GPU",
        "value": lookups / wall_q, "unit": "pooled-lookups/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_q * 1000.0 / args.steps, "clock": "sync",
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": spec.get("dtype", "f32"), "data": "synthetic",
        "verified": True,
        "verify": {"bags_bit_exact_vs_torch_gather": n_checked, "what": "every request of every rotating slot the queued loop wrote "
                   "(outputs NaN-filled before it), after the timed region"},
        "coalesce": {"requests_per_flush": R, "us_per_request_queued": wall_q * 1e6 / (args.steps * R),
                     "us_per_request_one_by_one": wall_1 * 1e6 / (args.steps * R),
                     "value_one_by_one": lookups / wall_1, "speedup": wall_1 / wall_q,
                     "device_us_per_flush": dev_q * 1e6 / args.steps, "device_us_per_R_launches": dev_1 * 1e6 / args.steps,
                     "add": "each" if one_add_per_request else "many",
                     "what": "a step = R requests (each: all %d tables, %d bags) -- queued: emb_queue_add_many (or R x emb_queue_add: "
                             "PIMEMB_QUEUE_ADD=each) + ONE emb_queue_flush (one fused launch); one by one: R prepared-plan launches "
                             "back to back" % (T, B)},
        "config": {"workload": "%s; served as %d independent requests per step through the request queue, %d rotating slots"
                               % (spec["name"], R, nslots),
                   "tables": T, "dim": dim, "bags_per_table": B, "pooling": spec["L"], "requests_per_step": R,
                   "parallelism": "single"},
        "roofline": roofline_object(alg_bytes * R, dev_q * 1e6 / args.steps, None, None),
    }
    print(json.dumps(result))
    for reqs in slots:
        for rq in reqs:
            rq["plan"].destroy()
    q.close()
    eng.close()


def self_launch(n_ranks: int, argv, script: str | None = None) -> int:
    """`python bench.py --gpus N` with no launcher around it.  This process has made no GPU call and makes
    none: it starts one fresh child per rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, rendezvous on
    127.0.0.1 and a free port), relays rank 0's stdout -- the ONE JSON line -- and returns the worst child's
    exit status.  A child that fails takes the job down: the others get PIMEMB_LAUNCH_GRACE seconds (default
    30) to end by themselves, then SIGTERM / SIGKILL by their exact PIDs.  The reference's counterpart is one
    lookup() call that fans out to every device (upmem/include/emb_host.h:258-270, 297, 312-321)."""
    import socket
    import subprocess
    import threading

    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    base = dict(os.environ, WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), MASTER_ADDR="127.0.0.1",
                MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between processes needs it on this image
    cmd = [sys.executable, script or os.path.abspath(__file__)] + list(argv)   # (script: the CPU test's stand-in rank)
    procs = []
    for r in range(n_ranks):
        env = dict(base, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1))

    def relay(r, pipe):      # rank 0's JSON line is the job's stdout; everything else (library chatter) goes to stderr
        for line in pipe:
            if r == 0 and line.lstrip().startswith("{"):
                sys.stdout.write(line)
                sys.stdout.flush()
            else:
                sys.stderr.write("[rank %d] %s" % (r, line))
                sys.stderr.flush()

    threads = [threading.Thread(target=relay, args=(r, p.stdout), daemon=True) for r, p in enumerate(procs)]
    for t in threads:
        t.start()
    limit = float(os.environ.get("PIMEMB_LAUNCH_TIMEOUT", "3000"))
    grace = float(os.environ.get("PIMEMB_LAUNCH_GRACE", "30"))
    t0, first_fail = time.time(), None
    while any(p.poll() is None for p in procs):
        time.sleep(0.05)
        now = time.time()
        if first_fail is None and any(p.poll() not in (None, 0) for p in procs):
            first_fail = now
        if (first_fail is not None and now - first_fail > grace) or now - t0 > limit:
            why = "a rank failed" if first_fail is not None else "time limit of %.0f s" % limit
            print("bench.py: ending the remaining ranks (%s)" % why, file=sys.stderr, flush=True)
            for sig_wait in (5.0, 0.0):
                live = [p for p in procs if p.poll() is None]
                for p in live:
                    (p.terminate if sig_wait else p.kill)()
                t1 = time.time()
                while sig_wait and any(p.poll() is None for p in live) and time.time() - t1 < sig_wait:
                    time.sleep(0.05)
            if first_fail is None:
                first_fail = now
            break
    for p in procs:
        p.wait()
    for t in threads:
        t.join(timeout=5)
    codes = [p.returncode for p in procs]
    worst = max((c if c >= 0 else 128 - c) for c in codes)
    if worst == 0 and first_fail is not None:
        worst = 124
    if worst:
        print("bench.py: rank exit codes %s" % codes, file=sys.stderr, flush=True)
    return worst


def main():
    args = parse_args()
    launched = "WORLD_SIZE" in os.environ or "RANK" in os.environ
    if args.gpus > 1 and not launched:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1 or os.environ.get("PIMEMB_FORCE_DIST") == "1":   # last: 1-rank RCCL rehearsal
        from importlib import import_module
        import_module("pim-embedding-lookup_amd.dist_bench").run(args, HBM_PEAK_GBS)
    elif args.coalesce > 0:
        run_coalesced(args)
    else:
        run_single(args)


if __name__ == "__main__":
    main()
