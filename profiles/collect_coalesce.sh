#!/bin/bash
# The request queue against one-request-per-launch: bench.py --coalesce R (device pointers: R requests -> ONE launch vs R
# prepared-plan launches) for the reference's serving shapes, and emb_queue_bench (host pointers, client threads against one
# front end).  usage: bash profiles/collect_coalesce.sh r04
round=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_${round}/coalesce
mkdir -p "$out"
for R in 1 2 4 8 16 32; do
  echo "coalesce R=$R"
  python3 "$root/bench.py" --workload c1 --coalesce $R --steps 2000 --warmup 200 --no-cpu-baseline > "$out/c1_R$R.json" 2> "$out/err.txt" || { echo "FAILED c1 $R"; tail -5 "$out/err.txt"; }
  python3 "$root/bench.py" --workload c2 --batch 32 --coalesce $R --steps 2000 --warmup 200 --no-cpu-baseline > "$out/c2b32_R$R.json" 2> "$out/err.txt" || echo "FAILED b32 $R"
  python3 "$root/bench.py" --workload c2 --batch 512 --coalesce $R --steps 1000 --warmup 100 --no-cpu-baseline > "$out/c2b512_R$R.json" 2> "$out/err.txt" || echo "FAILED b512 $R"
done
rm -f "$out/err.txt"
bash "$root/tools/queue_bench_sweep.sh" > "$out/emb_queue_bench_host_pointers.log" 2>&1
python3 - "$out" > "$out/SUMMARY.md" <<'PY'
import json, glob, sys, os
out = sys.argv[1]
print("# Request queue: R pending small requests -> ONE launch\n")
print("Collected by `profiles/collect_coalesce.sh` on the one-GPU box.  A request = one `lookup()` call's worth (all 26 Kaggle tables, B bags per")
print("table, one index per bag).\n")
print("## Device pointers (`bench.py --workload W [--batch B] --coalesce R`)\n")
print("queued = `emb_queue_add_many` + ONE `emb_queue_flush` per step; one by one = the same R requests as R prepared-plan launches back to")
print("back (the fastest one-request-per-launch form).  Every request of every rotating slot verified bit for bit after the timed loop.\n")
print("| shape | R | queued us / request | one by one us / request | speed-up | pooled lookups / s (queued) |\n|---|---|---|---|---|---|")
names = {"c1": "mini-batch 1 (reference README.md:6)", "c2b32": "mini-batch 32 (run.sh:119)", "c2b512": "512 bags (MAX_NR_BATCHES, run.sh:40-45)"}
for f in sorted(glob.glob(os.path.join(out, "*_R*.json")), key=lambda x: (os.path.basename(x).split("_R")[0], int(x.split("_R")[1][:-5]))):
    try:
        d = json.load(open(f))
    except ValueError:
        continue
    c = d["coalesce"]
    k = os.path.basename(f).split("_R")[0]
    print("| %s | %d | %.2f | %.2f | %.2fx | %.3e |" % (names[k], c["requests_per_flush"], c["us_per_request_queued"], c["us_per_request_one_by_one"], c["speedup"], d["value"]))
print("\n## Host pointers -- the reference's own calling convention (`csrc/tools/emb_queue_bench.cpp`, `tools/queue_bench_sweep.sh`)\n")
print("Client threads post requests with HOST buffers and wait for their rows; one front-end thread flushes whatever is pending.  one by one =")
print("`emb_lookup_batched(EMB_MEM_HOST)` per request from the same threads (serialised inside the engine).  Every row checked.\n```")
print(open(os.path.join(out, "emb_queue_bench_host_pointers.log")).read().strip())
print("```")
PY
cat "$out/SUMMARY.md" | head -30
