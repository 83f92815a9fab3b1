"""The reference's three experiments (upmem/r.sh: NR_COLS=64, MAX_INDICES_PER_BATCH=120, 32 tables,
64 bags -- table size :21-39, number of tables :41-66, batch size :68-88) through populate_mram /
lookup of libpimemb.so.  The reference rebuilds per point (-D macros); here shapes are arguments of
emb_host_bench.  Prints one markdown table per sweep: median lookup() wall time, host pointers, PCIe
inclusive, results validated on the CPU.   python reference_sweeps.py [--quick]"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
BENCH = os.path.join(HERE, "..", "..", "lib", "emb_host_bench")
COLS, L = 64, 120


def run(tables, rows, bags, keep):
    out = subprocess.run([BENCH, str(tables), str(COLS), str(rows), str(bags), str(L), "100", str(keep)],
                         capture_output=True, text=True, timeout=1500)
    txt = out.stdout + out.stderr
    m = re.search(r"median ([0-9.]+) ms", txt)
    ok = "Validation result: true" in txt
    pop = re.search(r"populate: ([0-9.]+) ms", txt)
    if out.returncode != 0 or not m:
        raise SystemExit("emb_host_bench failed:\n" + txt[-2000:])
    return float(m.group(1)), ok, float(pop.group(1))


def table(title, header, points):
    print("\n" + title + "\n")
    print("| " + header + " | lookup() median ms | pooled outputs/s | row gathers/s | validated | populate s |")
    print("|---|---|---|---|---|---|")
    for label, (tables, rows, bags, keep) in points:
        ms, ok, pop = run(tables, rows, bags, keep)
        print("| %s | %.3f | %.2e | %.2e | %s | %.1f |" % (label, ms, tables * bags / (ms * 1e-3),
                                                       tables * bags * L / (ms * 1e-3), "yes" if ok else "NO", pop / 1e3),
              flush=True)


quick = "--quick" in sys.argv
sizes = [125_000, 500_000, 2_000_000] if quick else [125_000, 250_000, 500_000, 1_000_000, 2_000_000, 4_000_000,
                                                    8_000_000, 13_900_000]
table("Table size (r.sh:21-39): 32 tables x 64 columns, 64 bags x 120 indices", "rows per table",
      [("%d" % n, (32, n, 64, 32 if n <= 1_000_000 else 2)) for n in sizes])
table("Number of tables (r.sh:41-66): 500 000 rows x 64 columns, 64 bags x 120 indices", "tables",
      [("%d" % t, (t, 500_000, 64, t)) for t in ([2, 32] if quick else [2, 4, 8, 16, 32])])
table("Batch size (r.sh:68-88): 32 tables x 500 000 rows x 64 columns, 120 indices per bag", "bags per table",
      [("%d" % b, (32, 500_000, b, 32)) for b in ([8, 100] if quick else [8, 16, 32, 64, 100])])
