#!/usr/bin/env python3
"""Copies the rocprofv3 / bench outputs of one gpurun_out/<dir> into profiles/ and writes the HBM traffic summary bench.py reads
(FETCH_SIZE doubled as MI355X_MICROARCH prescribes for gfx950, + WRITE_SIZE; KB -> bytes; separate --pmc passes).
usage: python tools/refresh_profiles.py gpurun_out/<dir> <round tag, e.g. r02> <config, e.g. nc3500> <trees in the profiled run>
PROFILES_DIR=<dir> writes there instead of profiles/ (tools/measure.sh runs this ON THE BOX so that only the summaries travel back);
KEEP_RAW=0 leaves the per-dispatch counter CSVs (megabytes) where they are."""
import collections
import csv
import glob
import json
import os
import shutil
import sqlite3
import sys

src, tag, config, trees = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
OUT = os.environ.get("PROFILES_DIR", "profiles")
KEEP_RAW = os.environ.get("KEEP_RAW", "1") != "0"
os.makedirs(OUT, exist_ok=True)


def short(n):
    n = n.replace("void ", "").split("(")[0]
    base = n.split("::")[-1].split("<")[0]
    if base == "k_schur_panel" and "<48" in n:
        return "k_schur_panel_48"  # the pass over the tiles the 32-slot variant flagged (most work-groups leave at once)
    return base


def counter(name):
    agg = collections.defaultdict(lambda: [0, 0.0])
    files = glob.glob(f"{src}/pmc_{name}/**/*counter_collection.csv", recursive=True)
    if files:
        if KEEP_RAW:
            shutil.copy(files[0], f"{OUT}/{tag}_pmc_{name}_counter_collection_{config}.csv")
        for r in csv.DictReader(open(files[0])):
            agg[short(r["Kernel_Name"])][0] += 1
            agg[short(r["Kernel_Name"])][1] += float(r["Counter_Value"])
        return agg
    dbs = glob.glob(f"{src}/pmc_{name}/**/*.db", recursive=True)
    con = sqlite3.connect(dbs[0])
    cur = con.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    view = "counters_collection" if "counters_collection" in tabs else [t for t in tabs if "counter" in t.lower()][0]
    cols = [d[0] for d in cur.execute(f"select * from {view} limit 1").description]
    kn = [c for c in cols if "kernel" in c.lower() and "name" in c.lower()][0]
    val = [c for c in cols if c.lower() in ("value", "counter_value")][0]
    rows = list(cur.execute(f"select {kn}, {val} from {view}"))
    with open(f"{OUT}/{tag}_pmc_{name}_per_kernel_{config}.csv", "w") as f:
        f.write("Kernel_Name,Counter_Value\n")
        for k, v in rows:
            f.write(f"\"{k}\",{v}\n")
            agg[short(k)][0] += 1
            agg[short(k)][1] += float(v)
    return agg


out = {c: counter(c) for c in ("FETCH_SIZE", "WRITE_SIZE")}
names = sorted(out["FETCH_SIZE"], key=lambda n: -(2 * out["FETCH_SIZE"][n][1] + out["WRITE_SIZE"][n][1]))
per_launch, detail, total = {}, {}, 0.0
for n in names:
    cnt, fk, wk = out["FETCH_SIZE"][n][0], out["FETCH_SIZE"][n][1], out["WRITE_SIZE"][n][1]
    hbm = (2 * fk + wk) * 1024
    total += hbm
    per_launch[n] = hbm / max(cnt, 1)
    detail[n] = dict(launches=cnt, fetch_KB_raw_total=fk, write_KB_total=wk, hbm_bytes_per_launch_corrected=hbm / max(cnt, 1))
# bench.py's K9 "launch" is one Schur assembly of a level: the panel variants (16 / 32 / 48 slots, by tile) and k_schur_w for
# the tiles none of them takes are launched together, once per level -- count the step, not the kernels
if "k_schur_w" in detail and "k_schur_panel" in detail:
    steps = detail["k_schur_w"]["launches"]
    k9 = sum(detail[n]["hbm_bytes_per_launch_corrected"] * detail[n]["launches"] for n in ("k_schur_panel", "k_schur_panel_48", "k_schur_w") if n in detail)
    per_launch["k_schur_panel"] = k9 / max(steps, 1)
    detail["k_schur_panel"]["hbm_bytes_per_level_all_variants"] = per_launch["k_schur_panel"]
    detail["k_schur_panel"]["levels"] = steps
for n in names[:14]:
    print(f"{n[:44]:44s} n={detail[n]['launches']:5d} -> {per_launch[n] / 1e6:8.1f} MB/launch")
res = dict(config=config, trees_in_profiled_run=trees, bytes_per_tree=total / trees, per_launch=per_launch, kernels=detail,
           source=f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py --config {config}; FETCH_SIZE doubled (gfx950), KB -> bytes")
json.dump(res, open(f"{OUT}/{tag}_pmc_traffic_summary_{config}.json", "w"), indent=1)
print("HBM bytes per tree: %.2f GB" % (total / trees / 1e9))
st = glob.glob(f"{src}/stats/**/*kernel_stats.csv", recursive=True)
if st:
    shutil.copy(st[0], f"{OUT}/{tag}_bench_{config}_kernel_stats.csv")
for name, dst in (("bench_default.log", f"{OUT}/{tag}_bench_default.json"), ("bench_prof.log", f"{OUT}/{tag}_bench_under_rocprof.json")):
    if os.path.exists(f"{src}/{name}"):
        lines = [l for l in open(f"{src}/{name}") if l.startswith("{")]
        if lines:
            open(dst, "w").write(lines[0])
for f in glob.glob(f"{src}/full_parity_*.json"):
    shutil.copy(f, f"{OUT}/{tag}_{os.path.basename(f)}")
