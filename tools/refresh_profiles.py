#!/usr/bin/env python3
"""Copies the rocprofv3 / bench outputs of one gpurun_out/<dir> into profiles/ (round 1 names) and prints the HBM traffic
per launch (FETCH_SIZE doubled as MI355X_MICROARCH prescribes for gfx950, + WRITE_SIZE; KB -> bytes).
usage: python tools/refresh_profiles.py gpurun_out/r01e"""
import collections
import csv
import glob
import json
import shutil
import sys

src = sys.argv[1]
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{src}/pmc_{c}/*/*counter_collection.csv")[0]
    shutil.copy(f, f"profiles/r01_pmc_{c}_counter_collection.csv")
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[n][0] += 1
        agg[n][1] += float(r["Counter_Value"])
    out[c] = agg
names = sorted(out["FETCH_SIZE"], key=lambda n: -(2 * out["FETCH_SIZE"][n][1] + out["WRITE_SIZE"][n][1]))[:12]
res = {}
for n in names:
    cnt, fk, wk = out["FETCH_SIZE"][n][0], out["FETCH_SIZE"][n][1], out["WRITE_SIZE"][n][1]
    hbm = (2 * fk + wk) * 1024 / cnt
    res[n] = dict(launches=cnt, fetch_KB_raw_total=fk, write_KB_total=wk, hbm_bytes_per_launch_corrected=hbm)
    print(f"{n[:44]:44s} n={cnt:4d} -> {hbm / 1e6:8.1f} MB/launch")
json.dump(res, open("profiles/r01_pmc_traffic_summary.json", "w"), indent=1)
shutil.copy(glob.glob(f"{src}/stats/*/*kernel_stats.csv")[0], "profiles/r01_bench_nc3500_kernel_stats.csv")
open("profiles/r01_bench_default.json", "w").write([l for l in open(f"{src}/bench_default.log") if l.startswith("{")][0])
open("profiles/r01_bench_under_rocprof.json", "w").write([l for l in open(f"{src}/bench_prof.log") if l.startswith("{")][0])
d = json.load(open("profiles/r01_bench_default.json"))
print(d["value"], d["device_breakdown_ms"], d["roofline"]["achieved"], d["roofline"]["frac"], d["cpu_baseline"]["value"])
