#!/usr/bin/env python3
"""The matrix cores' busy share of the SIMD cycles per kernel from a counter pass
    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --output-format csv -d <dir> -o run -- python3 bench.py --config <c> --steps 2 --warmup 1 --cpu-baseline 0 --extras 0
(counters alone: never combined with a trace domain) -> profiles/<round>_pmc_mfma_busy_<config>.json
usage: python tools/mfma_busy.py <dir>/run_counter_collection.csv <round> <config>"""
import collections
import csv
import json
import os
import re
import sys

path, rnd, cfg = sys.argv[1], sys.argv[2], sys.argv[3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for r in csv.DictReader(open(path)):
    m = re.search(r"lsfm::(k_\w+(?:<[^>]*>)?)", r["Kernel_Name"])
    if not m or not any(x in m.group(1) for x in ("k_schur_panel", "k_small_solve", "k_sn_panel", "k_sn_syrk")):
        continue
    acc[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_BUSY_CU_CYCLES":
        n[m.group(1)] += 1
out = {"command": f"rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES -- python3 bench.py --config {cfg} --steps 2 --warmup 1 --cpu-baseline 0 --extras 0 (counters alone, no trace domain)",
       "note": "SQ_VALU_MFMA_BUSY_CYCLES sums the four SIMDs of a CU, SQ_BUSY_CU_CYCLES counts the CU: their ratio / 4 = the share of SIMD cycles the matrix cores were busy while the kernel held the CU",
       "kernels": {}}
for k, v in sorted(acc.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", 0)):
    cu, mf = v.get("SQ_BUSY_CU_CYCLES", 0), v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0)
    out["kernels"][k] = {"launches": n[k], "SQ_VALU_MFMA_BUSY_CYCLES": mf, "SQ_BUSY_CU_CYCLES": cu, "mfma_busy_share_of_simd_cycles": round(mf / cu / 4, 4) if cu else None}
dst = os.path.join(os.environ.get("PROFILES_DIR", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")), f"{rnd}_pmc_mfma_busy_{cfg}.json")
os.makedirs(os.path.dirname(dst), exist_ok=True)
json.dump(out, open(dst, "w"), indent=1)
for k, v in out["kernels"].items():
    print(k, v["mfma_busy_share_of_simd_cycles"], v["launches"])
