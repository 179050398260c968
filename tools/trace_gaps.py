"""Reads a rocprofv3 kernel trace (…_kernel_trace.csv[.gz]) of tools/trace_run.py: the LAST run on the busiest queue (the library's main
stream; the runs are alike, so the queue's dispatches split evenly) -- busy time, a histogram of the idle time between consecutive
dispatches, and the (kernel before -> kernel behind) pairs that own the idle time.  Under the profiler the HOST is slower: gaps of
hundreds of microseconds at the level boundaries are the profiler's; the 5-50 us ones are the device's.
    python tools/trace_gaps.py <kernel_trace.csv> [runs in the trace = 4]"""
import csv
import gzip
import re
import sys
from collections import defaultdict

path = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
op = gzip.open if path.endswith(".gz") else open
rows = []
with op(path, "rt") as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print(f"queue {q}: {len(lst)} dispatches, busy {1e-6 * sum(e - s for s, e, _, _ in lst):.2f} ms over the {runs} runs")
main = max(byq.values(), key=len)
n = len(main) // runs
run = main[(runs - 1) * n:]


def short(nm):
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"\(.*", "", nm)
    return nm.replace("lsfm::", "")[:52]


print(f"\nlast run, main queue: {len(run)} dispatches, {1e-6 * (run[-1][1] - run[0][0]):.2f} ms first start -> last end, busy {1e-6 * sum(e - s for s, e, _, _ in run):.2f} ms")
gaps = [(b[0] - a[1], short(a[2]), short(b[2])) for a, b in zip(run, run[1:])]
for lo, hi in [(-1e18, 0), (0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1e18)]:
    sel = [g[0] for g in gaps if lo < g[0] / 1e3 <= hi]
    print(f"  gaps in ({lo if lo > -1e17 else '-inf'}, {hi if hi < 1e17 else 'inf'}] us: {len(sel):5d}, together {1e-6 * sum(sel):7.3f} ms")
agg = defaultdict(lambda: [0, 0.0])
for d, a, b in gaps:
    if d > 2000:
        agg[(a, b)][0] += 1
        agg[(a, b)][1] += d / 1e3
print("\nidle time by the pair of kernels around it (gaps > 2 us):")
for (a, b), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:45]:
    print(f"  {t:8.1f} us in {c:4d} gaps   {a} -> {b}")
per = defaultdict(lambda: [0, 0])
for s, e, nm, _ in run:
    per[short(nm)][0] += 1
    per[short(nm)][1] += e - s
print("\nkernels of the run on the main queue:")
for nm, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:40]:
    print(f"  {nm:52s} {c:5d} x  {t / 1e6:7.3f} ms")
