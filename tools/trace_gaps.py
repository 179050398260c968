"""Reads a rocprofv3 kernel trace (…_kernel_trace.csv) of tools/trace_run.py: the LAST run's dispatches per queue -- busy time,
idle time between consecutive dispatches, the twenty largest gaps with the kernels on either side, and per kernel name the
launches / time on every queue.   python tools/trace_gaps.py <kernel_trace.csv> [runs in the trace]"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Stream_Id", "0"))))
rows.sort()
# the runs are separated by the host's synchronisation at the end of lsfm_tree_run: the largest gaps over all queues
ends = []
last_end = rows[0][1]
for i, (s, e, n, q) in enumerate(rows):
    if i and s - last_end > 0:
        ends.append((s - last_end, i))
    last_end = max(last_end, e)
cuts = sorted(i for _, i in sorted(ends, reverse=True)[:runs - 1])
first = cuts[-1] if cuts else 0
sel = rows[first:]
t0, t1 = sel[0][0], max(e for _, e, _, _ in sel)
print(f"last run: {len(sel)} dispatches, {1e-6 * (t1 - t0):.2f} ms from first start to last end")


def short(n):
    n = re.sub(r"^void ", "", n)
    n = re.sub(r"\(.*", "", n)
    return n.replace("lsfm::", "")[:60]


byq = defaultdict(list)
for r in sel:
    byq[r[3]].append(r)
for q, lst in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    busy = sum(e - s for s, e, _, _ in lst)
    gaps = []
    for a, b in zip(lst, lst[1:]):
        gaps.append((b[0] - a[1], short(a[2]), short(b[2])))
    pos = [g for g in gaps if g[0] > 0]
    print(f"\nqueue {q}: {len(lst)} dispatches, busy {1e-6 * busy:.2f} ms, idle between its own dispatches {1e-6 * sum(g[0] for g in pos):.2f} ms "
          f"({len(pos)} gaps, median {sorted(g[0] for g in pos)[len(pos) // 2] / 1e3 if pos else 0:.1f} us)")
    for g in sorted(pos, reverse=True)[:20]:
        print(f"    {g[0] / 1e3:8.1f} us   {g[1]}  ->  {g[2]}")
    per = defaultdict(lambda: [0, 0])
    for s, e, n, _ in lst:
        per[short(n)][0] += 1
        per[short(n)][1] += e - s
    for n, (c, t) in sorted(per.items(), key=lambda kv: -kv[1][1])[:25]:
        print(f"      {n:60s} {c:5d} x  {t / 1e6:7.3f} ms")
