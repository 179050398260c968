"""The launch sequence of ONE tree level on every queue, from a rocprofv3 kernel trace of tools/trace_run.py: start (us from the level's start),
duration, queue, kernel.   python tools/trace_level_seq.py <kernel_trace.csv> <level> [runs = 4]"""
import csv
import re
import sys
from collections import defaultdict

path, level = sys.argv[1], int(sys.argv[2])
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"], int(r.get("Workgroup_Size_X", 0) or 0), int(r.get("Grid_Size_X", 0) or 0)))
rows.sort()
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
main = max(byq.values(), key=len)
finds = [i for i, r in enumerate(main) if "k_tr_find" in r[2]]
run = main[finds[len(finds) - len(finds) // runs]:]  # the last run: it starts at its first level's k_tr_find
idx = [i for i, r in enumerate(run) if "k_tr_find" in r[2]] + [len(run)]
t0 = run[idx[level]][0]
t1 = run[idx[level + 1]][0] if idx[level + 1] < len(run) else run[-1][1]


def short(nm):
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"\(.*", "", nm)
    return nm.replace("lsfm::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim ")[:60]


qs = {q: i for i, q in enumerate(sorted(byq, key=lambda q: -len(byq[q])))}
print(f"level {level}: {(t1 - t0) / 1e3:.1f} us")
for s, e, nm, q, wg, grid in rows:
    if t0 <= s < t1:
        print(f"{(s - t0) / 1e3:9.1f} +{(e - s) / 1e3:8.1f} us  q{qs[q]}  grid {grid:9d}  {short(nm)}")
