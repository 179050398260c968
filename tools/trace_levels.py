"""Per tree level of the LAST run in a rocprofv3 kernel trace of tools/trace_run.py (main queue = the busiest): span, time inside kernels, the
largest kernels.  A level starts at its k_tr_find.  Under the profiler the host falls behind at the level boundaries: the idle time at a
level's end is the profiler's, the kernel time is the device's.   python tools/trace_levels.py <kernel_trace.csv[.gz]> [runs = 4]"""
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
main = max(byq.values(), key=len)
finds = [i for i, r in enumerate(main) if "k_tr_find" in r[2]]
run = main[finds[len(finds) - len(finds) // runs]:]  # the last run: it starts at its first level's k_tr_find


def short(nm):
    nm = re.sub(r"^void ", "", nm)
    nm = re.sub(r"\(.*", "", nm)
    return nm.replace("lsfm::", "").replace("rocprim::ROCPRIM_400200_NS::detail::", "rocprim ")[:34]


idx = [i for i, r in enumerate(run) if "k_tr_find" in r[2]] + [len(run)]
print(f"{len(idx) - 1} levels (the last k_tr_find is the root's return to its first frame), main queue, last of {runs} runs")
for lv in range(len(idx) - 1):
    seg = run[idx[lv]:idx[lv + 1]]
    span = (run[idx[lv + 1]][0] if idx[lv + 1] < len(run) else seg[-1][1]) - seg[0][0]
    busy = sum(e - s for s, e, _, _ in seg)
    gaps = [b[0] - a[1] for a, b in zip(seg, seg[1:])]
    tail = (run[idx[lv + 1]][0] - seg[-1][1]) if idx[lv + 1] < len(run) else 0
    per = defaultdict(float)
    for s, e, nm, _ in seg:
        per[short(nm)] += (e - s) / 1e3
    top = ", ".join(f"{k} {v:.0f}" for k, v in sorted(per.items(), key=lambda kv: -kv[1])[:6])
    print(f"level {lv:2d}: {len(seg):4d} launches, in kernels {busy / 1e6:6.3f} ms, idle inside the level {sum(g for g in gaps if g > 0) / 1e6:5.3f} ms, "
          f"idle before the next level {tail / 1e6:5.3f} ms | us: {top}")
