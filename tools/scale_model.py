"""A PREDICTION of `bench.py --gpus G` (one tree, strong scaling, blocks below + feature-sharded levels above) from a ONE-GPU kernel trace:
what the first real SCALE run is to be held against (no run on more than one GPU exists).

    rocprofv3 --kernel-trace --output-format csv -d <dir> -o run -- python3 tools/trace_run.py <config> <runs>
    python tools/scale_model.py <dir>/..._kernel_trace.csv <runs> [maps = the configuration's]

Per tree level of the LAST run (main queue, a level starts at its k_tr_find) the kernel time is split into
    feature side  everything that walks features or W blocks (transform, join, K7 / K9 / K11, pattern inserts): 1 / G on a rank that
                  holds 1 / G of the features (top levels) or 1 / G of the level's joins (block levels)
    pose side     factorisation, triangular solves, refinement, SpMV (k_sn_*, k_chol_*, k_pcg_*, k_spmv*, k_perm*, k_rz*): at the block
                  levels 1 / G like the rest; at the top levels (1 - sigma) / G + sigma with sigma = the replicated share of the block
                  products the library reports (lsfm_stats.dist_work_shared / dist_work_total: 1.2 % synth-16k, 12.2 % nc3500 at 8 ranks)
    launches      a level cannot take less than its launch count x 4.7 us (the cadence of dependent kernels in one queue, measured: a
                  tiny kernel takes 4.5-5 us start to start) -- the floor of a level whose work has been divided away
Communication (the prompt's figure: 7 xGMI links x ~153 GB/s per GPU, point to point):
    all-to-all of the slices   every rank sends (G - 1) / G of its block's root map, over min(G - 1, 7) links at once
    per top level              one all-reduce of S and E (288 B per block of S; ring: 2 (G - 1) / G x bytes / 153 GB/s) + the pose rows of the
                               transform + the pose solution (48 M B each way) + the separators' accumulators (small); 15 us of latency each
"""
import csv
import re
import sys
from collections import defaultdict

path = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
rows.sort()
byq = defaultdict(list)
for r in rows:
    byq[r[3]].append(r)
main = max(byq.values(), key=len)
finds = [i for i, r in enumerate(main) if "k_tr_find" in r[2]]
per_run = len(finds) // runs                      # levels (+ the final re-anchoring) of one run: every one starts at its k_tr_find
run = main[finds[len(finds) - per_run]:]          # the LAST run (the first one of a process also carries the code objects' first use)
idx = [i for i, r in enumerate(run) if "k_tr_find" in r[2]] + [len(run)]
POSE = re.compile(r"k_sn_|k_chol_|k_pcg_|k_spmv|k_perm|k_rz|k_x_init|k_copy\b")
levels = []
for lv in range(len(idx) - 1):
    seg = run[idx[lv]:idx[lv + 1]]
    tp = sum(e - s for s, e, nm, _ in seg if POSE.search(nm)) / 1e6
    tf = sum(e - s for s, e, nm, _ in seg if not POSE.search(nm)) / 1e6
    levels.append((tf, tp, len(seg)))
L = len(levels) - 1  # the last "level" is the root's return to its first frame
T1 = sum(a + b for a, b, _ in levels)
print(f"{L} join levels + the final re-anchoring; kernel time on the main queue of one GPU: {T1:.1f} ms (feature side {sum(a for a, _, _ in levels):.1f}, pose side {sum(b for _, b, _ in levels):.1f})")
CAD = 4.7e-3   # ms per launch
LINK = 153.0   # GB/s per xGMI link
LAT = 0.015    # ms per collective


def predict(G, sigma, root_bytes, s_bytes_top, m_poses):
    g = G.bit_length() - 1
    t_block = t_top = t_comm = 0.0
    for lv, (tf, tp, nl) in enumerate(levels):
        if lv < L - g:
            t_block += max((tf + tp) / G, nl * CAD)
        else:
            t_top += max(tf / G + tp * ((1 - sigma) / G + sigma), nl * CAD)
            if lv < L and G > 1:
                k = lv - (L - g)  # 0 .. g - 1: the S of the top level is the largest; the ones below have half the poses each step down
                sb = s_bytes_top / (2 ** (g - 1 - k))
                t_comm += 2 * (G - 1) / G * (sb + 3 * 48 * m_poses) / (LINK * 1e6) + 4 * LAT
    t_a2a = 0.0 if G == 1 else ((G - 1) / G * root_bytes / G) / (min(G - 1, 7) * LINK * 1e6) + LAT
    return t_block, t_a2a, t_top, t_comm


cfg = sys.argv[3] if len(sys.argv) > 3 else "synth16k"
# bytes of the whole level at the blocks' root (W dominates: 152 B per block) and of S at the top join, from the run's own sizes
PAR = {"synth16k": dict(root_bytes=3.0e9, s_top=0.2e9, m=16386, sigma8=0.012), "nc3500": dict(root_bytes=1.0e9, s_top=19e6, m=3499, sigma8=0.122)}[cfg]
print(f"{cfg}: sigma(8 ranks) = {PAR['sigma8']}, W of the whole level at the hand-over ~{PAR['root_bytes'] / 1e9:.1f} GB, S of the top join ~{PAR['s_top'] / 1e6:.0f} MB")
print(f"{'G':>2} {'blocks':>9} {'all-to-all':>11} {'top levels':>11} {'all-reduces':>12} {'total ms':>9} {'speed-up':>9} {'efficiency':>10}")
base = None
for G in (1, 2, 4, 8):
    sigma = PAR["sigma8"] * (G - 1) / 7.0 if G > 1 else 0.0
    tb, ta, tt, tc = predict(G, sigma, PAR["root_bytes"], PAR["s_top"], PAR["m"])
    tot = tb + ta + tt + tc
    base = base or tot
    print(f"{G:>2} {tb:9.1f} {ta:11.2f} {tt:11.1f} {tc:12.2f} {tot:9.1f} {base / tot:9.2f} {base / tot / G:10.2f}")
