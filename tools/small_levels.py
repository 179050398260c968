"""Per level: the one-launch dense path (lsfm_small.hip) against the sparse level pipeline.  The tree is stopped after 1, 2, ... levels
(lsfm_tree_set_stop_level); the difference of two stops is a level's wall time -- with the dense path and without it.
usage: python tools/small_levels.py [config] [levels] [plans|analysing] [max poses of the dense path: 5]"""
import sys

sys.path.insert(0, ".")
import numpy as np

from linearsfm_amd import api, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
nlev = int(sys.argv[2]) if len(sys.argv) > 2 else 5
plans = len(sys.argv) > 3 and sys.argv[3] == "plans"
typ, maps = synth.make_config(cfg)
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], typ == "Monocular")
ctx.tree_set_plans(t, plans)
res = {}
for small in (True, False):
    ctx.set_small_solve(int(sys.argv[4]) if small and len(sys.argv) > 4 else (5 if small else 0))
    for k in range(1, nlev + 1):
        ctx.tree_set_stop_level(t, k)
        ts, tsm = [], []
        for _ in range(6):
            st, rc = ctx.tree_run(t)
            ts.append(st["t_total_ms"]); tsm.append(st["t_small_ms"])
        res[(small, k)] = (float(np.median(ts[1:])), float(np.median(tsm[1:])), st["small_levels"])
print(cfg, "plans" if plans else "analysing runs")
for k in range(1, nlev + 1):
    a, b = res[(True, k)], res[(False, k)]
    pa = res[(True, k - 1)] if k > 1 else (0, 0, 0)
    pb = res[(False, k - 1)] if k > 1 else (0, 0, 0)
    print("level %d: dense path %6.3f ms (its kernel %6.3f, levels on it %d) | pipeline %6.3f ms | the level alone: %6.3f vs %6.3f"
          % (k - 1, a[0], a[1] - pa[1], a[2], b[0], a[0] - pa[0], b[0] - pb[0]))
