"""How stable is the refinement at the top of a synth-16k Mono tree?  N analysing runs of the resident tree; prints the runs that
ended with a system not converged and how many runs needed a second attempt."""
import sys, json
sys.path.insert(0, ".")
import numpy as np
from linearsfm_amd import api, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
typ, maps = synth.make_config(sys.argv[2] if len(sys.argv) > 2 else "synth16k")
mono = typ == "Monocular"
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], mono)
del maps
plans = len(sys.argv) > 3 and sys.argv[3] == "plans"
ctx.tree_set_plans(t, plans)
res = []
times = []
for i in range(n):
    st, rc = ctx.tree_run(t)
    times.append(st["t_total_ms"])
    res.append((rc, st["not_converged"], float("%.2e" % st["max_rel_residual"]), st["pcg_iterations"], st["attempts"]))
bad = [r for r in res if r[0] != 0 or r[2] > 1e-8]
import numpy as _np
print("plans" if plans else "analysing", "mean ms %.1f" % _np.mean(times), "runs", n, "failed", len(bad), bad[:6], "repeated", sum(1 for r in res if r[4] > 1), "worst ok", max(r[2] for r in res if r not in bad))
ctx.tree_free(t)
