"""LSFM_POISON=1: arenas filled with 0xFF before every run -- a kernel that reads what the run has not written sees NaN / -1."""
import sys
sys.path.insert(0, ".")
import numpy as np
from linearsfm_amd import api, synth
cfg, n = sys.argv[1], int(sys.argv[2])
typ, maps = synth.make_config(cfg, n) if n else synth.make_config(cfg)
mono = typ == "Monocular"
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], mono)
for plans in (False, True):
    ctx.tree_set_plans(t, plans)
    for i in range(3):
        try:
            st, rc = ctx.tree_run(t)
            print(cfg, "plans" if plans else "analysing", i, "rc", rc, "nc", st["not_converged"], "res %.2e" % st["max_rel_residual"], "attempts", st["attempts"], "its", st["pcg_iterations"])
        except Exception as e:
            print(cfg, "plans" if plans else "analysing", i, "EXC", str(e)[:200])
ctx.tree_free(t)
