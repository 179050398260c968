import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api, synth
typ, maps = synth.make_config("synth16k", int(sys.argv[1]) if len(sys.argv) > 1 else 16384)
ctx = api.Context(0)
t = ctx.tree_upload(maps, True)
ctx.tree_set_plans(t, False)
for i in range(int(sys.argv[2]) if len(sys.argv) > 2 else 3):
    st, rc = ctx.tree_run(t)
    print("run", i, "rc", rc, "ms", round(st["t_total_ms"], 1), "maxres", st["max_rel_residual"], "its", st["pcg_iterations"], "notconv", st["not_converged"], flush=True)
