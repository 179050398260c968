import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 33
maps = synth.make_stereo_set(N, 6, 5, seed=4, lap=12 if N > 40 else 0)
ctx = api.Context(0)
t = ctx.tree_upload(maps, False)
ctx.tree_set_plans(t, False)
for i in range(3):
    st, rc = ctx.tree_run(t)
    print("run", i, rc, st["t_total_ms"], st["max_rel_residual"], st["pcg_iterations"], flush=True)
out = ctx.tree_download(t)
print("ok", out["m"], out["n"], flush=True)
