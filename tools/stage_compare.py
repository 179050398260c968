"""Stage times (HIP events) of analysing runs and of repeat runs of one resident tree, side by side: where the two differ.
usage: python tools/stage_compare.py <config> [runs = 10]"""
import sys
sys.path.insert(0, ".")
import numpy as np
from linearsfm_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
typ, maps = synth.make_config(cfg)
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], typ == "Monocular")
del maps
keys = ("t_total_ms", "t_transform_ms", "t_join_ms", "t_schur_ms", "t_pcg_ms", "t_backsub_ms", "t_small_ms")
for mode, plans in (("analysing", False), ("repeat", True)):
    ctx.tree_set_plans(t, plans)
    for _ in range(3):
        ctx.tree_run(t)
    acc = {k: [] for k in keys}
    for _ in range(runs):
        st, rc = ctx.tree_run(t)
        for k in keys:
            acc[k].append(st[k])
    print(mode, " ".join(f"{k[2:-3]} {np.median(v):.2f}" for k, v in acc.items()))
ctx.tree_free(t)
