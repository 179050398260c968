"""Where the one-launch dense path (lsfm_small.hip) spends its time: the NC3500-like tree with phases of k_small_solve switched OFF one
at a time (LSFM_SMALL_DEBUG, read at every launch; the results of such runs are garbage, their t_small_ms is what is looked at).
usage: python tools/small_probe.py [config] [maps]"""
import os
import sys

sys.path.insert(0, ".")
from linearsfm_amd import api, synth

cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
typ, maps = synth.make_config(cfg, int(sys.argv[2])) if len(sys.argv) > 2 else synth.make_config(cfg)
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], typ == "Monocular")
ctx.tree_set_plans(t, False)
ctx.tree_set_stop_level(t, 4)  # the four levels of small systems only (garbage must not reach the sparse levels)
for name, bits in (("everything", 0), ("no V^-1 / L / y arithmetic", 1), ("no staging", 2), ("no E part", 4), ("no MFMA", 8), ("no Cholesky / solves", 16),
                   ("no back-substitution", 32), ("no panel zeroing", 64), ("passes only (16+32)", 48), ("no passes' work (1+2+4+8+64)", 79), ("nothing (127)", 127)):
    os.environ["LSFM_SMALL_DEBUG"] = str(bits)
    best = None
    for _ in range(4):
        try:
            st, rc = ctx.tree_run(t)
        except api.LsfmError:
            st = None
            continue
        if best is None or st["t_small_ms"] < best["t_small_ms"]:
            best = st
    print("%-34s t_small_ms %8.3f  (levels %d)  tree %.2f ms" % (name, best["t_small_ms"] if best else -1, best["small_levels"] if best else -1, best["t_total_ms"] if best else -1), flush=True)
