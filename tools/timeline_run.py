"""Host marks of one analysing run of a resident tree (LSFM_TIMELINE=1: where the enqueuing thread is when, microseconds between marks)
and the per-level solver report (LSFM_DEBUG=1).  usage: LSFM_TIMELINE=1 [LSFM_DEBUG=1] python tools/timeline_run.py <config> [maps]"""
import sys
sys.path.insert(0, ".")
from linearsfm_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "synth16k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else None
typ, maps = synth.make_config(cfg, n) if n else synth.make_config(cfg)
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], typ == "Monocular")
del maps
ctx.tree_set_plans(t, False)
ctx.tree_run(t)
print("---- second analysing run ----", file=sys.stderr, flush=True)
st, rc = ctx.tree_run(t)
print({k: round(v, 2) if isinstance(v, float) else v for k, v in st.items() if k.startswith("t_") or k in ("attempts", "pcg_iterations")})
ctx.tree_free(t)
