"""A few analysing runs of a resident tree and nothing else, for `rocprofv3 --kernel-trace -- python tools/trace_run.py <config> [runs]`
(tools/trace_gaps.py reads the trace: per queue busy time, gaps, the library's copy / fill launches)."""
import sys
sys.path.insert(0, ".")
from linearsfm_amd import api, synth
cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
typ, maps = synth.make_config(cfg)
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], typ == "Monocular")
del maps
ctx.tree_set_plans(t, False)
for _ in range(runs):
    st, rc = ctx.tree_run(t)
print({k: round(v, 2) if isinstance(v, float) else v for k, v in st.items() if k.startswith("t_")})
ctx.tree_free(t)
