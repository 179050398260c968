import sys
sys.path.insert(0, ".")
from linearsfm_amd import api, synth
typ, maps = synth.make_config("synth16k")
dicts = [m.__dict__ for m in maps]
ctx = api.Context(0)
for rep in range(int(sys.argv[1])):
    t = ctx.tree_upload(dicts, True)
    ctx.tree_set_plans(t, False)
    out = []
    for i in range(6):
        st, rc = ctx.tree_run(t)
        out.append((rc, st["attempts"], float("%.1e" % st["max_rel_residual"]), st["pcg_iterations"]))
    print(rep, out, flush=True)
    ctx.tree_free(t)
