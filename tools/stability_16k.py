"""How stable is a resident tree from run to run?  N runs of one tree; prints the runs that ended with a system not converged, how
many needed a second attempt, and -- with LSFM_FACTOR_DIGEST=1 in the environment -- how many distinct camera systems (s_digest),
factors (factor_digest) and final states the runs produced, and whether equal systems always gave equal factors (they must: the
factorisation accumulates in fixed point).
usage: python tools/stability_16k.py <runs> <config> [plans] [maps]"""
import hashlib
import os
import sys

sys.path.insert(0, ".")
import numpy as np

from linearsfm_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = sys.argv[2] if len(sys.argv) > 2 else "synth16k"
nmaps = int(sys.argv[4]) if len(sys.argv) > 4 else None
typ, maps = synth.make_config(cfg, nmaps) if nmaps else synth.make_config(cfg)
mono = typ == "Monocular"
ctx = api.Context(0)
t = ctx.tree_upload([m.__dict__ for m in maps], mono)
del maps
plans = len(sys.argv) > 3 and sys.argv[3] == "plans"
ctx.tree_set_plans(t, plans)
res, times, dig = [], [], []
want_digest = bool(os.environ.get("LSFM_FACTOR_DIGEST"))
errors = []
for i in range(n):
    try:
        st, rc = ctx.tree_run(t)
    except api.LsfmError as e:       # a run that ended with an error (e.g. a pivot far below zero): counted, the tree stays usable
        errors.append((i, str(e)[-90:]))
        continue
    times.append(st["t_total_ms"])
    res.append((rc, st["not_converged"], float("%.2e" % st["max_rel_residual"]), st["pcg_iterations"], st["attempts"]))
    if want_digest:
        _, _, _, stv = ctx.tree_download_state(t)
        dig.append((st["s_digest"], st["factor_digest"], hashlib.sha1(np.ascontiguousarray(stv).tobytes()).hexdigest()[:12], st["refactor_mismatch"], st["s_rebuild_mismatch"]))
bad = [r for r in res if r[0] != 0 or r[2] > 1e-8]
print(cfg, "plans" if plans else "analysing", "mean ms %.1f" % np.mean(times), "runs", n, "failed", len(bad), bad[:6], "repeated",
      sum(1 for r in res if r[4] > 1), "worst ok", max([r[2] for r in res if r not in bad] or [0]), "steps/run", sorted(set(r[3] for r in res)))
if errors:
    print("   runs that ended with an ERROR: %d of %d: %s" % (len(errors), n, errors[:4]))
if want_digest:
    by_s = {}
    for s, f, h, _, _ in dig:
        by_s.setdefault(s, set()).add(f)
    print("   distinct camera-system digests %d, factor digests %d, final states %d; runs whose systems were equal but whose factors differed: %d"
          % (len(by_s), len(set(d[1] for d in dig)), len(set(d[2] for d in dig)), sum(1 for v in by_s.values() if len(v) > 1)))
    print("   every camera system of every run factored twice: systems whose two factors were not the same bits: %d (of %d runs x levels)"
          % (sum(d[3] for d in dig), n))
    print("   the camera systems of every level assembled twice (S and E: U scatter, K9, fallback kernel): levels whose two assemblies were not the same bits: %d"
          % sum(d[4] for d in dig))
ctx.tree_free(t)
