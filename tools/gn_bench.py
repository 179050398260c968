"""lsfm_gn_polish on a named stand-in set, from the device's own tree result: the objective / gradient trace and the wall clock of the
call's parts (LSFM_GN_TIMING=1: structure + upload, per assembly, per solve).  usage: python tools/gn_bench.py <config> [steps] [maps]
-> one JSON object per call on stdout (profiles/r06_gn_polish_<config>.json)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] != "--child":
    # the library's timing line goes to stderr of the process: run the work in a child and pick it up
    env = dict(os.environ, LSFM_GN_TIMING="1")
    p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"] + sys.argv[1:], env=env, capture_output=True, text=True)
    line = [l for l in p.stdout.splitlines() if l.startswith("{")]
    tim = [l for l in p.stderr.splitlines() if l.startswith("lsfm_gn:")]
    if not line:
        sys.exit(p.stdout + p.stderr)
    d = json.loads(line[0])
    d["library_timing"] = tim
    print(json.dumps(d, indent=1))
    sys.exit(0)

import numpy as np  # noqa: E402
from linearsfm_amd import api, synth  # noqa: E402

cfg = sys.argv[2]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nmaps = int(sys.argv[4]) if len(sys.argv) > 4 else 0
typ, maps = synth.make_config(cfg, nmaps or None)
mono = typ == "Monocular"
d = [m.__dict__ for m in maps]
ctx = api.Context(0)
G, stats, rc = ctx.divide_conquer(d, mono)
calls = []
for rep in range(2):
    t0 = time.perf_counter()
    st, obj, gn, hv, rc2 = ctx.gn_polish(d, mono, G, steps)
    calls.append(1e3 * (time.perf_counter() - t0))
print(json.dumps({"config": cfg, "type": typ, "maps": len(maps), "poses": int(G["m"]), "features": int(G["n"]), "steps": steps, "tree_ms": stats["t_total_ms"], "tree_rc": rc,
                  "gn_rc": rc2, "objective": obj.tolist(), "gradient_max": gn.tolist(), "halvings": hv.tolist(), "call_wall_ms": calls,
                  "max_state_change": float(np.max(np.abs(st - G["stVal"]))),
                  "note": "lsfm_gn_polish from the device's own tree result; call_wall_ms includes building the lsfm_map views in Python, the upload of the "
                          "local maps and the host's structure pass; library_timing = the library's own clocks (second call warm)"}))
