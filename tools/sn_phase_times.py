#!/usr/bin/env python3
"""Where k_sn_panel (the supernode panel kernel of the factorisation chain) spends its clocks, phase by phase, over one tree.
Needs the library built with the profiling aid:  make -C linearsfm_amd/csrc K9_TIMING=1 (touch lsfm_pcg.hip first).
usage: python tools/sn_phase_times.py [config]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
_, maps = synth.make_config(cfg, seed=0)
mono = synth.CONFIGS[cfg][0] == "Monocular"
ctx = api.Context(0)
L = api.lib()
if not hasattr(L, "lsfm_debug_sn"):
    raise SystemExit("library built without K9_TIMING=1")
t = ctx.tree_upload([dict(m if isinstance(m, dict) else m.__dict__) for m in maps], mono)
ctx.tree_run(t)
ctx.tree_run(t)
out = (C.c_ulonglong * 32)()
L.lsfm_debug_sn(out, 1)
ctx.tree_run(t)
L.lsfm_debug_sn(out, 0)
v = np.array(list(out), dtype=np.float64).reshape(2, 16)
names = ["index set-up", "blocks -> LDS", "after the column loop (flush)", "rhs + inverse diagonal + stores", "rank update (fused)"]
for k, tag in enumerate(("k_sn_panel<false>", "k_sn_panel<true>")):
    n = v[k, 5]
    if not n:
        continue
    tot = v[k, :5].sum() + v[k, 7:10].sum()
    print(f"{tag}: {int(n)} groups (first chunk's work-group each), {v[k, 6] / n:.1f} block columns on average, {tot / n:.0f} clocks = {tot / n / 2400:.1f} us per group")
    for nm, x in zip(names, v[k, :5]):
        print(f"    {nm:34s} {100 * x / tot:5.1f} %   {x / n:9.0f} clocks")
    cols = v[k, 6]
    if cols:
        print(f"    first diagonal block factored by the pivot wave (nothing to overlap): {v[k, 7] / n:.0f} clocks per group")
        pn = ["slot reads", "finish arithmetic", "writes", "wait at the barrier after B", "dot products", "wait at the barrier after A"]
        print("    the first panel lane (tid 128), per block column: " + ", ".join(f"{a} {v[k, 10 + q] / cols:.0f}" for q, a in enumerate(pn)))
        print(f"    inside the column loop, per block column ({cols / n:.1f} columns per group): B (finish the column against the published L_tt) "
              f"{v[k, 9] / cols:.0f} clocks, A (dot products of the next column | pivot wave: next L_tt) {v[k, 8] / cols:.0f} clocks")
ctx.tree_free(t)
ctx.close()
