#!/usr/bin/env python3
"""Where a tile of K9 (k_schur_panel) spends its time, phase by phase and per panel variant, over one NC3500-like tree.
Needs the library built with the profiling aid:  make -C linearsfm_amd/csrc clean; make -C linearsfm_amd/csrc K9_TIMING=1
(lane 0 of every work-group adds up shader clocks per phase; a few per cent slower than the product build).
usage: python tools/k9_phase_times.py [config]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api, synth  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
    _, maps = synth.make_config(cfg, seed=0)
    mono = synth.CONFIGS[cfg][0] == "Monocular"
    ctx = api.Context(0)
    L = api.lib()
    if not hasattr(L, "lsfm_debug_k9"):
        raise SystemExit("library built without K9_TIMING=1")
    t = ctx.tree_upload([dict(m if isinstance(m, dict) else m.__dict__) for m in maps], mono)
    ctx.tree_run(t)
    ctx.tree_run(t)
    out = (C.c_ulonglong * 64)()
    L.lsfm_debug_k9(out, 1)
    ctx.tree_run(t)
    L.lsfm_debug_k9(out, 0)
    v = np.array(list(out), dtype=np.float64).reshape(4, 16)
    names = ["header + poses -> slots", "barrier (pass consumed)", "zero panel, y", "stage P = W L", "barrier (staged)", "MFMA", "tile end (rhs, scatter)", "wait for the prefetch"]
    for k, smax in enumerate((8, 16, 32, 48)):
        tiles = v[k, 8]
        if not tiles:
            continue
        tot = v[k, :8].sum()
        print(f"variant <= {smax} slots: {int(tiles)} tiles, {v[k, 9] / tiles:.1f} poses and {v[k, 10] / tiles:.1f} MFMA tiles per wave on average, "
              f"{tot / tiles:.0f} clocks per tile")
        for n, x in zip(names, v[k, :8]):
            print(f"    {n:26s} {100 * x / tot:5.1f} %   {x / tiles:9.0f} clocks per tile")
        print(f"    {'prefetch issue':26s} {100 * v[k, 12] / (tot + v[k, 11] + v[k, 12]):5.1f} %   {v[k, 12] / tiles:9.0f} clocks per tile (on top of the 100 % above)")
    ctx.tree_free(t)
    ctx.close()


if __name__ == "__main__":
    main()
