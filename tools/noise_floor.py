#!/usr/bin/env python3
"""How far two evaluations of the reference path differ on a generated set: the oracle (fp64, the reference's arithmetic) against
its long-double twin (every solve carried in extended precision).  That distance is the floor below which a parity tolerance
cannot decide anything; the test docstrings quote it.  CPU only.
usage: python tools/noise_floor.py mono|stereo N new_per_frame vis "dict(lap=40, home=20, revisit=0.5, turn=0.15)"
Measured (8-core container): Mono spiral npf 64: 512 maps 4.1e-8, 1024 maps 1.8e-6, 2048 maps 4.5e-5; aerial 238 maps 4.7e-8 (40-frame
strips x 12: 9.4e-6); Stereo flower npf 64: 4096 maps 1.8e-7."""
import sys, time, numpy as np, json
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from linearsfm_amd import synth
from oracle import pyoracle as po
typ, N, npf, vis, path = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), eval(sys.argv[5])
mono = typ == "mono"
maps = (synth.make_mono_set if mono else synth.make_stereo_set)(N, npf, vis, 0, **path)
dicts = [po.localmap_to_dict(m) for m in maps]
t = time.time(); out, tm, rc = po.divide_conquer(dicts, mono, match_hash=True, threads=4); t1 = time.time() - t
t = time.time(); ext, _, rc2 = po.divide_conquer(dicts, mono, match_hash=True, extended=True); t2 = time.time() - t
mask = out["stno"] <= 0
def perr(a, b): return float(np.max(np.abs(a[mask] - b[mask]) / np.maximum(1, np.abs(b[mask]))))
print(json.dumps(dict(typ=typ, N=N, path=str(path), rc=[rc, rc2], oracle4_s=t1, twin_s=t2, oracle_vs_twin=perr(out["stVal"], ext["stVal"]),
                      m=int(out["m"]), n=int(out["n"]), nW=int(out["nW"]), nU=int(out["nU"]))))
