#!/usr/bin/env python3
"""How fast the chip moves W blocks (144 bytes each) with the access pattern of the W kernels, against the plain stream copy
(lsfm_wstream_bench): the ceiling that one-lane-per-block kernels (k_tr_entries, k_join_rhs_w, k_backsub, k_schur_w) work under.
usage: python tools/wstream_bench.py [blocks=4000000]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
ctx = api.Context(0)
for mode, name in ((0, "one lane per block, 18 doubles"), (1, "one lane per block, nine 16-byte pieces"), (2, "consecutive lanes on consecutive 16 bytes")):
    ms = ctx.wstream_bench(n, mode, reps=10)
    print(json.dumps(dict(pattern=name, blocks=n, MB_read_plus_written=2 * 144 * n / 1e6, avg_launch_ms=ms, TBps=2 * 144 * n / (ms * 1e-3) / 1e12)))
ctx.close()
