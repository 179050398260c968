#!/usr/bin/env python3
"""Poses per K9 tile, level by level, of one tree (LSFM_K9_HIST=1: launch_schur_panel prints the histogram of every level on stderr).
usage: LSFM_K9_HIST=1 python tools/k9_hist.py [config]"""
import os
import sys

os.environ["LSFM_K9_HIST"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api, synth  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
_, maps = synth.make_config(cfg, seed=0)
mono = synth.CONFIGS[cfg][0] == "Monocular"
ctx = api.Context(0)
t = ctx.tree_upload([dict(m if isinstance(m, dict) else m.__dict__) for m in maps], mono)
ctx.tree_run(t)
ctx.tree_free(t)
ctx.close()
