#!/usr/bin/env python3
"""One-off robustness sweep (GPU box): random small/medium trees, Stereo and Mono, HIP path vs the oracle.
usage: python tools/random_parity_sweep.py [cases=24] [seed=0]"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linearsfm_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    po.build()
    ctx = api.Context(0)
    worst = 0.0
    for c in range(cases):
        mono = bool(rng.integers(0, 2))
        N = int(rng.integers(1, 70 if mono else 140))
        npf = int(rng.integers(3, 30))
        vis = int(rng.integers(3 if not mono else 4, 14))
        seed = int(rng.integers(0, 10000))
        maps = (synth.make_mono_set if mono else synth.make_stereo_set)(N, new_per_frame=npf, vis=vis, seed=seed)
        dicts = [po.localmap_to_dict(m) for m in maps]
        exp, _, orc = po.divide_conquer(dicts, mono)
        try:
            got, stats, rc = ctx.divide_conquer(dicts, mono)
        except api.LsfmError as e:
            print(json.dumps(dict(case=c, mono=mono, N=N, npf=npf, vis=vis, seed=seed, oracle_rc=orc, error=str(e)[:120])))
            continue
        same = bool(np.array_equal(got["stno"], exp["stno"]) and np.array_equal(got["photo"], exp["photo"]) and
                    np.array_equal(got["Ui"], exp["Ui"]) and np.array_equal(got["Uj"], exp["Uj"]))
        err = float(np.max(np.abs(got["stVal"] - exp["stVal"]) / np.maximum(1.0, np.abs(exp["stVal"]))))
        ierr = max(float(np.max(np.abs(np.asarray(got[k]) - np.asarray(exp[k]))) / np.max(np.abs(np.asarray(exp[k])))) for k in ("U", "W", "V"))
        worst = max(worst, err)
        print(json.dumps(dict(case=c, mono=mono, N=N, npf=npf, vis=vis, seed=seed, oracle_rc=orc, rc=rc, same_structure=same,
                              state_max_rel_err=err, info_max_rel_err=ierr)))
    print(json.dumps(dict(worst_state_err=worst)))


if __name__ == "__main__":
    main()
