#!/usr/bin/env python3
"""Full-size parity check (run on the GPU box): oracle vs HIP path on a whole synthetic Stereo set, plus two facts
that put the number in context on long open chains, where the camera system of the top joins reaches condition
numbers ~1e12 and ANY two fp64 solves of the same system differ far above 1e-6 in the global (drift) directions:
  * oracle vs oracle with a different (equally valid) elimination order   -> the reference path's own noise floor
  * parity of the RELATIVE poses between consecutive frames               -> the well-determined local geometry
usage: python tools/full_parity.py [maps=3499] [new_per_frame=130] [vis=5] [selfcheck=1] [mono=0]"""
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linearsfm_amd import api, synth  # noqa: E402
from linearsfm_amd.synth import rot_ypr, ypr_from_rot  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def rel_poses(st, stno):
    """pose k+1 expressed in the frame of pose k, for consecutive pose ids present in the state."""
    idx = np.nonzero(stno <= 0)[0][::6]
    ids = -stno[idx]
    order = np.argsort(ids)
    out = []
    for a, b in zip(order[:-1], order[1:]):
        if ids[b] != ids[a] + 1:
            continue
        pa, pb = st[idx[a]:idx[a] + 6], st[idx[b]:idx[b] + 6]
        Ra, Rb = rot_ypr(*pa[3:]), rot_ypr(*pb[3:])
        t = Ra @ (pb[:3] - pa[:3])
        out.append(np.concatenate([t, ypr_from_rot(Rb @ Ra.T)]))
    return np.array(out)


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 3499
    npf = int(sys.argv[2]) if len(sys.argv) > 2 else 130
    vis = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    selfcheck = int(sys.argv[4]) if len(sys.argv) > 4 else 1
    mono = bool(int(sys.argv[5])) if len(sys.argv) > 5 else False
    gen = synth.make_mono_set if mono else synth.make_stereo_set
    if os.environ.get("ORC_CHILD"):
        maps = gen(N, new_per_frame=npf, vis=vis, seed=0)
        out, _, _ = po.divide_conquer([po.localmap_to_dict(m) for m in maps], mono, match_hash=True)
        np.save(os.environ["ORC_CHILD"], out["stVal"])
        return
    po.build()
    maps = gen(N, new_per_frame=npf, vis=vis, seed=0)
    dicts = [po.localmap_to_dict(m) for m in maps]
    ctx = api.Context(0)
    got, stats, rc = ctx.divide_conquer(dicts, mono)
    t0 = time.time()
    exp, timing, orc = po.divide_conquer(dicts, mono, match_hash=True)
    mask = exp["stno"] <= 0

    def perr(a, b, m):
        return float(np.max(np.abs(a[m] - b[m]) / np.maximum(1.0, np.abs(b[m]))))
    rg, re = rel_poses(got["stVal"], exp["stno"]), rel_poses(exp["stVal"], exp["stno"])
    res = dict(maps=N, new_per_frame=npf, vis=vis, mono=mono, m=int(exp["m"]), n=int(exp["n"]), nU=int(exp["nU"]), nW=int(exp["nW"]),
               same_labels=bool(np.array_equal(got["stno"], exp["stno"])),
               same_structure=bool(np.array_equal(got["photo"], exp["photo"]) and np.array_equal(got["feature"], exp["feature"])
                                   and np.array_equal(got["Ui"], exp["Ui"]) and np.array_equal(got["Uj"], exp["Uj"])),
               pose_max_rel_err=perr(got["stVal"], exp["stVal"], mask), feature_max_rel_err=perr(got["stVal"], exp["stVal"], ~mask),
               relative_pose_max_abs_err=float(np.max(np.abs(rg - re))),
               info_rel_err={k: float(np.max(np.abs(got[k] - exp[k])) / np.max(np.abs(exp[k]))) for k in ("U", "W", "V")},
               gpu_ms=stats["t_total_ms"], gpu_rc=rc, gpu_stats=stats, oracle_s=timing, oracle_rc=orc, host_cores=os.cpu_count())
    if selfcheck:
        # the oracle against itself with the degree ordering instead of the nested-dissection one
        tmp = "/tmp/orc_selfcheck.npy"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), str(N), str(npf), str(vis), str(selfcheck), str(int(mono))],
                              env=dict(os.environ, ORC_CHILD=tmp, ORC_ORDER="1"))
        alt = np.load(tmp)
        res["oracle_vs_oracle_reordered_pose_max_rel_err"] = perr(alt, exp["stVal"], mask)
        res["oracle_vs_oracle_reordered_relative_pose_max_abs_err"] = float(np.max(np.abs(rel_poses(alt, exp["stno"]) - re)))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
