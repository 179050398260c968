#!/usr/bin/env python3
"""Full-size parity record (run on the GPU box): the HIP path vs the oracle on a whole stand-in set, next to the numbers that
make the comparison decidable:
  * oracle vs oracle with a different (equally valid) elimination order      -> how far two fp64 evaluations of the
                                                                                 reference path differ on this input
  * oracle and HIP vs the oracle with every solve carried in long double     -> which fp64 answer is nearer the exact
    (oracle/lsfm_solve_num.inc compiled for a second type; transform and        solution of the systems the reference
    assembly stay fp64, they are pinned to the reference)                       assembles
usage: python tools/full_parity.py [config=nc3500] [maps=0 (the configuration's own)] [extended=1]"""
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
    config = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    extended = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    typ, maps = synth.make_config(config, N or None)
    mono = typ == "Monocular"
    if os.environ.get("ORC_CHILD"):
        out, _, _ = po.divide_conquer([po.localmap_to_dict(m) for m in maps], mono, match_hash=True, extended=bool(int(os.environ.get("ORC_EXT", "0"))))
        np.save(os.environ["ORC_CHILD"], out["stVal"])
        return
    po.build()
    dicts = [po.localmap_to_dict(m) for m in maps]
    ctx = api.Context(0)
    got, stats, rc = ctx.divide_conquer(dicts, mono)
    exp, timing, orc = po.divide_conquer(dicts, mono, match_hash=True)
    mask = exp["stno"] <= 0

    def perr(a, b, m=mask):
        return float(np.max(np.abs(a[m] - b[m]) / np.maximum(1.0, np.abs(b[m]))))

    def child(env):
        tmp = "/tmp/orc_child.npy"
        subprocess.check_call([sys.executable, os.path.abspath(__file__), config, str(N), "0"], env=dict(os.environ, ORC_CHILD=tmp, **env))
        return np.load(tmp)
    res = dict(config=config, maps=len(maps), mono=mono, m=int(exp["m"]), n=int(exp["n"]), nU=int(exp["nU"]), nW=int(exp["nW"]),
               same_labels=bool(np.array_equal(got["stno"], exp["stno"])),
               same_structure=bool(np.array_equal(got["photo"], exp["photo"]) and np.array_equal(got["feature"], exp["feature"])
                                   and np.array_equal(got["Ui"], exp["Ui"]) and np.array_equal(got["Uj"], exp["Uj"])),
               tolerance=1e-6,
               hip_vs_oracle_pose_max_rel_err=perr(got["stVal"], exp["stVal"]),
               hip_vs_oracle_feature_max_rel_err=perr(got["stVal"], exp["stVal"], ~mask),
               hip_vs_oracle_relative_pose_max_abs_err=float(np.max(np.abs(rel_poses(got["stVal"], exp["stno"]) - rel_poses(exp["stVal"], exp["stno"])))),
               info_rel_err={k: float(np.max(np.abs(got[k] - exp[k])) / np.max(np.abs(exp[k]))) for k in ("U", "W", "V")},
               gpu_ms=stats["t_total_ms"], gpu_rc=rc, oracle_s=timing, oracle_rc=orc, host_cores=os.cpu_count())
    alt = child({"ORC_ORDER": "1"})
    res["oracle_vs_oracle_reordered_pose_max_rel_err"] = perr(alt, exp["stVal"])
    if extended:
        t0 = time.time()
        ext = child({"ORC_EXT": "1"})
        res["extended_oracle_s"] = time.time() - t0
        res["oracle_vs_extended_pose_max_rel_err"] = perr(exp["stVal"], ext)
        res["oracle_reordered_vs_extended_pose_max_rel_err"] = perr(alt, ext)
        res["hip_vs_extended_pose_max_rel_err"] = perr(got["stVal"], ext)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
