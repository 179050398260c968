"""How far two fp64-faithful evaluations of the REFERENCE's algorithm differ on a named set: the oracle (fp64 throughout) against its
long-double twin (every solve of the tree in long double, transform and assembly fp64 as pinned to the reference) -- under both error
definitions bench.py and tests/test_gpu_parity.py use for the device.  CPU only.  -> profiles/r06_oracle_twin_floor_<config>.json (r05_* : the same, made in round 5)
usage: python tools/oracle_twin_floor.py <config> [...]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
from common import feat_param_err, pose_param_err, pose_param_true_rel_err  # noqa: E402
from linearsfm_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402

po.build()
for cfg in sys.argv[1:] or ["nc3500"]:
    typ, maps = synth.make_config(cfg)
    mono = typ == "Monocular"
    dicts = [po.localmap_to_dict(m) for m in maps]
    t0 = time.time()
    a, _, rc = po.divide_conquer(dicts, mono)
    t1 = time.time()
    b, _, rcx = po.divide_conquer(dicts, mono, extended=True)
    t2 = time.time()
    assert rc == 0 and rcx == 0
    res = dict(config=cfg, maps=len(maps), oracle_s=t1 - t0, twin_s=t2 - t1,
               pose_param_max_rel_err_oracle_vs_twin=pose_param_err(a["stVal"], b["stVal"], b["stno"]),
               pose_param_max_true_rel_err_oracle_vs_twin=pose_param_true_rel_err(a["stVal"], b["stVal"], b["stno"]),
               feature_param_max_rel_err_oracle_vs_twin=feat_param_err(a["stVal"], b["stVal"], b["stno"]),
               note="oracle/lsfm_oracle.c (fp64) against the same tree with every solve in long double (orc_set_extended): the spread of the "
                    "reference's own arithmetic on this set, in the two metrics the device is held to")
    json.dump(res, open(os.path.join(ROOT, "profiles", f"r06_oracle_twin_floor_{cfg}.json"), "w"), indent=1)
    print(res, flush=True)
