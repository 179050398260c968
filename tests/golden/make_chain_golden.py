#!/usr/bin/env python3
"""A WHOLE mid-size tree evaluated without any arithmetic of the oracle or the library (tests/golden/chain_*.npz):

    every transform and every join assembly   by the REAL reference (oracle/_ref/ref_dump: lmj_Transform_PF3D*, lmj_LinearLS_PF3D* up
                                               to its call of lmj_solveLinearSFM*, LinearSFMImp.cpp compiled where it lies),
    every solve                                by tests/common.py schur_reference_solve: the exact solution of the FULL
                                               reference-assembled normal equations (long-double residuals of the full system, a
                                               dense LAPACK Cholesky factor of the Schur complement as the preconditioner),
    the tree                                   the reference's loop (lmj_PF3D_Divide_Conquer*, LinearSFMImp.cpp:1932-2063 / 6517-6630:
                                               pairing, the unpaired carry, re-anchoring of the odd outputs and of the final map),

on a few hundred synthetic local maps -- systems of up to 512 poses with lap closures, where the small fixtures stop at 90.  The
reference's own solve cannot run here (CHOLMOD is absent, no stand-in is written); a direct solve of an SPD system is unique, and this
is that unique solution to fp64.  Stored: the generator's arguments (the set is re-made from them), the final map's labels and state,
the poses after every level (a test that fails says where).  Used by tests/test_oracle_cpu.py (the ORACLE's tree against it) and
tests/test_gpu_parity.py (the device's).  Authoring container only (needs /root/reference); minutes of CPU time.

Usage:  python tests/golden/make_chain_golden.py [stereo|mono ...]
"""
import ctypes as C
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from linearsfm_amd import synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402  (only its map <-> text writer: formatting, no arithmetic)
from refdump import read_dump, sub  # noqa: E402
from common import schur_reference_solve  # noqa: E402

REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
HERE = os.path.dirname(os.path.abspath(__file__))
CHAINS = {
    # name: (type, generator keyword arguments) -- laps that return to their start: the upper joins close loops
    "chain_stereo_n512": ("Stereo", dict(n_maps=512, new_per_frame=16, vis=5, seed=31, lap=60, home=8, revisit=0.4)),
    "chain_mono_n200": ("Monocular", dict(n_maps=200, new_per_frame=10, vis=4, seed=32, **synth.SPIRAL)),
    # the same path at 768 maps: 10 levels, the top systems past the dense path and the leaf tasks of the device's factorisation
    "chain_mono_n768": ("Monocular", dict(n_maps=768, new_per_frame=10, vis=4, seed=34, **synth.SPIRAL)),
    # the NC3500-like path (laps of 120 frames through one place, synth.FLOWER) at 2 048 maps: 2 048 poses, 11 levels, the top
    # systems as wide as the device's 32-slot Schur panels and supernode groups see them on the headline set
    "chain_stereo_n2048": ("Stereo", dict(n_maps=2048, new_per_frame=24, vis=5, seed=33, **synth.FLOWER)),
}


def make_set(typ, kw):
    return synth.make_mono_set(**kw) if typ == "Monocular" else synth.make_stereo_set(**kw)


def write_map(path, d, mono):
    g = po.dict_to_map(d)
    po.lib().orc_write_map(path.encode(), int(mono), C.byref(g))
    po.lib().orc_map_free(C.byref(g))


def ref_join(typ, mono, A, B, tmp):
    """A transformed into B's frame and joined with it: everything up to the solver's arguments by the real reference, the solve by
    the exact dense-preconditioned refinement.  Returns the joint map dict."""
    fa, fb, fo = (os.path.join(tmp, x) for x in ("A.txt", "B.txt", "o.bin"))
    write_map(fa, A, mono)
    write_map(fb, B, mono)
    subprocess.check_call([REF_DUMP, "pair", typ, fa, fb, fo])
    D = read_dump(fo)
    sr = sub(D, "solve")
    J = dict(m=sr["m"], n=sr["n"], U=sr["U"], W=sr["W"], V=sr["V"], Ui=sr["Ui"], Uj=sr["Uj"], photo=sr["photo"], feature=sr["feature"])
    sa = [sr["Ref"], sr["ScaP"], sr["Fix"], sr["Sign"], sr["FixBlk"]] if mono else None
    st = schur_reference_solve(J, sr["ea"], sr["eb"], mono, sa)
    # (the first frame of a node is not part of the local-map file format: the reference's reader sets FRef = Ref, Imp.cpp:3053 / 6668 --
    # the joint map keeps End's, Imp.cpp:2624 / 7371, which is carried here)
    out = dict(J, stno=D["joint.stno"], stVal=st, FBlock=D["joint.FBlock"], Ref=int(D["joint.Ref"][0]), FRef=int(A["FRef"]))
    if mono:
        for k in ("ScaP", "Fix", "Sign"):
            out[k] = int(D[f"joint.{k}"][0])
        out["FScaP"], out["FFix"] = int(A["FScaP"]), int(A["FFix"])
    return out


def ref_reanchor(typ, mono, G, tmp):
    fa, fo = os.path.join(tmp, "A.txt"), os.path.join(tmp, "o.bin")
    write_map(fa, G, mono)
    args = [REF_DUMP, "trans", typ, fa, str(G["FRef"])]
    if mono:
        args += [str(G["FScaP"]), str(G["FFix"])]
    subprocess.check_call(args + [fo])
    o = sub(read_dump(fo), "out")
    o["FRef"] = G["FRef"]
    if mono:
        o["FScaP"], o["FFix"] = G["FScaP"], G["FFix"]
    return o


def run(name):
    typ, kw = CHAINS[name]
    mono = typ == "Monocular"
    maps = make_set(typ, kw)
    LM = [po.localmap_to_dict(m) for m in maps]
    store = {"type": np.array(typ), "N": np.array(len(LM)), "generator": np.array(repr(sorted(kw.items())))}
    for k, v in kw.items():
        store[f"gen.{k}"] = np.array(v)
    count, L = len(LM), 0
    t0 = time.time()
    with tempfile.TemporaryDirectory(dir="/dev/shm" if os.path.isdir("/dev/shm") else None) as tmp:
        while count > 1:
            N2 = count % 2
            count = int(count / 2.0 + 0.5)
            for i in range(count):
                num = 2 if (i < count - 1 or N2 == 0) else 1
                G = LM[2 * i]
                if num == 2:
                    G = ref_join(typ, mono, LM[2 * i], LM[2 * i + 1], tmp)
                if (i + 1) % 2 == 0 and G["Ref"] > G["FRef"]:       # Imp.cpp:1997 / 6576
                    G = ref_reanchor(typ, mono, G, tmp)
                LM[i] = G
            L += 1
            # the poses of every node of the level, in node order: where a failing comparison starts to differ
            if count <= 4:  # (of the last three levels)
                store[f"level{L}.pose_ids"] = np.concatenate([-np.asarray(LM[i]["stno"])[:6 * LM[i]["m"]:6] for i in range(count)]).astype(np.int32)
                store[f"level{L}.poses"] = np.concatenate([np.asarray(LM[i]["stVal"])[:6 * LM[i]["m"]] for i in range(count)])
            print(f"{name}: level {L} done, {count} nodes, largest {max(LM[i]['m'] for i in range(count))} poses, {time.time() - t0:.0f} s", flush=True)
        G = LM[0]
        if G["Ref"] > G["FRef"]:                                     # Imp.cpp:2039 / 6604
            G = ref_reanchor(typ, mono, G, tmp)
    store["levels"] = np.array(L)
    for k in ("Ref", "FRef", "m", "n") + (("ScaP", "Fix", "Sign") if mono else ()):
        store[f"result.{k}"] = np.array(int(G[k]))
    store["result.stno"] = np.asarray(G["stno"], np.int32)
    store["result.stVal"] = np.asarray(G["stVal"], np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **store)
    print(f"{path}: {len(maps)} maps, {L} levels, final map {G['m']} poses / {G['n']} features, {os.path.getsize(path) / 1024:.0f} KiB, {time.time() - t0:.0f} s")


if __name__ == "__main__":
    po.build()
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    for nm in (sys.argv[1:] or ["stereo", "mono"]):
        run({"stereo": "chain_stereo_n512", "mono": "chain_mono_n200", "stereo2048": "chain_stereo_n2048", "mono768": "chain_mono_n768"}.get(nm, nm))
