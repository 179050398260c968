#!/usr/bin/env python3
"""Golden BYTES of the reference's own result writers (tests/golden/writers.npz).

Runs the REAL lmj_SaveStateVector (LinearSFMImp.cpp:2102-2117) and lmj_SavePoses_3DPF (7876-7967) -- public, CHOLMOD-free
methods of the reference TU compiled where it lies (oracle/_ref/ref_dump, mode `save`, oracle/ref_harness.cpp) -- on a
handful of state vectors and stores, per case, the inputs (stno, stVal) and the three text files as uint8 arrays.  The
product's lsfm_save_state / lsfm_save_poses (csrc/lsfm_io.cpp) and the CLI's -st / -p / -f files are held to these bytes
(tests/test_oracle_cpu.py::test_writers_byte_for_byte_vs_reference, tests/test_gpu_cli.py).  Authoring container only
(needs /root/reference); the fixture is data: inputs and the reference's outputs.

Cases: the final states of four small golden trees (Stereo / Mono); and hand-made states with what the writers' logic
turns on -- ids out of order (std::set iterates sorted), repeated ids (std::map keeps the LAST occurrence), negative zeros
("-0.000000"), label 0 (a pose: `stno <= 0`), values that round at the sixth decimal, huge / tiny magnitudes, an empty
state; each also with only the pose path or only the feature path (the method takes NULL for either, Imp.cpp:7885-7898).

Usage:  python tests/golden/make_writer_golden.py        (from the repo root)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
REF_DUMP = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
HERE = os.path.dirname(os.path.abspath(__file__))


def cases():
    out = {}
    for name in ("stereo_n5", "stereo_n3", "mono_n5", "mono_n8"):
        z = np.load(os.path.join(HERE, name + ".npz"))
        out[name] = (z["result.stno"].astype(np.int32), z["result.stVal"].astype(np.float64))
    rng = np.random.default_rng(6)

    def state(poses, feats, order=None):
        stno, st = [], []
        items = [("p", i) for i in poses] + [("f", i) for i in feats]
        if order is not None:
            items = [items[k] for k in order]
        for kind, i in items:
            k = 6 if kind == "p" else 3
            stno += [(-i if kind == "p" else i)] * k
            st += list(rng.normal(scale=3.0, size=k))
        return np.array(stno, np.int32), np.array(st, np.float64)

    # ids out of order, poses and features interleaved
    out["shuffled"] = state([7, 2, 11, 0, 5], [40, 3, 17, 9, 1000003], order=[5, 0, 6, 1, 7, 2, 8, 3, 9, 4])
    # repeated ids: the last occurrence is the one written
    out["repeats"] = state([3, 1, 3, 2, 1], [8, 8, 5, 8])
    stno, st = state([1, 2], [4, 6, 5])
    st[:] = [-0.0, 0.0, -1e-9, 1e-9, 0.4999995, 0.5000005, -0.0000005, 0.0000015, 123456789.123456789, -1e15, 1e-300, -1e300,
             2.5e-7, -2.5e-7, 1.0000005, -0.0, 3.14159265358979, -2.718281828459045, 0.1, 0.2, 0.3]
    out["edge_values"] = (stno, st)
    out["empty"] = (np.zeros(0, np.int32), np.zeros(0, np.float64))
    out["poses_only_state"] = state([4, 1], [])
    out["features_only_state"] = state([], [9, 2, 5])
    return out


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    store = {}
    names = []
    with tempfile.TemporaryDirectory() as tmp:
        for name, (stno, st) in cases().items():
            fi = os.path.join(tmp, "in.bin")
            with open(fi, "wb") as f:
                for nm, a in (("stno", stno), ("stVal", st)):
                    f.write(f"{nm} {'f8' if a.dtype == np.float64 else 'i4'} {a.size}\n".encode())
                    f.write(np.ascontiguousarray(a).tobytes())
            store[f"{name}.stno"], store[f"{name}.stVal"] = stno, st
            for which in ("both", "pose", "feat"):
                fp, ff, fs = (os.path.join(tmp, f"{which}_{x}.txt") for x in ("pose", "feat", "state"))
                subprocess.check_call([REF_DUMP, "save", which, fi, fp, ff, fs])
                assert os.path.exists(fp) == (which != "feat") and os.path.exists(ff) == (which != "pose")
                if which != "feat":
                    store[f"{name}.{which}.pose"] = np.frombuffer(open(fp, "rb").read(), np.uint8)
                if which != "pose":
                    store[f"{name}.{which}.feat"] = np.frombuffer(open(ff, "rb").read(), np.uint8)
                if which == "both":
                    store[f"{name}.state"] = np.frombuffer(open(fs, "rb").read(), np.uint8)
                for p in (fp, ff, fs):
                    if os.path.exists(p):
                        os.remove(p)
            names.append(name)
    store["names"] = np.array(names)
    path = os.path.join(HERE, "writers.npz")
    np.savez_compressed(path, **store)
    print(f"{path}: {len(names)} cases, {os.path.getsize(path) / 1024:.0f} KiB")


if __name__ == "__main__":
    main()
