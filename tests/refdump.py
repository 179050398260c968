"""Reader for the tagged binary written by oracle/_ref/ref_dump (oracle/ref_harness.cpp) and canonical forms
used to compare maps irrespective of how duplicate block coordinates are split over slots."""
import numpy as np


def read_dump(path):
    out = {}
    with open(path, "rb") as f:
        while True:
            line = f.readline()
            if not line:
                break
            name, dt, cnt = line.decode().split()
            cnt = int(cnt)
            dtype = np.float64 if dt == "f8" else np.int32
            a = np.frombuffer(f.read(cnt * np.dtype(dtype).itemsize), dtype=dtype).copy()
            out[name] = a
    return out


def sub(d, prefix):
    """{'end.m': ..} -> {'m': ..} with scalars unwrapped and blocks reshaped."""
    o = {}
    for k, v in d.items():
        if k.startswith(prefix + "."):
            kk = k[len(prefix) + 1:]
            o[kk] = v
    for k in ("m", "n", "nU", "nW", "Ref", "FRef", "ScaP", "Fix", "Sign", "FScaP", "FFix", "FixBlk"):
        if k in o:
            o[k] = int(o[k][0])
    for k, w in (("U", 36), ("W", 18), ("V", 9)):
        if k in o:
            o[k] = o[k].reshape(-1, w)
    return o


def dense_info(d):
    """Full symmetric information matrix of a map dict (poses then features), duplicates summed."""
    m, n = int(d["m"]), int(d["n"])
    N = 6 * m + 3 * n
    I = np.zeros((N, N))
    U = np.asarray(d["U"]).reshape(-1, 6, 6)
    for k in range(len(d["Ui"])):
        a, b = int(d["Ui"][k]), int(d["Uj"][k])
        I[6 * a:6 * a + 6, 6 * b:6 * b + 6] += U[k]
        if a != b:
            I[6 * b:6 * b + 6, 6 * a:6 * a + 6] += U[k].T
    W = np.asarray(d["W"]).reshape(-1, 6, 3)
    for k in range(len(d["photo"])):
        p, f = int(d["photo"][k]), int(d["feature"][k])
        I[6 * p:6 * p + 6, 6 * m + 3 * f:6 * m + 3 * f + 3] += W[k]
        I[6 * m + 3 * f:6 * m + 3 * f + 3, 6 * p:6 * p + 6] += W[k].T
    V = np.asarray(d["V"]).reshape(-1, 3, 3)
    for f in range(n):
        I[6 * m + 3 * f:6 * m + 3 * f + 3, 6 * m + 3 * f:6 * m + 3 * f + 3] += V[f]
    return I
