"""Shared helpers of the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAPKEYS = ("Ref", "FRef", "m", "n", "ScaP", "Fix", "Sign", "FScaP", "FFix", "stno", "stVal", "U", "Ui", "Uj", "W",
           "photo", "feature", "V", "FBlock")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)


def get_map(z, prefix):
    d = {}
    for k in MAPKEYS:
        v = z[f"{prefix}.{k}"]
        d[k] = int(v) if v.ndim == 0 else v
    d["U"] = np.asarray(d["U"]).reshape(-1, 36)
    d["W"] = np.asarray(d["W"]).reshape(-1, 18)
    d["V"] = np.asarray(d["V"]).reshape(-1, 9)
    return d


def ref_map(z, prefix, m=None):
    """A map dumped by oracle/_ref/ref_dump ('end.*', 'out.*')."""
    d = {}
    for k in ("m", "n", "nU", "nW", "Ref", "FRef"):
        d[k] = int(z[f"{prefix}.{k}"][0])
    for k in ("stno", "stVal", "Ui", "Uj", "photo", "feature", "FBlock"):
        d[k] = z[f"{prefix}.{k}"]
    d["U"] = z[f"{prefix}.U"].reshape(-1, 36)
    d["W"] = z[f"{prefix}.W"].reshape(-1, 18)
    d["V"] = z[f"{prefix}.V"].reshape(-1, 9)
    for k in ("ScaP", "Fix", "Sign", "FScaP", "FFix"):
        if f"{prefix}.{k}" in z:
            d[k] = int(z[f"{prefix}.{k}"][0])
    return d


def canon_u(d):
    """(Ui,Uj) -> summed 6x6 block, independent of how duplicates are split over slots."""
    out = {}
    U = np.asarray(d["U"]).reshape(-1, 6, 6)
    for k in range(len(d["Ui"])):
        key = (int(d["Ui"][k]), int(d["Uj"][k]))
        out[key] = out.get(key, 0) + U[k]
    return out


def rel_err(a, b):
    a = np.asarray(a, np.float64).ravel()
    b = np.asarray(b, np.float64).ravel()
    assert a.shape == b.shape, (a.shape, b.shape)
    if a.size == 0:
        return 0.0
    return float(np.max(np.abs(a - b)) / max(1e-300, np.max(np.abs(b))))


def assert_maps_close(a, b, tol, what="", canonical_u=False):
    """a: implementation under test, b: expected.  Structure must be identical, values within tol (relative to the
    largest magnitude of each array -- information blocks span many orders of magnitude)."""
    for k in ("m", "n", "Ref", "FRef"):
        assert int(a[k]) == int(b[k]), (what, k, a[k], b[k])
    for k in ("stno", "photo", "feature", "FBlock", "Ui", "Uj"):
        assert np.array_equal(np.asarray(a[k]).ravel(), np.asarray(b[k]).ravel()), (what, k)
    for k in ("stVal", "W", "V"):
        e = rel_err(a[k], b[k])
        assert e < tol, (what, k, e)
    if canonical_u:
        ca, cb = canon_u(a), canon_u(b)
        assert ca.keys() == cb.keys(), what
        scale = max(np.abs(v).max() for v in cb.values())
        for key in cb:
            assert np.abs(ca[key] - cb[key]).max() / scale < tol, (what, "U", key)
    else:
        e = rel_err(a["U"], b["U"])
        assert e < tol, (what, "U", e)


def pose_param_err(stA, stB, stno):
    """max over pose scalars of |a-b| / max(1, |b|): 'relative on pose parameters' with unit floor for angles/zeros."""
    stno = np.asarray(stno)
    mask = stno <= 0
    a, b = np.asarray(stA)[mask], np.asarray(stB)[mask]
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def feat_param_err(stA, stB, stno):
    stno = np.asarray(stno)
    mask = stno > 0
    a, b = np.asarray(stA)[mask], np.asarray(stB)[mask]
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def oracle_noise_floor(oracle, dicts, mono, ref_stval, stno):
    """How far two valid fp64 evaluations of the reference path differ on this input: the oracle re-run with its
    Cholesky forced to a different (equally valid) elimination order (ORC_ORDER, oracle/lsfm_chol.c).  On long chains
    the camera systems are so ill-conditioned that this floor, not the implementation, limits any parity number."""
    import os
    old = os.environ.get("ORC_ORDER")
    os.environ["ORC_ORDER"] = "1"
    try:
        alt, _, rc = oracle.divide_conquer(dicts, mono)
    finally:
        if old is None:
            os.environ.pop("ORC_ORDER")
        else:
            os.environ["ORC_ORDER"] = old
    assert rc == 0
    return max(pose_param_err(alt["stVal"], ref_stval, stno), feat_param_err(alt["stVal"], ref_stval, stno))
