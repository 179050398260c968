"""Shared helpers of the parity tests."""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
MAPKEYS = ("Ref", "FRef", "m", "n", "ScaP", "Fix", "Sign", "FScaP", "FFix", "stno", "stVal", "U", "Ui", "Uj", "W",
           "photo", "feature", "V", "FBlock")


# fixtures made by the REAL reference (tests/golden/make_golden.py): whole small trees (every join, every re-anchoring transform,
# inputs and result), and the top joins of two mid-size trees (m = 64-90 poses, lap closures, features with 20-30 W blocks)
GOLD_SMALL = ["stereo_n2.npz", "stereo_n3.npz", "stereo_n5.npz", "stereo_n8.npz", "mono_n2.npz", "mono_n3.npz", "mono_n5.npz", "mono_n8.npz"]
GOLD_MID = ["stereo_n64_top1.npz", "mono_n88_top2.npz"]
# ... and of five sets with LONG tracks (every point visible for 34-44 frames): the one tile of 128 features of their top join is seen by
# 48 / 62 / 64 (Stereo) and 47 / 61 (Mono) poses, features have runs of up to 41-60 W blocks -- what the 48- and the 64-slot variant of
# the device's Schur panel kernel and its tile lists (k_schur_lists) take, and (64 poses: one more than the widest panel holds beside the
# right-hand side's two rows) the per-feature kernel k_schur_w; expected PANEL_SLOTS[name] = the variant's width, 0 = k_schur_w
GOLD_WIDE = ["stereo_n48_wide_top1.npz", "stereo_n62_wide_top1.npz", "stereo_n64_wide_top1.npz", "mono_n46_wide_top1.npz", "mono_n60_wide_top1.npz"]
PANEL_SLOTS = {"stereo_n48_wide_top1.npz": 48, "stereo_n62_wide_top1.npz": 64, "stereo_n64_wide_top1.npz": 0, "mono_n46_wide_top1.npz": 48, "mono_n60_wide_top1.npz": 64}
PANEL_RANGE = {48: (33, 48), 64: (49, 63), 0: (64, 1 << 30)}  # poses of a tile -> variant (lsfm_schur_panel.hip PmShared::CAP)


# whole mid-size trees evaluated by the REAL reference (every transform, every assembly) + exact solves of the reference-assembled systems
# (schur_reference_solve): tests/golden/make_chain_golden.py.  No arithmetic of the oracle or the library is in them.
GOLD_CHAIN = ["chain_stereo_n512", "chain_mono_n200", "chain_stereo_n2048", "chain_mono_n768"]
# Where fp64 itself stops: on the 768-map monocular chain three fp64-faithful evaluations of the reference's algorithm -- the chain (real
# reference transforms / assemblies, exact solves), the oracle, the oracle with every solve in long double -- lie 0.75e-6 .. 1.9e-6 apart
# on the pose parameters (features 0.9e-6 .. 2.9e-6): the state is rounded to fp64 between ten levels of a scale-gauged chain.  A result is held to
# max(BASELINE.json's 1e-6, 2 x that spread) there, like the full-size sets are to their oracle-twin floor; the other chains to 1e-6.
CHAIN_FLOOR = {"chain_mono_n768": 2.9e-6}


def chain_bar(name, default, factor=2.0):
    return max(default, factor * CHAIN_FLOOR.get(name, 0.0))


def chain_set(name):
    """(type, mono, local maps, fixture) of a reference-chain fixture: the set is re-made from the generator's stored arguments"""
    from linearsfm_amd import synth
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    typ = str(z["type"])
    kw = {k[4:]: z[k].item() for k in z.files if k.startswith("gen.")}
    maps = synth.make_mono_set(**kw) if typ == "Monocular" else synth.make_stereo_set(**kw)
    return typ, typ == "Monocular", maps, z


def tile_pose_counts(photo, feature, n, tile=128):
    """number of distinct poses that see each tile of `tile` consecutive features (what selects the width of K9's panel)"""
    photo, feature = np.asarray(photo), np.asarray(feature)
    return [len(np.unique(photo[(feature >= t0) & (feature < t0 + tile)])) for t0 in range(0, int(n), tile)]


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


def pose_param_true_rel_err(stA, stB, stno):
    """max over pose scalars of |a-b| / |b| -- the relative error proper, without the unit floor of pose_param_err; scalars
    smaller than 1e-3 of the largest of their kind (translations / angles) are held against that 1e-3 instead (a relative
    error of a number that happens to be ~0 says nothing)."""
    stno = np.asarray(stno)
    mask = stno <= 0
    a, b = np.asarray(stA)[mask].reshape(-1, 6), np.asarray(stB)[mask].reshape(-1, 6)
    if not a.size:
        return 0.0
    worst = 0.0
    for cols in (slice(0, 3), slice(3, 6)):
        fl = 1e-3 * max(1e-300, float(np.max(np.abs(b[:, cols]))))
        worst = max(worst, float(np.max(np.abs(a[:, cols] - b[:, cols]) / np.maximum(fl, np.abs(b[:, cols])))))
    return worst


def feat_param_err(stA, stB, stno):
    stno = np.asarray(stno)
    mask = stno > 0
    a, b = np.asarray(stA)[mask], np.asarray(stB)[mask]
    return float(np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b)))) if a.size else 0.0


def golden_system(z, j):
    """The system the REAL reference assembled for join j of a golden file and handed to lmj_solveLinearSFM* (captured by
    oracle/ref_harness.cpp): joint arrays, right-hand sides, Mono call-site arguments."""
    J = dict(m=int(z[f"join{j}.solve.m"][0]), n=int(z[f"join{j}.solve.n"][0]), U=z[f"join{j}.solve.U"].reshape(-1, 36),
             W=z[f"join{j}.solve.W"].reshape(-1, 18), V=z[f"join{j}.solve.V"].reshape(-1, 9), Ui=z[f"join{j}.solve.Ui"],
             Uj=z[f"join{j}.solve.Uj"], photo=z[f"join{j}.solve.photo"], feature=z[f"join{j}.solve.feature"])
    mono = str(z["type"]) == "Monocular"
    sa = [int(z[f"join{j}.solve.{k}"][0]) for k in ("Ref", "ScaP", "Fix", "Sign", "FixBlk")] if mono else None
    return J, z[f"join{j}.solve.ea"], z[f"join{j}.solve.eb"], mono, sa


def _inv3_longdouble(M):
    """[n,3,3] long double inverse by the adjugate"""
    a = M.astype(np.longdouble)
    c = np.empty_like(a)
    c[:, 0, 0] = a[:, 1, 1] * a[:, 2, 2] - a[:, 1, 2] * a[:, 2, 1]
    c[:, 0, 1] = a[:, 0, 2] * a[:, 2, 1] - a[:, 0, 1] * a[:, 2, 2]
    c[:, 0, 2] = a[:, 0, 1] * a[:, 1, 2] - a[:, 0, 2] * a[:, 1, 1]
    c[:, 1, 0] = a[:, 1, 2] * a[:, 2, 0] - a[:, 1, 0] * a[:, 2, 2]
    c[:, 1, 1] = a[:, 0, 0] * a[:, 2, 2] - a[:, 0, 2] * a[:, 2, 0]
    c[:, 1, 2] = a[:, 0, 2] * a[:, 1, 0] - a[:, 0, 0] * a[:, 1, 2]
    c[:, 2, 0] = a[:, 1, 0] * a[:, 2, 1] - a[:, 1, 1] * a[:, 2, 0]
    c[:, 2, 1] = a[:, 0, 1] * a[:, 2, 0] - a[:, 0, 0] * a[:, 2, 1]
    c[:, 2, 2] = a[:, 0, 0] * a[:, 1, 1] - a[:, 0, 1] * a[:, 1, 0]
    det = a[:, 0, 0] * c[:, 0, 0] + a[:, 0, 1] * c[:, 1, 0] + a[:, 0, 2] * c[:, 2, 0]
    return c / det[:, None, None]


def dense_reference_solve(J, ea, eb, mono, sa=None, IV=None):
    """Expected value of lmj_solveLinearSFM{Stereo,Mono} that owes nothing to the oracle's or the library's solver: the FULL
    normal equations [[U, W], [W^T, V]] x = [ea; eb] of the assembled system (no Schur complement, no sparse factorisation),
    solved densely by LAPACK (LU) and refined with residuals in extended precision until the correction is below 1e-17
    relative -- i.e. the exact solution rounded to fp64.  Mono: the 6 scalars of the reference pose (block sa[0]) and scalar
    sa[2] (Fix) are removed, their solution is 0 and finally x[Fix] = Sign (Imp.cpp:6981-7026).
    IV (optional, [n,9]): the output of the reference's own pba_inverseV for this V (fixture `parts.IV`).  The reference
    eliminates the features with that matrix -- the computed 3x3 inverse, symmetrised from its upper triangle
    (Imp.cpp:3027-3040) -- as if it were V^-1 exactly; with IV given the feature blocks of the dense system are IV^-1 (in
    extended precision), so that the expected value is the exact solution of the very system the reference's algebra
    solves (the two differ by cond(V) * 1e-16, which the camera system of a monocular join amplifies to ~1e-11).
    Returns the state vector in the reference's layout (6m pose scalars, 3n feature scalars)."""
    import scipy.linalg as sl
    from refdump import dense_info
    m, n = int(J["m"]), int(J["n"])
    A = dense_info(J)
    A = np.triu(A) + np.triu(A, 1).T  # the reference reads the upper triangle of a diagonal block only (Imp.cpp:2224-2229)
    b = np.concatenate([np.asarray(ea, np.float64), np.asarray(eb, np.float64)])
    keep = np.ones(6 * m + 3 * n, bool)
    if mono:
        keep[6 * sa[0]:6 * sa[0] + 6] = False
        keep[sa[2]] = False
    Al = A.astype(np.longdouble)
    if IV is not None:
        Ve = _inv3_longdouble(np.asarray(IV).reshape(-1, 3, 3))
        for f in range(n):
            Al[6 * m + 3 * f:6 * m + 3 * f + 3, 6 * m + 3 * f:6 * m + 3 * f + 3] = Ve[f]
    Al, bl = Al[np.ix_(keep, keep)], b[keep].astype(np.longdouble)
    lu = sl.lu_factor(np.asarray(Al, np.float64))
    x = sl.lu_solve(lu, b[keep])
    xl = x.astype(np.longdouble)
    for _ in range(8):
        r = bl - Al @ xl
        dx = sl.lu_solve(lu, np.asarray(r, np.float64))
        xl = xl + dx.astype(np.longdouble)
        if np.max(np.abs(dx)) <= 1e-17 * np.max(np.abs(x)):
            break
    out = np.zeros(6 * m + 3 * n)
    out[keep] = np.asarray(xl, np.float64)
    if mono:
        out[sa[2]] = sa[3]
    return out


def schur_reference_solve(J, ea, eb, mono, sa=None, tol=1e-17, max_it=12):
    """The same expected value as dense_reference_solve -- the exact solution of the FULL assembled normal equations
    [[U, W], [W^T, V]] x = [ea; eb], rounded to fp64 -- for systems whose dense full matrix does not fit (hundreds of poses, tens
    of thousands of features).  Nothing of the oracle or the library: the residual of the FULL system is formed in long double
    from the blocks as they are (the upper triangle of a diagonal U block mirrored, Imp.cpp:2224-2229; entries with equal
    coordinates add up), and corrected through a dense LAPACK Cholesky factor of the Schur complement S = U - W V^-1 W^T
    (numpy / scipy, fp64: a preconditioner -- the fixed point of the iteration is the full system's solution whatever rounding S
    carries) until the correction is below `tol` relative.  Mono: block sa[0] and scalar sa[2] removed, x[Fix] = Sign at the end."""
    import scipy.linalg as sl
    import scipy.sparse as sp
    m, n = int(J["m"]), int(J["n"])
    U = np.asarray(J["U"], np.float64).reshape(-1, 6, 6)
    Ui, Uj = np.asarray(J["Ui"]), np.asarray(J["Uj"])
    W = np.asarray(J["W"], np.float64).reshape(-1, 6, 3)
    ph, fe = np.asarray(J["photo"]), np.asarray(J["feature"])
    V = np.asarray(J["V"], np.float64).reshape(-1, 3, 3)
    # U as the symmetric matrix the reference's solver reads: a diagonal block by its upper triangle
    Ud = U.copy()
    dg = Ui == Uj
    Ud[dg] = np.triu(Ud[dg]) + np.transpose(np.triu(Ud[dg], 1), (0, 2, 1))
    keep = np.ones(6 * m, bool)
    if mono:
        keep[6 * sa[0]:6 * sa[0] + 6] = False
        keep[sa[2]] = False
    Ul, Wl, Vl = Ud.astype(np.longdouble), W.astype(np.longdouble), V.astype(np.longdouble)
    eal, ebl = np.asarray(ea, np.float64).astype(np.longdouble), np.asarray(eb, np.float64).astype(np.longdouble)

    def full_residual(xp, xf):
        """[ea - U xp - W xf ; eb - W^T xp - V xf] in long double"""
        rp = eal.copy().reshape(m, 6)
        xp6, xf3 = xp.reshape(m, 6), xf.reshape(n, 3)
        np.add.at(rp, Ui, -np.einsum("kij,kj->ki", Ul, xp6[Uj]))
        off = ~dg
        np.add.at(rp, Uj[off], -np.einsum("kji,kj->ki", Ul[off], xp6[Ui[off]]))
        np.add.at(rp, ph, -np.einsum("kij,kj->ki", Wl, xf3[fe]))
        rf = ebl.copy().reshape(n, 3) - np.einsum("kij,kj->ki", Vl, xf3)
        np.add.at(rf, fe, -np.einsum("kji,kj->ki", Wl, xp6[ph]))
        return rp.reshape(-1), rf.reshape(-1)

    # the preconditioner: dense Schur complement in fp64
    Vi = np.linalg.inv(V)
    rows = (6 * ph[:, None, None] + np.arange(6)[None, :, None]).repeat(3, 2)
    cols = (3 * fe[:, None, None] + np.arange(3)[None, None, :]).repeat(6, 1)
    Ws = sp.csr_matrix((W.ravel(), (rows.ravel(), cols.ravel())), shape=(6 * m, 3 * n))
    bi = (3 * np.arange(n)[:, None, None] + np.arange(3)[None, :, None]).repeat(3, 2)
    bj = (3 * np.arange(n)[:, None, None] + np.arange(3)[None, None, :]).repeat(3, 1)
    Vis = sp.csr_matrix((Vi.ravel(), (bi.ravel(), bj.ravel())), shape=(3 * n, 3 * n))
    S = np.zeros((6 * m, 6 * m))
    ur = (6 * Ui[:, None, None] + np.arange(6)[None, :, None]).repeat(6, 2)
    uc = (6 * Uj[:, None, None] + np.arange(6)[None, None, :]).repeat(6, 1)
    np.add.at(S, (ur.ravel(), uc.ravel()), Ud.ravel())
    off = ~dg
    np.add.at(S, (uc[off].ravel(), ur[off].ravel()), Ud[off].ravel())
    S -= (Ws @ Vis @ Ws.T).toarray()
    S = 0.5 * (S + S.T)
    cf = sl.cho_factor(S[np.ix_(keep, keep)])

    def correct(rp, rf):
        rp64, rf64 = np.asarray(rp, np.float64), np.asarray(rf, np.float64)
        g = rp64 - Ws @ (Vis @ rf64)
        dp = np.zeros(6 * m)
        dp[keep] = sl.cho_solve(cf, g[keep])
        df = Vis @ (rf64 - Ws.T @ dp)
        return dp, df

    xp, xf = np.zeros(6 * m, np.longdouble), np.zeros(3 * n, np.longdouble)
    for it in range(max_it):
        rp, rf = full_residual(xp, xf)
        rp[~keep] = 0
        dp, df = correct(rp, rf)
        xp += dp
        xf += df
        scale = max(np.max(np.abs(xp)), np.max(np.abs(xf)) if n else 0.0, 1e-300)
        if max(np.max(np.abs(dp)), np.max(np.abs(df)) if n else 0.0) <= tol * scale:
            break
    out = np.concatenate([np.asarray(xp, np.float64), np.asarray(xf, np.float64)])
    if mono:
        out[sa[2]] = sa[3]
    return out


def reference_writer_bytes(stno, st):
    """What the reference's lmj_SaveStateVector (LinearSFMImp.cpp:2102-2117) and lmj_SavePoses_3DPF (7876-7967) write for a state, as
    bytes: (state, pose, feature).  A Python statement of the two writers for states no fixture holds (the CLI's own output on the GPU
    box); pinned to the real writers' bytes on every case of tests/golden/writers.npz by
    test_oracle_cpu.py::test_python_statement_of_the_writers_vs_reference_bytes.  `%lf` = Python's `%f`: both print the exactly
    rounded decimal expansion of the binary double."""
    stno, st = np.asarray(stno), np.asarray(st, np.float64)
    state = "".join("%d %f\n" % (int(a), float(b)) for a, b in zip(stno, st))
    poses, feats = {}, {}
    i = 0
    while i < len(stno):
        if stno[i] <= 0:
            poses[-int(stno[i])] = i          # std::map: a repeated id keeps its last occurrence
            i += 6
        else:
            feats[int(stno[i])] = i
            i += 3
    pose = "".join("%d  %f  %f  %f %f  %f  %f\n" % ((k,) + tuple(float(v) for v in st[poses[k]:poses[k] + 6])) for k in sorted(poses))
    feat = "".join("%d  %f  %f %f\n" % ((k,) + tuple(float(v) for v in st[feats[k]:feats[k] + 3])) for k in sorted(feats))
    return state.encode(), pose.encode(), feat.encode()
