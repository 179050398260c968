"""BASELINE.json configs[3] and configs[4] under `-m gpu`, the whole-tree reference fixtures, the device's block pattern of S
against the real pba_constructAuxCSS{LM,GN}, and a reload of a resident tree with values its plans do not fit.

Tolerances are fixed.  TREE_TOL = 1e-6 is BASELINE.json's bar on pose parameters and is used wherever two fp64 evaluations
of the reference path themselves agree to well below it (the "floor" quoted in each test was measured with the oracle and
its long-double twin, tools/noise_floor.py, and is printed again by the test).  Where the oracle takes minutes (16 384
maps) the checks are the size-independent ones: every camera system solved to a direct solve's residual, re-anchoring
there and back is the identity, the information quadratic form is preserved."""
import os

import numpy as np
import pytest

from common import GOLD_MID, GOLD_SMALL, GOLD_WIDE, feat_param_err, get_map, golden_system, load_golden, pose_param_err, rel_err
from linearsfm_amd import synth

pytestmark = pytest.mark.gpu

TREE_TOL = 1e-6
HOST_THREADS = max(1, min(32, os.cpu_count() or 1))


def _same_structure(got, exp):
    assert np.array_equal(got["stno"], exp["stno"])
    assert got["Ref"] == exp["Ref"] and got["FRef"] == exp["FRef"]
    for k in ("photo", "feature", "Ui", "Uj", "FBlock"):
        assert np.array_equal(got[k], exp[k]), k


# ---------------------------------------------------------------------------------------------------------------------
# configs[4], first half: the aerial monocular block (AP_Vaihingen stand-in)
# ---------------------------------------------------------------------------------------------------------------------
def test_aerial_block_mono_vs_oracle(ctx, oracle):
    """synth.CONFIGS['aerial']: 12 parallel strips of 20 frames, 30 % side overlap, every strip shares features with its
    neighbour along its whole length (cross-strip common features at every level of the join tree: a grid, not a chain).
    Floor (oracle vs long-double twin) 4.7e-8."""
    typ, maps = synth.make_config("aerial")
    assert typ == "Monocular" and len(maps) == 238
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    got, stats, rc = ctx.divide_conquer(dicts, True)
    assert rc == 0 and stats["not_converged"] == 0, stats
    exp, _, orc = oracle.divide_conquer(dicts, True, match_hash=True)
    assert orc == 0
    _same_structure(got, exp)
    for k in ("ScaP", "Fix", "Sign"):
        assert got[k] == exp[k], k
    ep, ef = pose_param_err(got["stVal"], exp["stVal"], exp["stno"]), feat_param_err(got["stVal"], exp["stVal"], exp["stno"])
    print(f"aerial: 238 maps, {got['m']} poses / {got['n']} features, pose parameter max rel err vs oracle {ep:.2e}, features {ef:.2e}, "
          f"{stats['t_total_ms']:.1f} ms, max rel residual {stats['max_rel_residual']:.1e}")
    assert ep < TREE_TOL and ef < TREE_TOL, (ep, ef)
    # the strips really are linked sideways: features of the first strip's maps reappear in maps of the second strip
    ids = [set(np.asarray(m.stno)[np.asarray(m.stno) > 0].tolist()) for m in maps]
    assert sum(len(ids[5] & ids[k]) for k in range(28, 38)) > 50


# ---------------------------------------------------------------------------------------------------------------------
# configs[3]: synthetic monocular 16k
# ---------------------------------------------------------------------------------------------------------------------
def test_synth16k_mono_prefix_vs_oracle(ctx, oracle):
    """The first 512 local maps of the synth-16k set against the oracle at the fixed 1e-6.  Why 512 and not more: a monocular
    chain joined by this algorithm loses about 1.5 decimal orders per doubling of its length whatever solves it -- the oracle
    and its long-double twin differ by 4.1e-8 on 512 maps, 1.8e-6 on 1024, 4.5e-5 on 2048 (tools/noise_floor.py; nearer /
    farther points, skip links between distant laps: all measured, none better) -- so beyond 512 maps a comparison at 1e-6
    with ANY fp64 evaluation of the reference path decides nothing.  The full 16 384 maps are covered by the
    size-independent properties of the next test."""
    typ, maps = synth.make_config("synth16k", 512)
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    got, stats, rc = ctx.divide_conquer(dicts, True)
    assert rc == 0 and stats["not_converged"] == 0, stats
    exp, _, orc = oracle.divide_conquer(dicts, True, match_hash=True, threads=HOST_THREADS)
    assert orc == 0
    _same_structure(got, exp)
    ep, ef = pose_param_err(got["stVal"], exp["stVal"], exp["stno"]), feat_param_err(got["stVal"], exp["stVal"], exp["stno"])
    print(f"synth16k[:512]: pose parameter max rel err vs oracle {ep:.2e}, features {ef:.2e}, {stats['t_total_ms']:.1f} ms")
    assert ep < TREE_TOL and ef < TREE_TOL, (ep, ef)


def _quad_form(m, x):
    """x^T I x of a map's information matrix (U upper blocks with duplicates adding up, W, V) for a state-sized x."""
    M, n = int(m["m"]), int(m["n"])
    xp, xf = x[:6 * M].reshape(M, 6), x[6 * M:].reshape(n, 3)
    U, W, V = np.asarray(m["U"]).reshape(-1, 6, 6), np.asarray(m["W"]).reshape(-1, 6, 3), np.asarray(m["V"]).reshape(-1, 3, 3)
    ui, uj = np.asarray(m["Ui"]), np.asarray(m["Uj"])
    t = np.einsum("ki,kij,kj->k", xp[ui], U, xp[uj])
    q = float(np.sum(np.where(ui == uj, t, 2 * t)))
    q += 2 * float(np.sum(np.einsum("ki,kij,kj->k", xp[np.asarray(m["photo"])], W, xf[np.asarray(m["feature"])])))
    q += float(np.sum(np.einsum("ki,kij,kj->k", xf, V, xf)))
    return q


def test_synth16k_mono_full_size_properties(ctx):
    """All 16 384 local maps (16 386 poses, ~1 M features, 14 levels).  The oracle would take a quarter of an hour, so: every
    camera system of every level solved to the residual where the library calls a system converged (relative 1e-9); the result is in its first frame; re-anchoring the
    final map to a pose in the middle and back is the identity on the state (1e-9) and keeps the information quadratic form
    (1e-5 relative on random probes: the probe sums 34 M block products of both signs, two transforms deep -- measured 7e-7),
    i.e. the Mono transform at 16k poses / two hub columns inverts itself."""
    typ, maps = synth.make_config("synth16k")
    assert typ == "Monocular" and len(maps) == 16384
    t = ctx.tree_upload([m.__dict__ for m in maps], True)
    del maps
    try:
        ctx.tree_set_plans(t, False)
        # Every run analyses, and every one must end with all systems converged.  The root system of this tree is at the edge of
        # fp64: in one run out of fifteen the last 6x6 block of its top separator comes out slightly indefinite (the atomic sums
        # of 16 000 columns of updates land in another order every run); the factorisation then takes the pivot by its
        # magnitude, and a tree whose refinement still stalls is joined again (stats["attempts"] > 1) -- never reported as
        # converged when it is not.
        attempts = []
        for _ in range(4):
            stats, rc = ctx.tree_run(t)
            assert rc == 0 and stats["not_converged"] == 0 and stats["max_rel_residual"] < 1e-8, stats
            attempts.append(stats["attempts"])
        print("attempts per run:", attempts)
        assert max(attempts) <= 3
        out = ctx.tree_download(t)
    finally:
        ctx.tree_free(t)
    print(f"synth16k: {out['m']} poses / {out['n']} features / {out['nW']} W blocks, {stats['levels']} levels, {stats['t_total_ms']:.0f} ms, "
          f"max rel residual {stats['max_rel_residual']:.2e}")
    # (most systems end at 1e-12 .. 1e-14; the camera systems of a monocular chain this deep are conditioned ~1e10 and one or
    # two of them stop where the true residual stops shrinking instead: 1e-12 in one run, 1e-9 in the next -- the order of
    # the atomic sums differs.  The library counts a system as not converged above 1e-8, or above 1e-9 when it was still
    # shrinking; that count must be zero)
    assert stats["max_rel_residual"] < 1e-8, stats
    M = int(out["m"])
    assert M == 16386 and stats["levels"] == 14 and out["Ref"] == out["FRef"]
    st = np.asarray(out["stVal"])
    assert np.all(st[:6] == 0.0) and abs(st[6 + out["Fix"]]) == 1.0  # gauge of the first local map: Imp.cpp:7010-7026
    ids = -np.asarray(out["stno"])[:6 * M:6]
    k = M // 2
    other = int(ids[k])
    assert ids[k + 1] == other + 1 and ids[0] == out["Ref"] and ids[1] == out["ScaP"]
    # the new scale is the baseline to the next frame along its dominant axis in the new reference frame (as a local map's is)
    base = synth.rot_ypr(*st[6 * k + 3:6 * k + 6]) @ (st[6 * k + 6:6 * k + 9] - st[6 * k:6 * k + 3])
    there = ctx.transform(out, True, other, other + 1, int(np.argmax(np.abs(base))))
    back = ctx.transform(there, True, int(out["Ref"]), int(out["ScaP"]), int(out["Fix"]))
    assert np.array_equal(back["stno"], out["stno"])
    err = np.max(np.abs(np.asarray(back["stVal"]) - st) / np.maximum(1.0, np.abs(st)))
    assert err < 1e-9, err
    rng = np.random.default_rng(5)
    for _ in range(2):
        x = rng.normal(size=6 * M + 3 * int(out["n"]))
        x[:6] = 0.0
        x[6 + out["Fix"]] = 0.0  # the gauge scalars carry no information in either map
        a, b = _quad_form(out, x), _quad_form(back, x)
        assert abs(a - b) / abs(a) < 1e-5, (a, b)


# ---------------------------------------------------------------------------------------------------------------------
# configs[4], second half: synthetic 64k-frame stereo, fp64 and fp32-mixed
# ---------------------------------------------------------------------------------------------------------------------
def _run_both_precisions(ctx, dicts):
    t = ctx.tree_upload(dicts, False)
    try:
        s64, rc = ctx.tree_run(t)
        assert rc == 0 and s64["not_converged"] == 0, s64
        a = ctx.tree_download(t)
        ctx.set_precision(True)
        s32, rc = ctx.tree_run(t)
        assert rc == 0 and s32["not_converged"] == 0, s32
        b = ctx.tree_download(t)
    finally:
        ctx.set_precision(False)
        ctx.tree_free(t)
    assert s64["max_rel_residual"] < 1e-11 and s32["max_rel_residual"] < 1e-11, (s64, s32)
    assert s32["pcg_iterations"] > s64["pcg_iterations"]
    return a, b, s64, s32


def test_synth64k_stereo_4096_maps_fp64_and_mixed_vs_oracle(ctx, oracle):
    """The first 4096 local maps of the synth-64k Stereo set in both precisions of the library AGAINST THE ORACLE: fp64, and
    lsfm_set_precision(1) -- Cholesky factor kept and applied in fp32, residual correction in fp64 (BASELINE.json configs[4]).
    4096 maps because that is where the comparison still decides something: oracle vs its long-double twin 1.8e-7 here,
    1.2e-6 at 16 384 maps (tools/noise_floor.py) -- above the 1e-6 bar by itself."""
    typ, maps = synth.make_config("synth64k", 4096)
    assert typ == "Stereo"
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    del maps
    a, b, s64, s32 = _run_both_precisions(ctx, dicts)
    exp, _, orc = oracle.divide_conquer(dicts, False, match_hash=True, threads=HOST_THREADS)
    assert orc == 0
    _same_structure(a, exp)
    _same_structure(b, exp)
    e64 = pose_param_err(a["stVal"], exp["stVal"], exp["stno"])
    e32 = pose_param_err(b["stVal"], exp["stVal"], exp["stno"])
    f64 = feat_param_err(a["stVal"], exp["stVal"], exp["stno"])
    f32 = feat_param_err(b["stVal"], exp["stVal"], exp["stno"])
    print(f"synth64k[:4096]: fp64 vs oracle pose {e64:.2e} / features {f64:.2e}; mixed vs oracle pose {e32:.2e} / features {f32:.2e}; "
          f"steps {s32['pcg_iterations']} vs {s64['pcg_iterations']}; {s64['t_total_ms']:.0f} / {s32['t_total_ms']:.0f} ms")
    assert e64 < TREE_TOL and f64 < TREE_TOL, (e64, f64)
    assert e32 < TREE_TOL and f32 < TREE_TOL, (e32, f32)


@pytest.mark.parametrize("n_maps,levels", [(16384, 14), (65536, 16)])
def test_synth64k_stereo_fp64_and_mixed_properties(ctx, n_maps, levels):
    """16 384 local maps (16 384 poses, 1.05 M features, 14 levels) and the WHOLE set of BASELINE.json configs[4] (65 536 maps, 65 536
    poses, 4.19 M features, 16 levels) in both precisions.  Two fp64 evaluations of the reference path differ by 1.2e-6 at 16 384
    maps already (see above) and the oracle takes minutes, so: every system of every level solved to a direct solve's residual in
    both modes, identical structure, and the two modes -- same assembled systems, preconditioner in fp64 / fp32 -- agree far inside
    the bar (they differ only by what the refinement leaves: 1e-12 residuals)."""
    typ, maps = synth.make_config("synth64k", n_maps)
    a, b, s64, s32 = _run_both_precisions(ctx, maps)
    del maps
    assert int(a["m"]) == n_maps and s64["levels"] == levels and a["Ref"] == a["FRef"] == 1
    _same_structure(b, a)
    e = pose_param_err(b["stVal"], a["stVal"], a["stno"])
    f = feat_param_err(b["stVal"], a["stVal"], a["stno"])
    print(f"synth64k[:{n_maps}]: {a['n']} features, {a['nW']} W blocks; mixed vs fp64 pose {e:.2e} / features {f:.2e}; steps {s32['pcg_iterations']} vs "
          f"{s64['pcg_iterations']}; residuals {s64['max_rel_residual']:.1e} / {s32['max_rel_residual']:.1e}; {s64['t_total_ms']:.0f} / {s32['t_total_ms']:.0f} ms")
    assert e < 1e-7 and f < 1e-7, (e, f)


@pytest.mark.parametrize("config,n_maps", [("rs468", 466), ("aerial", 238)])
def test_mixed_precision_mono_vs_oracle(ctx, oracle, config, n_maps):
    """The fp32-mixed mode on the monocular sets against the oracle (not against the library's own fp64 run)."""
    typ, maps = synth.make_config(config, n_maps)
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    ctx.set_precision(True)
    try:
        got, stats, rc = ctx.divide_conquer(dicts, True)
    finally:
        ctx.set_precision(False)
    assert rc == 0 and stats["not_converged"] == 0, stats
    exp, _, orc = oracle.divide_conquer(dicts, True, match_hash=True, threads=HOST_THREADS)
    assert orc == 0
    _same_structure(got, exp)
    ep, ef = pose_param_err(got["stVal"], exp["stVal"], exp["stno"]), feat_param_err(got["stVal"], exp["stVal"], exp["stno"])
    print(f"{config} mixed: pose parameter max rel err vs oracle {ep:.2e}, features {ef:.2e}, {stats['pcg_iterations']} steps")
    assert ep < TREE_TOL and ef < TREE_TOL, (ep, ef)


# ---------------------------------------------------------------------------------------------------------------------
# whole trees of the reference fixtures; the device's pattern of S
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", GOLD_SMALL)
def test_whole_tree_from_fixture_inputs_to_fixture_result(ctx, name):
    """lsfm_divide_conquer on the fixture's input maps (`in*`) against the fixture's final map (`result.*`): the state that
    went through the REAL reference's transform and assembly at every join and re-anchoring (tests/golden/make_golden.py
    checks each of them against oracle/_ref/ref_dump while it follows lmj_PF3D_Divide_Conquer*'s loop, Imp.cpp:1932-2063),
    with the oracle's solves in between.  N = 5 and 8: an unpaired carry, re-anchored odd outputs, the final re-anchoring."""
    z = load_golden(name)
    mono = str(z["type"]) == "Monocular"
    maps = [get_map(z, f"in{k}") for k in range(int(z["N"]))]
    exp = get_map(z, "result")
    got, stats, rc = ctx.divide_conquer(maps, mono)
    assert rc == 0, stats
    _same_structure(got, exp)
    if mono:
        for k in ("ScaP", "Fix", "Sign"):
            assert got[k] == exp[k], k
    assert pose_param_err(got["stVal"], exp["stVal"], exp["stno"]) < 1e-8
    assert feat_param_err(got["stVal"], exp["stVal"], exp["stno"]) < 1e-8
    for k in ("W", "V"):
        assert rel_err(got[k], exp[k]) < 1e-8, k


def _pairs_from_csc(Ap, Aii):
    return [(int(Aii[k]), j) for j in range(len(Ap) - 1) for k in range(Ap[j], Ap[j + 1])]


@pytest.mark.parametrize("name", GOLD_SMALL + GOLD_MID + GOLD_WIDE)
def test_device_schur_pattern_vs_reference_aux_css(ctx, name):
    """The block pattern of S the DEVICE builds (hash set of pose pairs + U's pattern -> block CSR: lsfm_schur_pattern) on
    the index arrays of all 22 reference-assembled systems, against the pattern the REAL pba_constructAuxCSS{LM,GN}
    (Imp.cpp:2529-2549 / 7248-7280) listed for cholmod_amd from the reference's own mask (fixture parts.Ap / parts.Aii:
    upper blocks column by column; GN: without the reference pose's block row and column, later blocks renumbered)."""
    z = load_golden(name)
    for j in range(int(z["njoins"])):
        J, _, _, mono, sa = golden_system(z, j)
        m = J["m"]
        rowptr, colidx = ctx.schur_pattern(J)
        assert len(rowptr) == m + 1 and rowptr[0] == 0 and rowptr[m] == len(colidx)
        dev = []
        for p in range(m):
            cols = colidx[rowptr[p]:rowptr[p + 1]]
            assert len(cols) and cols[0] == p and np.all(np.diff(cols) > 0), (j, p)  # diagonal first, ascending, upper
            dev += [(p, int(q)) for q in cols]
        # the oracle's restatement of the mask (already pinned to the reference in the CPU suite) agrees block for block
        # (Stereo; a Mono join drops every block of the reference pose, the device keeps that row's diagonal block as a
        # placeholder -- its scalars are removed from the system -- so there the comparison is the one below)
        if not mono:
            assert np.array_equal(rowptr, z[f"join{j}.parts_in.rowptr"]) and np.array_equal(colidx, z[f"join{j}.parts_in.colidx"]), j
        if mono:
            ref = sa[0]
            dev = [(p - (p > ref), q - (q > ref)) for p, q in dev if p != ref and q != ref]
        exp = _pairs_from_csc(z[f"join{j}.parts.Ap"], z[f"join{j}.parts.Aii"])
        assert sorted(dev) == sorted(exp), (name, j)


# ---------------------------------------------------------------------------------------------------------------------
# a resident tree reloaded with values its plans do not fit
# ---------------------------------------------------------------------------------------------------------------------
def _single_map_packs(ctx, dicts, mono, origin0=0):
    """every map as one packed device buffer (lsfm_tree_export_dev of a one-map tree: nothing to join, the map is the result)"""
    import torch
    bufs = []
    for k, d in enumerate(dicts):
        t = ctx.tree_upload([dict(d, pose_origin=np.full(int(d["m"]), origin0 + k, np.int32))], mono)
        ctx.tree_set_final_reanchor(t, False)
        _, rc = ctx.tree_run(t)
        assert rc == 0
        n = ctx.tree_export_size(t)
        buf = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        ctx.tree_export_dev(t, buf.data_ptr(), n)
        ctx.tree_free(t)
        bufs.append(buf)
    torch.cuda.synchronize()
    return bufs


def test_reload_with_values_that_flip_a_planned_sign(ctx, oracle):
    """lsfm_tree_reload_dev keeps the plans of a resident tree.  A Mono plan holds the sign of every new scale
    (lmj_Transform_PF3DMono, Imp.cpp:3239-3244), which is a VALUE: here the same structure is reloaded with maps flown in
    the opposite direction (the second strip of an aerial block, relabelled onto the first), so every planned sign is
    wrong.  The run must notice on the device, drop the plans and repeat -- and agree with the oracle on the new values."""
    L, npf, vis = 12, 10, 4
    maps = synth.make_mono_set(2 * L - 2, npf, vis, seed=3, strip=L, spacing=10.0)  # spacing 10: no cross-strip features
    A = [oracle.localmap_to_dict(m) for m in maps[0:8]]
    B = []
    for m in maps[L:L + 8]:  # frames L .. of the second strip, labels shifted onto the first strip's
        d = oracle.localmap_to_dict(m)
        st = np.asarray(d["stno"]).copy()
        st[st <= 0] += L
        st[st > 0] -= L * npf
        d.update(stno=st, Ref=d["Ref"] - L, ScaP=d["ScaP"] - L, FRef=d["FRef"] - L, FScaP=d["FScaP"] - L)
        B.append(d)
    for a, b in zip(A, B):
        assert np.array_equal(a["stno"], b["stno"]) and a["Fix"] == b["Fix"] and a["Sign"] == -b["Sign"]
        for k in ("Ui", "Uj", "photo", "feature"):
            assert np.array_equal(a[k], b[k])
    pa, pb = _single_map_packs(ctx, A, True), _single_map_packs(ctx, B, True)
    t = ctx.tree_upload_dev([x.data_ptr() for x in pa], True)
    try:
        _, rc = ctx.tree_run(t)       # analyses, leaves plans
        assert rc == 0
        _, rc = ctx.tree_run(t)       # planned
        assert rc == 0
        got_a = ctx.tree_download(t)
        ctx.tree_reload_dev(t, [x.data_ptr() for x in pb])  # same labels and index arrays: the plans stay
        stats, rc = ctx.tree_run(t)
        assert rc == 0 and stats["not_converged"] == 0, stats
        got_b = ctx.tree_download(t)
        _, rc = ctx.tree_run(t)       # the plans of the repeated run fit the new values
        assert rc == 0
        got_b2 = ctx.tree_download(t)
    finally:
        ctx.tree_free(t)
    for got, dicts in ((got_a, A), (got_b, B), (got_b2, B)):
        exp, _, orc = oracle.divide_conquer(dicts, True)
        assert orc == 0
        assert np.array_equal(got["stno"], exp["stno"])
        assert got["Sign"] == exp["Sign"] and got["Fix"] == exp["Fix"]
        assert pose_param_err(got["stVal"], exp["stVal"], exp["stno"]) < 1e-8
        assert feat_param_err(got["stVal"], exp["stVal"], exp["stno"]) < 1e-8
    assert got_a["Sign"] == -got_b["Sign"]


def test_reload_with_another_structure_drops_the_plans(ctx, oracle):
    """Same sizes, other labels: the digest taken at upload and at reload differs, the plans go, the run analyses again."""
    maps = synth.make_stereo_set(8, 6, 5, seed=9)
    A = [oracle.localmap_to_dict(m) for m in maps]
    B = []
    for d in A:  # feature labels permuted inside every map pair-consistently: ids reversed over the whole set
        st = np.asarray(d["stno"]).copy()
        st[st > 0] = 10_000 - st[st > 0]
        B.append(dict(d, stno=st))
    pa, pb = _single_map_packs(ctx, A, False), _single_map_packs(ctx, B, False)
    t = ctx.tree_upload_dev([x.data_ptr() for x in pa], False)
    try:
        for _ in range(2):
            _, rc = ctx.tree_run(t)
            assert rc == 0
        ctx.tree_reload_dev(t, [x.data_ptr() for x in pb])
        _, rc = ctx.tree_run(t)
        assert rc == 0
        got = ctx.tree_download(t)
    finally:
        ctx.tree_free(t)
    exp, _, orc = oracle.divide_conquer(B, False)
    assert orc == 0
    assert np.array_equal(got["stno"], exp["stno"])
    assert pose_param_err(got["stVal"], exp["stVal"], exp["stno"]) < 1e-8


# ---------------------------------------------------------------------------------------------------------------------
# the early pattern of S (a Stereo level that analyses builds it from the level's inputs while the transform runs)
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("N,npf,vis,seed,lap", [(5, 6, 4, 7, 0), (33, 6, 5, 4, 0), (40, 4, 40, 8, 0), (300, 30, 5, 10, 50), (777, 12, 5, 11, 120)])
def test_early_schur_pattern_equals_the_joint_maps(ctx, oracle, monkeypatch, N, npf, vis, seed, lap):
    """LSFM_CHECK_EARLY_PATTERN=1 makes every analysing Stereo level build the pattern of S a second time, from the finished
    joint map (the way lsfm_schur_pattern and the stage-level calls do), and fail if the two differ in a single block.  Sets
    with an unpaired carry, re-anchored odd outputs, long tracks and loop closures; the result still matches the oracle."""
    maps = synth.make_stereo_set(N, new_per_frame=npf, vis=vis, seed=seed, lap=lap)
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    monkeypatch.setenv("LSFM_CHECK_EARLY_PATTERN", "1")
    got, stats, rc = ctx.divide_conquer(dicts, False)
    assert rc == 0, stats
    monkeypatch.delenv("LSFM_CHECK_EARLY_PATTERN")
    exp, _, orc = oracle.divide_conquer(dicts, False)
    assert orc == 0
    _same_structure(got, exp)
    assert pose_param_err(got["stVal"], exp["stVal"], exp["stno"]) < TREE_TOL


# ---------------------------------------------------------------------------------------------------------------------
# the factorisation gives the same bits whatever order its work-groups run in
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("config,n_maps", [("nc3500", 600), ("rs468", 200)])
def test_factorisation_is_bit_reproducible(config, n_maps):
    """LSFM_FACTOR_DIGEST=1 (read when the library loads: a process of its own): every camera system of every level is factored
    TWICE -- scatter, leaf sub-trees, supernode groups with their rank updates into the ancestors -- and the two factors (leaf columns,
    group columns, inverse diagonal blocks, the spent accumulators) are compared through an order-independent digest on the device.
    The updates that several work-groups of a launch add to one block are 64-bit fixed-point integers (lsfm_pcg.hip): whatever order
    the atomics land in, the sum is the same.  Round 3 added doubles there and two runs never gave the same factor.  Since round 5 the
    assembly of S and E (K9) is held to the same: `s_rebuild_mismatch`."""
    import subprocess
    import sys
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "from linearsfm_amd import api, synth\n"
        "typ, maps = synth.make_config(%r, %d)\n"
        "ctx = api.Context(0)\n"
        "t = ctx.tree_upload([m.__dict__ for m in maps], typ == 'Monocular')\n"
        "ctx.tree_set_plans(t, False)\n"
        "tot = tot_s = 0\n"
        "for _ in range(3):\n"
        "    st, rc = ctx.tree_run(t)\n"
        "    assert rc == 0 and st['factor_digest'] != 0\n"
        "    tot += st['refactor_mismatch']\n"
        "    tot_s += st['s_rebuild_mismatch']\n"
        "print('MISMATCH', tot, 'LEVELS', st['levels'], 'S_REBUILD_MISMATCH', tot_s)\n" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), config, n_maps))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, LSFM_FACTOR_DIGEST="1"), timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("MISMATCH")][0].split()
    assert int(line[1]) == 0 and int(line[3]) >= 8, line
    # round 5: the camera systems themselves -- S and E of every level ASSEMBLED twice from the same joint maps (U scatter, every panel
    # variant of K9, the per-feature fallback) -- are the same bits too: K9 adds in fixed point (k_schur_scale, lsfm_schur_panel.hip)
    assert int(line[5]) == 0, line


@pytest.mark.parametrize("n_maps", [24, 150])
def test_mono_pattern_from_the_level_below_equals_the_joint_maps(ctx, oracle, monkeypatch, n_maps):
    """Mono levels that analyse build the pattern of S from the pattern of the level below (through the join's pose renumbering) +
    the joint U + the pairs across the two sources of matched features (PatternSeed, lsfm_solve.hip).  LSFM_CHECK_MONO_SEED=1 makes
    every such level ALSO hash every pose pair of every feature of its joint maps, as before, and fail unless the two patterns are
    the same set -- a pair the seed lacked would lose its share of S silently.  The dense path is switched off so that every level
    has a pattern; the result is held to the oracle's."""
    maps = synth.make_mono_set(n_maps, 8, 4, seed=9, **synth.SPIRAL)
    dicts = [dict(m.__dict__) for m in maps]
    monkeypatch.setenv("LSFM_CHECK_MONO_SEED", "1")
    try:
        ctx.set_small_solve(0)
        got, stats, rc = ctx.divide_conquer(dicts, True)
    finally:
        ctx.set_small_solve(5)
    monkeypatch.delenv("LSFM_CHECK_MONO_SEED")
    assert rc == 0, stats
    exp, _, orc = oracle.divide_conquer(dicts, True)
    assert orc == 0
    _same_structure(got, exp)
    assert pose_param_err(got["stVal"], exp["stVal"], exp["stno"]) < TREE_TOL


def test_hinted_mono_refinement_asks_once_when_the_hint_is_short(ctx, oracle):
    """A run that analyses enqueues, per level, the refinement steps the run before needed.  A Mono level waits for the device at its end
    anyway, so when that count turns out short it asks ONCE and goes on step by step (lsfm_pcg.hip, `ask_after`) -- until round 6 the whole
    run was joined again.  Forced here: the first runs leave the step counts of the fp64 preconditioner (one step per level), then the
    preconditioner is switched to fp32, which needs two -- the next run's hints are short at every level above the dense path's, it
    must still converge in ONE attempt, and its map must be the oracle's."""
    typ, maps = synth.make_config("rs468", 200)
    assert typ == "Monocular"
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    c = ctx
    t = None
    try:
        t = c.tree_upload(dicts, True)
        c.tree_set_plans(t, False)
        for _ in range(2):
            s0, rc = c.tree_run(t)
            assert rc == 0 and s0["attempts"] == 1
        c.set_precision(True)
        s1, rc = c.tree_run(t)
        assert rc == 0 and s1["attempts"] == 1 and s1["not_converged"] == 0, s1
        assert s1["pcg_iterations"] > s0["pcg_iterations"], (s0["pcg_iterations"], s1["pcg_iterations"])
        assert s1["max_rel_residual"] < 1e-9, s1
        got = c.tree_download(t)
    finally:
        c.set_precision(False)  # (the session's context: back to the library's default)
        if t is not None:
            c.tree_free(t)
    exp, _, orc = oracle.divide_conquer(dicts, True, match_hash=True)
    assert orc == 0
    ep = pose_param_err(got["stVal"], exp["stVal"], exp["stno"])
    print(f"steps per tree {s0['pcg_iterations']} (fp64 preconditioner) -> {s1['pcg_iterations']} (fp32, hints of the fp64 runs), one attempt; pose parameters vs oracle {ep:.2e}")
    assert ep < TREE_TOL
