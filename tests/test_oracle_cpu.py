"""CPU-only tests: the oracle against the REAL reference's outputs stored in tests/golden/ (transform + join assembly at
every tree level, made by tests/golden/make_golden.py from oracle/_ref/ref_dump), algebraic self-checks of the oracle's
solve (the one stage the reference cannot pin here: CHOLMOD is absent), the synthetic generator, file formats."""
import os
import subprocess

import numpy as np
import pytest

from common import CHAIN_FLOOR, GOLD_CHAIN, GOLD_MID, GOLD_SMALL, GOLD_WIDE, assert_maps_close, dense_reference_solve, feat_param_err, get_map, golden_system, load_golden, pose_param_err, ref_map, rel_err
from linearsfm_amd import synth
from refdump import dense_info

GOLD = GOLD_SMALL


@pytest.mark.parametrize("name", GOLD + GOLD_MID + GOLD_WIDE)
def test_oracle_transform_and_assembly_vs_reference(oracle, name):
    z = load_golden(name)
    mono = str(z["type"]) == "Monocular"
    for j in range(int(z["njoins"])):
        A, B = get_map(z, f"join{j}.A"), get_map(z, f"join{j}.B")
        E = oracle.transform(A, mono, B["Ref"], B["ScaP"], B["Fix"])
        exp = ref_map(z, f"join{j}.end")
        exp["FRef"] = A["FRef"]
        assert_maps_close(E, exp, 1e-12, f"{name} join{j} transform")
        J, eP, eF, sa, _, _ = oracle.join_assemble(E, B, mono)
        for k in ("Ui", "Uj", "photo", "feature"):
            assert np.array_equal(J[k], z[f"join{j}.solve.{k}"]), (j, k)
        assert np.array_equal(J["stno"], z[f"join{j}.joint.stno"])
        assert np.array_equal(J["FBlock"], z[f"join{j}.joint.FBlock"])
        for k, x in (("U", J["U"]), ("W", J["W"]), ("V", J["V"]), ("ea", eP), ("eb", eF)):
            assert rel_err(x, z[f"join{j}.solve.{k}"]) < 1e-12, (j, k)
        if mono:
            ref_sa = [int(z[f"join{j}.solve.{k}"][0]) for k in ("Ref", "ScaP", "Fix", "Sign", "FixBlk")]
            assert sa == ref_sa
        # the oracle's own solution of this system is what the fixture carries to the next level
        st, rc, _ = oracle.solve(J, eP, eF, mono, sa)
        assert rc == 0
        assert rel_err(st, z[f"join{j}.sol"]) < 1e-9


@pytest.mark.parametrize("name", GOLD)
def test_oracle_solve_stage_vs_reference_methods_and_dense_lapack(oracle, name):
    """The solve stage, piece by piece, on every system the REAL reference assembled (22 joins, Stereo and Mono):
    V^-1, back-substitution and the CSC handed to CHOLMOD against the reference's own pba_inverseV / pba_solveFeatures /
    pba_constructCSS{LM,GN} / pba_constructAuxCSS{LM,GN} (fixture parts.*, run by oracle/_ref/ref_dump); the Schur
    complement against a dense numpy evaluation of U - W V^-1 W^T; the solution against the dense LAPACK expected value
    of the full normal equations (dense_sol: no Schur complement, no sparse factorisation)."""
    z = load_golden(name)
    for j in range(int(z["njoins"])):
        J, ea, eb, mono, sa = golden_system(z, j)
        m, n = J["m"], J["n"]
        rowptr, colidx, S, E, IV = oracle.schur(J, ea, eb, 1 if mono else 0)
        # pba_inverseV
        assert rel_err(IV, z[f"join{j}.parts.IV"]) < 1e-13
        # pattern: pba_constructAuxCSS* list the upper blocks column by column
        assert np.array_equal(rowptr, z[f"join{j}.parts_in.rowptr"]) and np.array_equal(colidx, z[f"join{j}.parts_in.colidx"])
        cols = [[] for _ in range(m)]
        for p in range(m):
            for k in range(rowptr[p], rowptr[p + 1]):
                cols[colidx[k]].append(p)
        if mono:  # block `Ref` dropped, later blocks renumbered -1 (Imp.cpp:7248-7280)
            ref = sa[0]
            cols = [[r - (r > ref) for r in c if r != ref] for q, c in enumerate(cols) if q != ref]
        Ap = np.cumsum([0] + [len(c) for c in cols]).astype(np.int32)
        assert np.array_equal(Ap, z[f"join{j}.parts.Ap"])
        assert np.array_equal(np.concatenate(cols).astype(np.int32), z[f"join{j}.parts.Aii"])
        # Schur complement, dense and independent: S = U - sum_f W_f IV_f W_f^T on the upper blocks, E = ea - W IV eb
        A = np.zeros((6 * m, 6 * m)); Ed = np.array(ea, np.float64)
        U = J["U"].reshape(-1, 6, 6)
        for k in range(len(J["Ui"])):
            a, b = int(J["Ui"][k]), int(J["Uj"][k])
            A[6 * a:6 * a + 6, 6 * b:6 * b + 6] += U[k]
        W = J["W"].reshape(-1, 6, 3); IVr = z[f"join{j}.parts.IV"].reshape(-1, 3, 3)
        ph, fe = J["photo"], J["feature"]
        for x in range(len(ph)):
            WV = W[x] @ IVr[fe[x]]
            Ed[6 * ph[x]:6 * ph[x] + 6] -= WV @ eb[3 * fe[x]:3 * fe[x] + 3]
            for y in np.nonzero(fe == fe[x])[0]:
                if ph[x] <= ph[y]:
                    A[6 * ph[x]:6 * ph[x] + 6, 6 * ph[y]:6 * ph[y] + 6] -= WV @ W[y].T
        scale = np.abs(A).max()
        for p in range(m):
            for k in range(rowptr[p], rowptr[p + 1]):
                q = colidx[k]
                got, exp = S[k], A[6 * p:6 * p + 6, 6 * q:6 * q + 6]
                if p == q:
                    got, exp = np.triu(got), np.triu(exp)  # diagonal blocks: upper triangle only (Imp.cpp:2224-2229, 2307)
                assert np.abs(got - exp).max() / scale < 1e-13, (j, p, q)
        assert rel_err(E, Ed) < 1e-12
        # pba_constructCSSLM / GN: the scalar CSC handed to cholmod_factorize, a copy of S's entries
        Sp, Si, Sx = oracle.schur_csc(z[f"join{j}.parts_in.S"], rowptr, colidx, m, sa[0] if mono else -1, sa[2] if mono else -1)
        assert np.array_equal(Sp, z[f"join{j}.parts.Sp"]) and np.array_equal(Si, z[f"join{j}.parts.Si"])
        assert np.array_equal(Sx, z[f"join{j}.parts.Sx"])
        # pba_solveFeatures for the pose values the fixture handed to the reference
        dpb = oracle.solve_features(J, z[f"join{j}.parts.IV"], eb, z[f"join{j}.parts_in.dpa"])
        assert rel_err(dpb, z[f"join{j}.parts.dpb"]) < 1e-13
        # the whole solve against the dense LAPACK expected value, recomputed here and as stored
        xd = z[f"join{j}.dense_sol"]
        live = dense_reference_solve(J, ea, eb, mono, sa, IV=z[f"join{j}.parts.IV"])
        assert np.max(np.abs(live - xd) / np.maximum(1, np.abs(xd))) < 1e-14
        st, rc, _ = oracle.solve(J, ea, eb, mono, sa)
        assert rc == 0
        assert np.max(np.abs(st - xd) / np.maximum(1, np.abs(xd))) < 1e-10
        stx, rc, _ = oracle.solve(J, ea, eb, mono, sa, extended=True)
        assert rc == 0
        assert np.max(np.abs(stx - xd) / np.maximum(1, np.abs(xd))) < 1e-10


@pytest.mark.parametrize("name", GOLD_MID + GOLD_WIDE)
def test_oracle_solve_stage_vs_reference_at_mid_size(oracle, name):
    """The top joins of the 64-map Stereo and the 88-map Mono tree (m = 64 / 66 / 90 poses), as the REAL reference assembled
    them: the oracle's V^-1 and back-substitution against pba_inverseV / pba_solveFeatures, its block pattern of S against
    pba_constructAuxCSS{LM,GN}, its solution against the dense LAPACK expected value (stored and recomputed).  The scalar CSC of S
    is not stored at this size (the small fixtures pin pba_constructCSS*)."""
    z = load_golden(name)
    for j in range(int(z["njoins"])):
        J, ea, eb, mono, sa = golden_system(z, j)
        m = J["m"]
        rowptr, colidx, S, E, IV = oracle.schur(J, ea, eb, 1 if mono else 0)
        assert rel_err(IV, z[f"join{j}.parts.IV"]) < 1e-13
        assert np.array_equal(rowptr, z[f"join{j}.parts_in.rowptr"]) and np.array_equal(colidx, z[f"join{j}.parts_in.colidx"])
        cols = [[] for _ in range(m)]
        for p in range(m):
            for k in range(rowptr[p], rowptr[p + 1]):
                cols[colidx[k]].append(p)
        if mono:
            ref = sa[0]
            cols = [[r - (r > ref) for r in c if r != ref] for q, c in enumerate(cols) if q != ref]
        assert np.array_equal(np.cumsum([0] + [len(c) for c in cols]).astype(np.int32), z[f"join{j}.parts.Ap"])
        assert np.array_equal(np.concatenate(cols).astype(np.int32), z[f"join{j}.parts.Aii"])
        dpb = oracle.solve_features(J, z[f"join{j}.parts.IV"], eb, z[f"join{j}.parts_in.dpa"])
        assert rel_err(dpb, z[f"join{j}.parts.dpb"]) < 1e-12
        xd = z[f"join{j}.dense_sol"]
        live = dense_reference_solve(J, ea, eb, mono, sa, IV=z[f"join{j}.parts.IV"])
        assert np.max(np.abs(live - xd) / np.maximum(1, np.abs(xd))) < 1e-13
        # (a monocular system of 90 poses is conditioned ~1e4 x worse than the 8-pose ones: 1e-9 here, 1e-10 there)
        st, rc, _ = oracle.solve(J, ea, eb, mono, sa)
        assert rc == 0
        assert np.max(np.abs(st - xd) / np.maximum(1, np.abs(xd))) < 1e-9


@pytest.mark.parametrize("name", GOLD)
def test_transform_is_congruence(oracle, name):
    """I' = J^T I J is symmetric positive semi-definite with the same inertia; checked densely (independent of the
    block bookkeeping): x^T I x of the old state perturbation equals x'^T I' x' to first order is implied by
    symmetry + unchanged rank."""
    z = load_golden(name)
    mono = str(z["type"]) == "Monocular"
    A, B = get_map(z, "join0.A"), get_map(z, "join0.B")
    E = oracle.transform(A, mono, B["Ref"], B["ScaP"], B["Fix"])
    I0, I1 = dense_info(A), dense_info(E)
    assert np.allclose(I1, I1.T, rtol=0, atol=1e-9 * np.abs(I1).max())
    w1 = np.linalg.eigvalsh(I1)
    assert w1.min() > -1e-9 * w1.max()
    if not mono:
        assert np.linalg.matrix_rank(I0, tol=1e-9 * np.abs(I0).max()) == np.linalg.matrix_rank(I1, tol=1e-9 * np.abs(I1).max())


def test_solve_satisfies_normal_equations(oracle):
    """The joint estimate solves (I1 + I2) x = I1 x1 + I2 x2 -- checked on the dense full system, which
    does not go through the Schur complement or the sparse Cholesky at all."""
    z = load_golden("stereo_n8.npz")
    j = int(z["njoins"]) - 1
    J = dict(m=int(z[f"join{j}.solve.m"][0]), n=int(z[f"join{j}.solve.n"][0]), U=z[f"join{j}.solve.U"],
             W=z[f"join{j}.solve.W"], V=z[f"join{j}.solve.V"], Ui=z[f"join{j}.solve.Ui"], Uj=z[f"join{j}.solve.Uj"],
             photo=z[f"join{j}.solve.photo"], feature=z[f"join{j}.solve.feature"])
    ea, eb = z[f"join{j}.solve.ea"], z[f"join{j}.solve.eb"]
    st, rc, stats = oracle.solve(J, ea, eb, False)
    assert rc == 0
    I = dense_info(J)
    b = np.concatenate([ea, eb])
    assert np.linalg.norm(I @ st - b) / np.linalg.norm(b) < 1e-10
    xd = np.linalg.solve(I, b)
    assert np.max(np.abs(st - xd) / np.maximum(1, np.abs(xd))) < 1e-8


@pytest.mark.parametrize("mono", [False, True])
def test_tree_reproduces_ground_truth_geometry(oracle, mono):
    """End-to-end sanity of generator + oracle: the joined trajectory follows the synthetic camera path."""
    N = 12
    maps = synth.make_mono_set(N, 12, 4, seed=3) if mono else synth.make_stereo_set(N, 8, 4, seed=3)
    out, timing, rc = oracle.divide_conquer([oracle.localmap_to_dict(m) for m in maps], mono)
    assert rc == 0
    stno, st = out["stno"], out["stVal"]
    ids = -stno[stno <= 0][::6]
    pos = st[: 6 * out["m"]].reshape(-1, 6)[:, :3]
    o = np.argsort(ids)
    steps = np.linalg.norm(np.diff(pos[o], axis=0), axis=1)
    if mono:
        steps = steps / steps[0]
        assert np.all(np.abs(steps - 1.0) < 0.2)
    else:
        assert np.all(np.abs(steps - 0.5) < 0.05)
    assert out["Ref"] == out["FRef"] == maps[0].Ref


def test_match_hash_equals_linear_find(oracle):
    maps = synth.make_stereo_set(9, 6, 5, seed=9)
    d = [oracle.localmap_to_dict(m) for m in maps]
    a, _, _ = oracle.divide_conquer(d, False, match_hash=False)
    b, _, _ = oracle.divide_conquer(d, False, match_hash=True)
    assert np.array_equal(a["stno"], b["stno"]) and np.array_equal(a["stVal"], b["stVal"])


def test_localmap_text_roundtrip(oracle, tmp_path):
    """Generator -> reference text format -> oracle reader / product reader give identical arrays."""
    from linearsfm_amd import api
    for mono, maps in ((False, synth.make_stereo_set(2, 4, 4, seed=1)), (True, synth.make_mono_set(2, 5, 4, seed=1))):
        p = str(tmp_path / f"localmap_{int(mono)}.txt")
        synth.write_localmap(p, maps[0])
        g = oracle.map_to_dict(oracle.read_map(p, mono))
        back = synth.read_localmap(p, mono)
        h = api.read_localmap(p, mono)
        for k in ("stno", "Ui", "Uj", "photo", "feature", "FBlock"):
            assert np.array_equal(g[k], getattr(maps[0], k)) and np.array_equal(getattr(back, k), getattr(maps[0], k))
            assert np.array_equal(h[k], getattr(maps[0], k))
        for k in ("stVal", "U", "W", "V"):
            assert np.array_equal(np.asarray(g[k]).ravel(), np.asarray(getattr(maps[0], k)).ravel())
            assert np.array_equal(np.asarray(h[k]).ravel(), np.asarray(getattr(maps[0], k)).ravel())
        assert g["Ref"] == maps[0].Ref == h["Ref"]
        if mono:
            assert (g["ScaP"], g["Fix"], g["Sign"]) == (maps[0].ScaP, maps[0].Fix, maps[0].Sign) == (h["ScaP"], h["Fix"], h["Sign"])


def test_oracle_cli_and_output_formats(oracle, tmp_path):
    """The oracle CLI takes the reference's flags and writes the reference's '%lf' files sorted by id."""
    maps = synth.make_stereo_set(4, 4, 4, seed=2)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(os.path.dirname(oracle.__file__), "lsfm_oracle")
    p, f, s = tmp_path / "Pose.txt", tmp_path / "Feature.txt", tmp_path / "State.txt"
    out = subprocess.run([exe, "-path", str(d), "-num", "4", "-type", "Stereo", "-p", str(p), "-f", str(f), "-st", str(s)],
                         capture_output=True, text=True, check=True).stdout
    assert "Join Level 0 Local Map 1" in out and "Generate Level 2 Local Map 1" in out and "Total Used Time:" in out
    poses = [l.split() for l in open(p)]
    assert [int(r[0]) for r in poses] == sorted(int(r[0]) for r in poses) and all(len(r) == 7 for r in poses)
    assert all(len(l.split()) == 4 for l in open(f))
    assert all(len(l.split()) == 2 for l in open(s))
    assert len(poses[0][1].split(".")[1]) == 6  # %lf


def test_reader_number_forms_are_exact(tmp_path):
    """The product reader tokenises by hand; every number form scanf("%lf") accepts must give the same bits as the C
    library (= Python float): short decimals (fast path), 17-digit decimals, exponents, signs, inf/nan, leading zeros."""
    from linearsfm_amd import api
    toks = ["0.5", "-0.000000", "+1.5", "12345.678901", "123456789.123456", "0.1", "1e-3", "1E5", "-2.5e+10", "007.250",
            "0.12345678901234567", "9007199254740993", "1234567890123456789012", "1e23", "4.9e-324", "1.7976931348623157e308",
            "inf", "-inf", "3", "-7", ".5", "5.", "1e-400", "0x1.8p1"]
    vals = [float.fromhex(t) if t.startswith("0x") else float(t) for t in toks]
    r = 6 + 3 * ((len(toks) - 6 + 2) // 3)
    vals_p = vals + [0.0] * (r - len(vals))
    toks_p = toks + ["0"] * (r - len(toks))
    n = (r - 6) // 3
    p = tmp_path / "localmap_1.txt"
    with open(p, "w") as f:
        f.write("3\n%d\n" % r)
        for i, t in enumerate(toks_p):
            f.write("%d   %s\n" % (-4 if i < 6 else 1 + (i - 6) // 3, t))
        f.write("1 %d 1\n" % n + " ".join(["1.25"] * 36) + "\n0\n0\n0\n\n\n" + " ".join(["2"] * (9 * n)) + "\n" + " ".join(["-1"] * n) + "\n")
    g = api.read_localmap(str(p), False)
    assert g["m"] == 1 and g["n"] == n and g["nU"] == 1 and g["nW"] == 0
    got = np.asarray(g["stVal"])
    exp = np.asarray(vals_p)
    assert np.array_equal(got.view(np.uint64), exp.view(np.uint64)), [(t, a, b) for t, a, b in zip(toks_p, got, exp) if a != b]
    assert np.all(np.asarray(g["U"]) == 1.25) and np.all(np.asarray(g["V"]) == 2.0)


@pytest.mark.parametrize("mono", [False, True])
def test_parallel_set_reader_equals_reference_reader(oracle, tmp_path, mono):
    """lsfm_read_localmaps (threads) == one lsfm_read_localmap per file == the fscanf port of Imp.cpp:3044-3132 / 6660-6754."""
    from linearsfm_amd import api
    maps = synth.make_mono_set(7, 6, 4, seed=3) if mono else synth.make_stereo_set(7, 6, 4, seed=3)
    synth.write_set(str(tmp_path), maps)
    par = api.read_localmaps(str(tmp_path), 7, mono, threads=3)
    for k in range(7):
        fn = str(tmp_path / f"localmap_{k + 1}.txt")
        ref = oracle.map_to_dict(oracle.read_map(fn, mono))
        one = api.read_localmap(fn, mono)
        for key in ("stno", "stVal", "U", "Ui", "Uj", "W", "photo", "feature", "V", "FBlock"):
            a, b, c = (np.asarray(x[key]).ravel() for x in (ref, one, par[k]))
            assert np.array_equal(a, b) and np.array_equal(b, c), (k, key)
        assert ref["Ref"] == one["Ref"] == par[k]["Ref"]
    with pytest.raises(api.LsfmError, match="localmap_8"):
        api.read_localmaps(str(tmp_path), 9, mono, threads=2)


@pytest.mark.parametrize("mono", [False, True])
def test_localmap_writer_roundtrip(oracle, tmp_path, mono):
    """lsfm_write_localmap -> lsfm_read_localmap is the identity (a joined map with its information matrix: the oracle's
    tree output), and the file is readable by the fscanf port of the reference's reader too."""
    from linearsfm_amd import api
    maps = synth.make_mono_set(5, 6, 4, seed=4) if mono else synth.make_stereo_set(5, 6, 4, seed=4)
    out, _, rc = oracle.divide_conquer([oracle.localmap_to_dict(m) for m in maps], mono)
    assert rc == 0
    p = str(tmp_path / "final_map.txt")
    api.write_localmap(p, out, mono)
    back = api.read_localmap(p, mono)
    ref = oracle.map_to_dict(oracle.read_map(p, mono))
    for key in ("stno", "stVal", "U", "Ui", "Uj", "W", "photo", "feature", "V"):
        a, b, c = (np.asarray(x[key]).ravel() for x in (out, back, ref))
        assert np.array_equal(a, b) and np.array_equal(a, c), key
    assert back["Ref"] == out["Ref"] and back["m"] == out["m"] and back["n"] == out["n"]
    if mono:
        assert (back["ScaP"], back["Fix"], back["Sign"]) == (out["ScaP"], out["Fix"], out["Sign"])


@pytest.mark.parametrize("mono", [False, True])
def test_tree_node_trailer_roundtrip_and_invisible_to_the_reference_reader(oracle, tmp_path, mono):
    """A node of a join tree written for a later resume carries FRef / FScaP / FFix and the origins of its poses behind FBlock:
    lsfm_read_localmap restores them, the fscanf port of the reference's reader (Imp.cpp:3044-3132 / 6660-6754) reads the same map
    and never reaches them, a file without the trailer reads as a local map (first frame = reference frame), a damaged one fails."""
    from linearsfm_amd import api
    maps = synth.make_mono_set(5, 6, 4, seed=4) if mono else synth.make_stereo_set(5, 6, 4, seed=4)
    out, _, rc = oracle.divide_conquer([oracle.localmap_to_dict(m) for m in maps], mono)
    assert rc == 0
    node = dict(out)
    node["FRef"] = out["stno"][0] * 0 + 1
    node["Ref"] = 3
    node["pose_origin"] = (np.arange(out["m"]) % 5).astype(np.int32)
    if mono:
        node["FScaP"], node["FFix"] = 2, 1
    p = str(tmp_path / "node.txt")
    api.write_localmap(p, node, mono)
    back = api.read_localmap(p, mono)
    assert back["Ref"] == 3 and back["FRef"] == 1 and np.array_equal(back["pose_origin"], node["pose_origin"])
    if mono:
        assert (back["FScaP"], back["FFix"]) == (2, 1) and (back["ScaP"], back["Fix"]) == (node["ScaP"], node["Fix"])
    ref = oracle.map_to_dict(oracle.read_map(p, mono))
    for key in ("stno", "stVal", "U", "Ui", "Uj", "W", "photo", "feature", "V"):
        assert np.array_equal(np.asarray(ref[key]).ravel(), np.asarray(back[key]).ravel()), key
    # no trailer: a local map
    plain = dict(out); plain.pop("pose_origin", None); plain["FRef"] = plain["Ref"]
    if mono:
        plain["FScaP"], plain["FFix"] = plain["ScaP"], plain["Fix"]
    q = str(tmp_path / "plain.txt")
    api.write_localmap(q, plain, mono)
    assert str(api.LSFM_NODE_MAGIC) not in open(q).read().split()
    b2 = api.read_localmap(q, mono)
    assert b2["FRef"] == b2["Ref"] and "pose_origin" not in b2
    # a trailer cut short is an error, not a silently different map
    text = open(p).read().split()
    cut = str(tmp_path / "cut.txt")
    open(cut, "w").write(" ".join(text[:-2]) + "\n")
    with pytest.raises(api.LsfmError):
        api.read_localmap(cut, mono)


@pytest.mark.parametrize("mono", [False, True])
def test_binary_mapset_cache_equals_the_text_reader(oracle, tmp_path, mono):
    """the binary cache of a set holds what the text reader (== the fscanf port of Imp.cpp:3044-3132 / 6660-6754, test
    above) produced, bit for bit, for whole sets and sub-ranges; a truncated file, a foreign file, the other map type and a range
    past the end are errors that leave nothing behind."""
    from linearsfm_amd import api
    maps = synth.make_mono_set(9, 6, 4, seed=6) if mono else synth.make_stereo_set(9, 6, 4, seed=6)
    synth.write_set(str(tmp_path), maps)
    text = api.read_localmaps(str(tmp_path), 9, mono, threads=2)
    cache = str(tmp_path / "set.lsfmbin")
    assert api.mapset_info(cache) is None
    api.write_mapset(cache, text, mono)
    assert api.mapset_info(cache) == (9, mono)
    keys = ("stno", "stVal", "U", "Ui", "Uj", "W", "photo", "feature", "V", "FBlock")
    scal = ("Ref", "FRef", "m", "n", "nU", "nW") + (("ScaP", "Fix", "Sign", "FScaP", "FFix") if mono else ())
    for first, count, threads in ((0, 9, 3), (2, 4, 1), (8, 1, 0), (3, 0, 0)):
        got = api.read_mapset(cache, mono, first, count, threads)
        assert len(got) == count
        for k in range(count):
            for key in keys:
                a, b = np.asarray(text[first + k][key]), np.asarray(got[k][key])
                assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), (first, k, key)
            for key in scal:
                assert text[first + k][key] == got[k][key], key
            assert "pose_origin" not in got[k]
    # a tree node (pose origins, another first frame) survives too
    node = dict(text[0]); node["FRef"] = 1; node["Ref"] = 2; node["pose_origin"] = np.arange(node["m"], dtype=np.int32)
    api.write_mapset(str(tmp_path / "node.lsfmbin"), [node], mono)
    back = api.read_mapset(str(tmp_path / "node.lsfmbin"), mono)[0]
    assert back["FRef"] == 1 and back["Ref"] == 2 and np.array_equal(back["pose_origin"], node["pose_origin"])
    # errors
    raw = open(cache, "rb").read()
    for name, data in (("cut", raw[:-16]), ("short", raw[:40]), ("foreign", b"not a cache at all" * 8), ("grown", raw + b"\0" * 8)):
        q = str(tmp_path / f"{name}.lsfmbin")
        open(q, "wb").write(data)
        with pytest.raises(api.LsfmError):
            api.read_mapset(q, mono)
    with pytest.raises(api.LsfmError):
        api.read_mapset(cache, not mono)
    with pytest.raises(api.LsfmError):
        api.read_mapset(cache, mono, 5, 5)
    assert not os.path.exists(cache + ".tmp")


def test_binary_state_dump(tmp_path):
    from linearsfm_amd import api
    import ctypes as C
    for n in (0, 5, 8):
        stno = np.arange(n, dtype=np.int32) - 3
        st = np.linspace(-1e3, 1e-7, n) if n else np.zeros(0)
        p = str(tmp_path / f"s{n}.bin")
        rc = api.lib().lsfm_save_state_bin(p.encode(), st.ctypes.data_as(C.POINTER(C.c_double)), stno.ctypes.data_as(C.POINTER(C.c_int)), n)
        assert rc == 0
        raw = open(p, "rb").read()
        assert np.frombuffer(raw[:8], np.int32).tolist() == [n, 0]
        assert np.array_equal(np.frombuffer(raw[8:8 + 4 * n], np.int32), stno)
        assert np.frombuffer(raw[8 + 4 * (n + (n & 1)):], np.float64).tobytes() == st.tobytes()


def _writer_cases():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "writers.npz"))
    return z, [str(n) for n in z["names"]]


@pytest.mark.parametrize("name", _writer_cases()[1])
def test_writers_byte_for_byte_vs_reference(tmp_path, name):
    """lsfm_save_state / lsfm_save_poses (csrc/lsfm_io.cpp) against the BYTES the reference's own lmj_SaveStateVector
    (LinearSFMImp.cpp:2102-2117) and lmj_SavePoses_3DPF (7876-7967) wrote for the same state (tests/golden/writers.npz, made by
    tests/golden/make_writer_golden.py running the real methods through oracle/_ref/ref_dump save): ids out of order, repeated
    ids (the last occurrence is kept), label 0, negative zeros, sixth-decimal rounding, 1e300, an empty state; both files, only
    the pose file, only the feature file (the method takes NULL for either path)."""
    import ctypes as C
    from linearsfm_amd import api
    z, _ = _writer_cases()
    stno, st = np.ascontiguousarray(z[f"{name}.stno"], np.int32), np.ascontiguousarray(z[f"{name}.stVal"], np.float64)
    ip, dp = stno.ctypes.data_as(C.POINTER(C.c_int)), st.ctypes.data_as(C.POINTER(C.c_double))
    L = api.lib()
    fs = str(tmp_path / "state.txt")
    assert L.lsfm_save_state(fs.encode(), dp, ip, len(stno)) == 0
    assert open(fs, "rb").read() == bytes(z[f"{name}.state"])
    for which in ("both", "pose", "feat"):
        fp, ff = str(tmp_path / f"{which}_pose.txt"), str(tmp_path / f"{which}_feat.txt")
        assert L.lsfm_save_poses(fp.encode() if which != "feat" else None, ff.encode() if which != "pose" else None, ip, dp, len(stno)) == 0
        assert os.path.exists(fp) == (which != "feat") and os.path.exists(ff) == (which != "pose")
        if which != "feat":
            assert open(fp, "rb").read() == bytes(z[f"{name}.{which}.pose"]), which
        if which != "pose":
            assert open(ff, "rb").read() == bytes(z[f"{name}.{which}.feat"]), which


@pytest.mark.parametrize("name", _writer_cases()[1])
def test_python_statement_of_the_writers_vs_reference_bytes(name):
    """tests/common.py reference_writer_bytes (used by the -m gpu CLI test for the device's own state) against the real writers' bytes."""
    from common import reference_writer_bytes
    z, _ = _writer_cases()
    state, pose, feat = reference_writer_bytes(z[f"{name}.stno"], z[f"{name}.stVal"])
    assert state == bytes(z[f"{name}.state"]) and pose == bytes(z[f"{name}.both.pose"]) and feat == bytes(z[f"{name}.both.feat"])


@pytest.mark.parametrize("mono", [False, True])
def test_threaded_oracle_tree_is_identical(oracle, mono):
    """orc_divide_conquer_omp (the multi-core CPU figure of bench.py) computes every join exactly like the serial tree."""
    maps = synth.make_mono_set(11, 6, 4, seed=8) if mono else synth.make_stereo_set(13, 6, 5, seed=8)
    d = [oracle.localmap_to_dict(m) for m in maps]
    a, _, rc1 = oracle.divide_conquer(d, mono)
    b, _, rc2 = oracle.divide_conquer(d, mono, threads=3)
    assert rc1 == 0 and rc2 == 0
    for k in ("stno", "stVal", "U", "Ui", "Uj", "W", "photo", "feature", "V"):
        assert np.array_equal(np.asarray(a[k]), np.asarray(b[k])), k


@pytest.mark.parametrize("name", GOLD)
def test_oracle_whole_tree_from_fixture_inputs_to_fixture_result(oracle, name):
    """orc_divide_conquer (the oracle's restatement of lmj_PF3D_Divide_Conquer*'s loop, Imp.cpp:1932-2063 / 6517-6630) on
    the fixture's input maps against the fixture's final map: that one was produced join by join by make_golden.py, which
    follows the reference's loop itself -- pairing, the unpaired carry (`NumLM == 1`, Imp.cpp:1940-1948), re-anchoring of the
    odd outputs ((i + 1) % 2 == 0, Imp.cpp:1997) and of the final map (2039) -- and checks every transform and assembly on
    the way against the real reference.  Same arithmetic in the same order: bit for bit."""
    z = load_golden(name)
    mono = str(z["type"]) == "Monocular"
    maps = [get_map(z, f"in{k}") for k in range(int(z["N"]))]
    exp = get_map(z, "result")
    got, _, rc = oracle.divide_conquer(maps, mono, match_hash=False)
    assert rc == 0
    for k in ("Ref", "FRef", "m", "n"):
        assert int(got[k]) == int(exp[k]), k
    for k in ("stno", "Ui", "Uj", "photo", "feature", "FBlock"):
        assert np.array_equal(np.asarray(got[k]).ravel(), np.asarray(exp[k]).ravel()), k
    for k in ("stVal", "U", "W", "V"):
        assert np.array_equal(np.asarray(got[k]).ravel(), np.asarray(exp[k]).ravel()), k
    if mono:
        for k in ("ScaP", "Fix", "Sign"):
            assert int(got[k]) == int(exp[k]), k


@pytest.mark.parametrize("name", GOLD_SMALL + GOLD_MID)
def test_schur_reference_solve_is_the_dense_expected_value(name):
    """tests/common.py schur_reference_solve (what the reference-chain fixtures' solves are: long-double residuals of the FULL system, a
    dense LAPACK factor of the Schur complement as preconditioner) against the fixtures' dense_sol -- the dense LU of the full normal
    equations: the same exact solution, reached another way.  (Mono: dense_sol takes the reference's computed V^-1 as exact, this one V:
    they differ by cond(V) x 1e-16, which a monocular camera system amplifies to 1e-11..1e-10.)"""
    from common import schur_reference_solve
    z = load_golden(name)
    for j in range(int(z["njoins"])):
        J, ea, eb, mono, sa = golden_system(z, j)
        x = schur_reference_solve(J, ea, eb, mono, sa)
        d = z[f"join{j}.dense_sol"]
        assert np.max(np.abs(x - d) / np.maximum(1, np.abs(d))) < (1e-9 if mono else 1e-12), (name, j)


@pytest.mark.parametrize("name", GOLD_CHAIN)
def test_oracle_whole_tree_vs_reference_chain(oracle, name):
    """The oracle's WHOLE tree -- transforms, assemblies, Schur complements, sparse Cholesky, back-substitutions, 9 / 8 levels, systems of up
    to 512 / 202 / 2 048 poses with lap closures -- against the same tree evaluated by the REAL reference (every transform and assembly:
    oracle/_ref/ref_dump) with every solve replaced by the exact solution of the reference-assembled system (make_chain_golden.py): the
    chain holds no arithmetic of the oracle.  Measured 6.8e-10 (512 Stereo) / 3.4e-8 (200 Mono) / 2.5e-8 (2 048 Stereo) on the pose parameters; the long-double twin of the
    oracle's solves lands at the same distance -- what is left is the rounding of the state between the levels, not a solver's.  On the
    768-map Mono chain that rounding alone is 1e-6 .. 3e-6 (common.CHAIN_FLOOR): the fixture shows where BASELINE.json's 1e-6 meets fp64."""
    from common import chain_bar, chain_set
    typ, mono, maps, z = chain_set(name)
    d = [oracle.localmap_to_dict(m) for m in maps]
    G, _, rc = oracle.divide_conquer(d, mono, match_hash=True)
    assert rc == 0
    assert np.array_equal(G["stno"], z["result.stno"])
    for k in ("Ref", "FRef") + (("ScaP", "Fix", "Sign") if mono else ()):
        assert int(G[k]) == int(z[f"result.{k}"]), k
    ep, ef = pose_param_err(G["stVal"], z["result.stVal"], z["result.stno"]), feat_param_err(G["stVal"], z["result.stVal"], z["result.stno"])
    bar = chain_bar(name, 1e-7)
    print(f"{name}: oracle vs the reference chain: pose parameters {ep:.2e}, features {ef:.2e} (bar {bar:.1e})")
    assert ep < bar and ef < bar, (ep, ef)
    if name in CHAIN_FLOOR:
        # (the spread really is fp64's, not the oracle's: its long-double twin is as far from the chain, and from the oracle)
        T, _, rc = oracle.divide_conquer(d, mono, match_hash=True, extended=True)
        assert rc == 0
        tp = pose_param_err(T["stVal"], z["result.stVal"], z["result.stno"])
        ot = pose_param_err(G["stVal"], T["stVal"], z["result.stno"])
        print(f"{name}: long-double twin vs the chain {tp:.2e}, oracle vs its twin {ot:.2e}")
        assert 1e-7 < tp < bar and 1e-7 < ot < bar, (tp, ot)


def test_generator_visibility_index_equals_a_pass_over_all_points():
    """synth._Visibility (two binary searches per window) against the definition, on the three kinds of path: laps, laps with
    skip links (second windows out of order), aerial strips (second windows run against the first ones)."""
    for kw in (dict(lap=12, home=4, revisit=0.5), dict(lap=8, home=4, revisit=0.6, skip=3), dict(strip=10, spacing=3.5)):
        pos, Rw, starts, starts2, pts = synth._world(70, 5, 4, 3, **kw)
        seen = synth._Visibility(starts, starts2, 4)
        assert np.any(starts2 != synth.NEVER)
        for first in range(0, 66):
            for last in (first + 1, first + 2):
                a = (starts <= first) & (starts + 3 >= last)
                b = (starts2 <= first) & (starts2 + 3 >= last)
                assert np.array_equal(seen(first, last), np.nonzero(a | b)[0]), (kw, first, last)


def test_aerial_block_structure():
    """The AP_Vaihingen stand-in: parallel strips flown to and fro; a strip shares features with the NEXT strip along its whole
    length (and with nothing further away), the first strip's maps look along +x, the second strip's along -x."""
    typ, maps = synth.make_config("aerial")
    assert typ == "Monocular" and len(maps) == 238
    L = synth.AERIAL["strip"]
    ids = [set(np.asarray(m.stno)[np.asarray(m.stno) > 0].tolist()) for m in maps]
    for k in (2, 7, 13):                                  # map k of strip 0: frames k .. k+2
        across = [len(ids[k] & ids[j]) for j in range(L, 2 * L - 2)]
        assert max(across) > 30 and sum(1 for c in across if c) <= 8   # the few maps of strip 1 that pass the same place
        assert all(len(ids[k] & ids[j]) == 0 for j in range(2 * L, 3 * L - 2))
    assert maps[3].Sign == 1 and maps[L + 3].Sign == -1 and maps[3].Fix == maps[L + 3].Fix


def test_generator_speed_of_the_large_sets():
    """A 16 384-map set must build in well under a minute (the per-map visibility pass of round 2 made it quadratic)."""
    import time
    t0 = time.time()
    typ, maps = synth.make_config("synth16k", 16384, only=(0, 1500))
    dt = time.time() - t0
    assert len(maps) == 1500 and dt < 10.0, dt
