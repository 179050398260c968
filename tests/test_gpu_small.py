"""The one-launch dense path for camera systems of at most 16 poses (lsfm_small.hip: one work-group per join, S in LDS, dense
Cholesky) against the sparse level pipeline (K7-K11: Schur panels, supernodal Cholesky, refinement, back-substitution), against
the dense LAPACK expected value of every system the REAL reference assembled, and against the oracle's trees.  The small golden
fixtures (trees of 2-8 maps, 2-9 poses a join) are exactly its size.  By default the path takes systems of at most 5 poses (where it
beats the pipeline); the tests here open it to all 16 the kernel holds."""
import numpy as np
import pytest

from common import GOLD_SMALL, dense_reference_solve, feat_param_err, golden_system, load_golden, pose_param_err
from linearsfm_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture()
def both(ctx):
    """runs f with the dense path for every system it can hold (16 poses: all four panel heights of the kernel) and with it off;
    always leaves the default (5) behind"""
    def run(f):
        try:
            ctx.set_small_solve(16)
            a = f()
            ctx.set_small_solve(0)
            b = f()
        finally:
            ctx.set_small_solve(5)
        return a, b
    return run


@pytest.mark.parametrize("name", GOLD_SMALL)
def test_small_path_and_pipeline_vs_dense_lapack(ctx, both, name):
    """lsfm_solve_{stereo,mono} on every reference-assembled system of the small fixtures, through BOTH paths: each within 1e-10 of the
    exact solution (dense LAPACK + extended-precision refinement, nothing of the oracle), Mono gauge scalars at their values."""
    z = load_golden(name)
    for j in range(int(z["njoins"])):
        J, ea, eb, mono, sa = golden_system(z, j)
        assert J["m"] <= 16
        (st_s, rc_s), (st_p, rc_p) = both(lambda: ctx.solve(J, ea, eb, mono, sa))
        assert rc_s == 0 and rc_p == 0
        xd = z[f"join{j}.dense_sol"]
        live = dense_reference_solve(J, ea, eb, mono, sa, IV=z[f"join{j}.parts.IV"])
        assert np.max(np.abs(live - xd) / np.maximum(1, np.abs(xd))) < 1e-13
        es = float(np.max(np.abs(st_s - xd) / np.maximum(1, np.abs(xd))))
        ep = float(np.max(np.abs(st_p - xd) / np.maximum(1, np.abs(xd))))
        assert es < 1e-10 and ep < 1e-10, (name, j, es, ep)
        if mono:
            assert st_s[sa[2]] == sa[3] and np.all(st_s[6 * sa[0]:6 * sa[0] + 6] == 0.0)


@pytest.mark.parametrize("mono,n_maps", [(False, 8), (False, 37), (True, 8), (True, 21)])
def test_trees_through_the_small_path(ctx, both, oracle, mono, n_maps):
    """Whole trees: the lowest levels (systems of 2-16 poses) by the dense path, the rest by the pipeline -- against the same tree
    with every level in the pipeline (1e-9: both are direct solves of the same systems) and against the ORACLE (1e-6, BASELINE.json);
    lsfm_stats.small_levels says how many levels took the path."""
    maps = synth.make_mono_set(n_maps, 8, 4, seed=5, **synth.SPIRAL) if mono else synth.make_stereo_set(n_maps, 8, 5, seed=5, lap=30, home=5)
    dicts = [dict(m.__dict__) for m in maps]
    (a, sa, rca), (b, sb, rcb) = both(lambda: ctx.divide_conquer(dicts, mono))
    assert rca == 0 and rcb == 0
    # (16 poses: Stereo levels 0-3; Mono maps hold three poses each: levels 0-2)
    assert sa["small_levels"] >= (3 if n_maps >= 8 else 1) and sb["small_levels"] == 0, (sa["small_levels"], sb["small_levels"])
    assert sa["t_small_ms"] > 0.0
    for k in ("stno", "Ui", "Uj", "photo", "feature"):
        assert np.array_equal(a[k], b[k]), k
    # (Mono: two runs of the SAME path differ by up to 7e-9 on sets of this kind -- profiles/r05_sharded_noise.txt, "single tree vs
    # itself" -- the monocular scale is observable through shared points only; Stereo: 1e-12)
    tol = 1e-7 if mono else 1e-9
    assert pose_param_err(a["stVal"], b["stVal"], b["stno"]) < tol
    assert feat_param_err(a["stVal"], b["stVal"], b["stno"]) < tol
    exp, _, rc = oracle.divide_conquer(dicts, mono)
    assert rc == 0
    assert np.array_equal(a["stno"], exp["stno"])
    assert pose_param_err(a["stVal"], exp["stVal"], exp["stno"]) < 1e-6
    assert feat_param_err(a["stVal"], exp["stVal"], exp["stno"]) < 1e-6


def test_small_path_reports_a_system_that_is_not_positive_definite(ctx):
    """A camera system whose information matrix is indefinite: the dense path's Cholesky meets a non-positive pivot and the call
    fails with LSFM_ERR_NOT_SPD's message, like the pipeline's factorisation."""
    from linearsfm_amd import api
    z = load_golden("stereo_n3.npz")
    J, ea, eb, mono, sa = golden_system(z, 0)
    J = dict(J)
    J["U"] = -np.asarray(J["U"])
    assert J["m"] <= 5  # (the default path takes it)
    with pytest.raises(api.LsfmError, match="not positive definite"):
        ctx.solve(J, ea, eb, mono, sa)


def test_default_takes_the_two_lowest_stereo_levels(ctx):
    """lsfm_set_small_solve's default (5 poses): an 8-map Stereo tree has joins of 2, 4 and 8 poses -- two levels on the dense path."""
    maps = synth.make_stereo_set(8, 8, 5, seed=6)
    _, st, rc = ctx.divide_conquer([dict(m.__dict__) for m in maps], False)
    assert rc == 0 and st["small_levels"] == 2, st["small_levels"]
