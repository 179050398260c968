"""CPU-only: the oracle's Gauss-Newton polish of the map-joining objective (oracle/lsfm_gn.inc) pinned by PROPERTIES -- the reference
has no iterative step, nothing of it can check the result (parity unpinned): the right-hand side of a step against central differences of
the objective, the matrix of a step against differences of the right-hand side, the objective never rising, the gradient falling, a
minimiser being a fixed point, the long-double twin."""
import numpy as np
import pytest

from linearsfm_amd import synth


def _set(mono, n, noise=1e-3, **kw):
    maps = synth.make_mono_set(n, 8, 4, seed=5, noise=noise, **kw) if mono else synth.make_stereo_set(n, 8, 4, seed=5, noise=noise, **kw)
    return maps


def _free_mask(G, mono):
    R = 6 * G["m"] + 3 * G["n"]
    free = np.ones(R, bool)
    if mono:
        ids = -np.asarray(G["stno"])[:6 * G["m"]:6]
        pr, ps = int(np.where(ids == G["Ref"])[0][0]), int(np.where(ids == G["ScaP"])[0][0])
        free[6 * pr:6 * pr + 6] = False
        free[6 * ps + G["Fix"]] = False
    return free


@pytest.mark.parametrize("mono", [False, True])
def test_right_hand_side_is_minus_half_the_gradient(oracle, mono):
    """b = sum_k J_k^T I_k r_k against central differences of F(x) = sum_k ||x^_k - f_k(x)||^2_{I_k}: the Jacobian the oracle takes from
    the reference's transform (J1 / J2 / J3, Imp.cpp:485-683 / 3383-3688) IS the derivative of the frame change the objective is made of."""
    d = [oracle.localmap_to_dict(m) for m in _set(mono, 9)]
    G, _, rc = oracle.divide_conquer(d, mono)
    assert rc == 0
    F, b = oracle.gn_objective(d, mono, G)
    assert F > 0
    x = G["stVal"].copy()
    rng = np.random.default_rng(0)
    free = _free_mask(G, mono)
    idx = rng.choice(np.nonzero(free)[0], 60, replace=False)
    scale = np.abs(b).max()
    for i in idx:
        h = 1e-6
        Fp, _ = oracle.gn_objective(d, mono, dict(G, stVal=x + h * (np.arange(len(x)) == i)), False)
        Fm, _ = oracle.gn_objective(d, mono, dict(G, stVal=x - h * (np.arange(len(x)) == i)), False)
        num = (Fp - Fm) / (2 * h)
        assert abs(num + 2 * b[i]) <= 1e-4 * max(abs(num), 1e-3 * scale), (i, num, b[i])


@pytest.mark.parametrize("mono", [False, True])
def test_step_matrix_is_the_derivative_of_the_right_hand_side_at_consistent_maps(oracle, mono):
    """With noise-free local maps the residuals vanish at the joined state and H = sum J^T I J is exactly -db/dx: H v (from the assembled U /
    W / V blocks, the layout the solver takes) against central differences of b along v -- hub columns, hub-hub blocks, diagonal blocks
    stored in full, Mono's second hub and the poses that ARE hubs included."""
    d = [oracle.localmap_to_dict(m) for m in _set(mono, 9, noise=0.0)]
    G, _, rc = oracle.divide_conquer(d, mono)
    assert rc == 0
    x = G["stVal"].copy()
    free = _free_mask(G, mono)
    rng = np.random.default_rng(1)
    m6 = 6 * G["m"]
    for trial in range(3):
        v = rng.normal(size=len(x)) * free
        if trial == 1:
            v[m6:] = 0
        if trial == 2:
            v[:m6] = 0
        h = 1e-6
        _, bp = oracle.gn_objective(d, mono, dict(G, stVal=x + h * v))
        _, bm = oracle.gn_objective(d, mono, dict(G, stVal=x - h * v))
        num = -(bp - bm) / (2 * h)
        Hv = oracle.gn_hessian_times(d, mono, G, v)
        assert np.max(np.abs((num - Hv) * free)) <= 1e-9 * np.max(np.abs(Hv)), trial


@pytest.mark.parametrize("mono,n,kw,steps,drop", [(False, 9, {}, 4, 1e5), (False, 40, dict(lap=12, home=4, revisit=0.5), 4, 1e5),
                                                  (True, 9, {}, 25, 50.0), (True, 40, synth.SPIRAL, 25, 1e3)])
def test_polish_properties(oracle, mono, n, kw, steps, drop):
    """F never rises, the gradient falls (Stereo: quadratically -- five decades in four steps; Mono, whose local maps leave depth and scale
    weak and whose Gauss-Newton matrix therefore misses curvature the residuals add, linearly), and the polished state is a fixed point:
    one more step moves nothing and changes F by rounding only."""
    d = [oracle.localmap_to_dict(m) for m in _set(mono, n, **kw)]
    G, _, rc = oracle.divide_conquer(d, mono)
    assert rc == 0
    st, obj, gn, hv, rc = oracle.gn_polish(d, mono, G, steps)
    assert rc == 0
    assert np.all(np.diff(obj) <= 1e-12 * obj[0])
    assert obj[-1] < obj[0]
    assert gn[-1] * drop <= gn[0], (gn[0], gn[-1])
    assert np.all(hv <= 8)
    if not mono:
        st2, obj2, gn2, _, rc = oracle.gn_polish(d, mono, dict(G, stVal=st), 1)
        assert abs(obj2[1] - obj2[0]) <= 1e-10 * obj2[0]
        assert np.max(np.abs(st2 - st)) <= 1e-9
    # the gauge did not move
    free = _free_mask(G, mono)
    assert np.array_equal(st[~free], np.asarray(G["stVal"])[~free])


@pytest.mark.parametrize("mono", [False, True])
def test_long_double_twin_of_the_steps(oracle, mono):
    """orc_set_extended: the Schur complements, factorisations and back-substitutions of the steps in long double -- same iterates to the
    precision the steps' systems allow."""
    d = [oracle.localmap_to_dict(m) for m in _set(mono, 12, **(synth.SPIRAL if mono else {}))]
    G, _, rc = oracle.divide_conquer(d, mono)
    a, obj_a, _, _, _ = oracle.gn_polish(d, mono, G, 2)
    b, obj_b, _, _, _ = oracle.gn_polish(d, mono, G, 2, extended=True)
    assert np.max(np.abs(a - b) / np.maximum(1.0, np.abs(b))) < (1e-7 if mono else 1e-10)
    assert np.max(np.abs(obj_a - obj_b) / obj_b) < 1e-9


def test_invalid_inputs_are_refused(oracle):
    d = [oracle.localmap_to_dict(m) for m in _set(False, 5)]
    G, _, _ = oracle.divide_conquer(d, False)
    # a feature of the global state that no map holds
    G2 = dict(G, n=G["n"] + 1, stno=np.concatenate([G["stno"], [999999] * 3]).astype(np.int32), stVal=np.concatenate([G["stVal"], [0.0, 0.0, 1.0]]))
    with pytest.raises(ValueError):
        oracle.gn_objective(d, False, G2, False)
    # a local feature that is not in the global state
    G3 = dict(G, n=G["n"] - 1, stno=G["stno"][:-3], stVal=G["stVal"][:-3])
    with pytest.raises(ValueError):
        oracle.gn_objective(d, False, G3, False)
