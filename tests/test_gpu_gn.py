"""-m gpu: lsfm_gn_polish (csrc/lsfm_gn.hip) -- the Gauss-Newton polish of the map-joining objective over all local maps -- against
the oracle's statement of the same steps (oracle/lsfm_gn.inc) and by its properties.  The reference has no iterative step: parity is
UNPINNED for this entry point; tests/test_gn_cpu.py pins the oracle's step to the objective itself."""
import os
import subprocess

import numpy as np
import pytest

from common import feat_param_err, pose_param_err
from linearsfm_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dicts(oracle, maps):
    return [oracle.localmap_to_dict(m) for m in maps]


@pytest.mark.parametrize("mono,n,npf,vis,kw,steps", [
    (False, 2, 6, 4, {}, 3), (False, 9, 8, 4, {}, 3), (False, 40, 8, 5, dict(lap=12, home=4, revisit=0.5), 3), (False, 200, 20, 5, dict(lap=50), 2),
    (True, 2, 6, 4, {}, 2), (True, 9, 8, 4, {}, 2), (True, 40, 8, 4, synth.SPIRAL, 2), (True, 120, 20, 4, synth.SPIRAL, 2)])
def test_polish_steps_vs_oracle(ctx, oracle, mono, n, npf, vis, kw, steps):
    """The same steps from the same start (the ORACLE's tree result, so that both start from identical bits): objective and gradient of
    every iterate, and the polished state.  Sets with one map in the global frame (Stereo: map 1), poses that are hubs of their own
    map (Mono), features seen from hub and own pose alike (repeated (pose, feature) blocks in the joint W), lap closures."""
    maps = synth.make_mono_set(n, npf, vis, seed=9, **kw) if mono else synth.make_stereo_set(n, npf, vis, seed=9, **kw)
    d = _dicts(oracle, maps)
    G, _, rc = oracle.divide_conquer(d, mono)
    assert rc == 0
    exp, eobj, egn, ehv, erc = oracle.gn_polish(d, mono, G, steps)
    got, obj, gn, hv, rc = ctx.gn_polish(d, mono, G, steps)
    assert rc == 0 and erc == 0
    assert np.array_equal(hv, ehv)
    assert np.max(np.abs(obj - eobj) / eobj) < 1e-9, (obj, eobj)
    # (the gradient of a later iterate is a small difference of large terms: it is held against the first one's size)
    assert np.max(np.abs(gn - egn)) < 1e-7 * egn[0] + 1e-6 * np.max(egn[1:]), (gn, egn)
    tol = 1e-6
    assert pose_param_err(got, exp, G["stno"]) < tol and feat_param_err(got, exp, G["stno"]) < tol


@pytest.mark.parametrize("config", ["rs90", "nc3500-512"])
def test_polish_properties_on_the_named_sets(ctx, oracle, config):
    """RS90-like (Mono, 88 maps, its full size) and the first 512 maps of the NC3500-like set (Stereo), from the DEVICE's own tree result:
    the objective never rises, the first step already takes the gradient down by an order of magnitude (Stereo: three steps five
    orders), the gauge scalars stay, and the state the device ends at scores the same objective under the oracle's evaluation."""
    if config == "rs90":
        typ, maps = synth.make_config("rs90")
        steps = 6
    else:
        typ, maps = synth.make_config("nc3500", 512)
        steps = 3
    mono = typ == "Monocular"
    d = _dicts(oracle, maps)
    G, stats, rc = ctx.divide_conquer(d, mono)
    assert rc == 0
    st, obj, gn, hv, rc = ctx.gn_polish(d, mono, G, steps)
    assert rc == 0
    assert np.all(np.diff(obj) <= 1e-12 * obj[0]) and obj[-1] < obj[0]
    assert np.all(hv <= 8)
    assert gn[1] * 10 <= gn[0], gn
    if not mono:
        assert gn[-1] * 1e5 <= gn[0], gn
    F, _ = oracle.gn_objective(d, mono, dict(G, stVal=st), False)
    assert abs(F - obj[-1]) <= 1e-9 * F
    F0, _ = oracle.gn_objective(d, mono, G, False)
    assert abs(F0 - obj[0]) <= 1e-9 * F0
    print(f"{config}: F {obj[0]:.6f} -> {obj[-1]:.6f}, gradient {gn[0]:.3e} -> {gn[-1]:.3e}, halvings {hv.tolist()}")


def test_minimiser_is_a_fixed_point(ctx, oracle):
    maps = synth.make_stereo_set(24, 10, 5, seed=3, lap=12, home=4, revisit=0.5)
    d = _dicts(oracle, maps)
    G, _, rc = ctx.divide_conquer(d, False)
    st, obj, gn, _, rc = ctx.gn_polish(d, False, G, 5)
    assert rc == 0 and gn[-1] * 1e6 < gn[0]
    st2, obj2, gn2, hv2, rc = ctx.gn_polish(d, False, dict(G, stVal=st), 1)
    assert rc == 0
    assert np.max(np.abs(st2 - st)) < 1e-9 and abs(obj2[1] - obj2[0]) <= 1e-10 * obj2[0]


def test_polish_refuses_what_it_cannot_place(ctx, oracle):
    from linearsfm_amd import api
    maps = synth.make_stereo_set(5, 6, 4, seed=2)
    d = _dicts(oracle, maps)
    G, _, rc = ctx.divide_conquer(d, False)
    with pytest.raises(api.LsfmError):   # a local feature that is not in the global state
        ctx.gn_polish(d, False, dict(G, n=G["n"] - 1, stno=G["stno"][:-3], stVal=G["stVal"][:-3]), 1)
    with pytest.raises(api.LsfmError):   # a global feature that no map holds
        ctx.gn_polish(d, False, dict(G, n=G["n"] + 1, stno=np.concatenate([G["stno"], [999999] * 3]).astype(np.int32),
                                     stVal=np.concatenate([G["stVal"], [0.0, 0.0, 1.0]])), 1)
    # and the context is still usable
    st, obj, gn, hv, rc = ctx.gn_polish(d, False, G, 1)
    assert rc == 0 and obj[1] <= obj[0]


@pytest.mark.parametrize("typ", ["Stereo", "Monocular"])
def test_cli_gn_flag(oracle, tmp_path, typ):
    """LinearSFM -gn <steps>: the tree, then the polish, files written from the polished state; without the flag nothing changes."""
    mono = typ == "Monocular"
    maps = synth.make_mono_set(7, 8, 4, seed=4) if mono else synth.make_stereo_set(7, 8, 4, seed=4)
    dd = tmp_path / "set"
    synth.write_set(str(dd), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    out = {}
    for tag, extra in (("plain", []), ("gn", ["-gn", "2"])):
        fb = str(tmp_path / f"{tag}.bin")
        r = subprocess.run([exe, "-path", str(dd), "-num", "7", "-type", typ, "-fullbin", fb] + extra, capture_output=True, text=True, check=True)
        raw = open(fb, "rb").read()
        n = int(np.frombuffer(raw[:4], np.int32)[0])
        out[tag] = (np.frombuffer(raw[8:8 + 4 * n], np.int32), np.frombuffer(raw[8 + 4 * (n + (n & 1)):], np.float64), r.stdout)
    assert "Gauss-Newton Step 2:" in out["gn"][2] and "Gauss-Newton" not in out["plain"][2]
    d = _dicts(oracle, maps)
    G, _, rc = oracle.divide_conquer(d, mono)
    exp, _, _, _, _ = oracle.gn_polish(d, mono, G, 2)
    assert np.array_equal(out["gn"][0], G["stno"])
    assert pose_param_err(out["plain"][1], G["stVal"], G["stno"]) < 1e-6
    assert pose_param_err(out["gn"][1], exp, G["stno"]) < 1e-6 and feat_param_err(out["gn"][1], exp, G["stno"]) < 1e-6
