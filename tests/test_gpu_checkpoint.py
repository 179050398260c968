"""Level checkpoint / resume (include/lsfm.h lsfm_tree_set_stop_level ...).  The reference keeps the nodes of every
level in RAM (m_LMsetS[i] = m_GMapS, LinearSFMImp.cpp:2032) and can neither store nor reload them; here a run can end after L
levels, its nodes go through the local-map text format (with the trailer the reference's reader never reaches) and a second tree
built from them finishes the job -- with the result of the uninterrupted run."""
import os
import subprocess

import numpy as np
import pytest

from linearsfm_amd import api, synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RESUME_TOL = 1e-9   # both runs solve the same camera systems to a 1e-10 residual; the nodes travel at %.17g (exact)


def _sets(mono, n):
    return synth.make_mono_set(n, 8, 4, seed=77) if mono else synth.make_stereo_set(n, 6, 4, seed=77)


@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("n_maps,stop", [(13, 1), (13, 2), (16, 3), (7, 1)])
def test_resume_from_the_nodes_of_a_level(ctx, tmp_path, mono, n_maps, stop):
    maps = [m.__dict__ for m in _sets(mono, n_maps)]
    t = ctx.tree_upload(maps, mono)
    _, rc = ctx.tree_run(t)
    assert rc == 0 and ctx.tree_node_count(t) == 1
    whole = ctx.tree_download(t)
    # the same tree, stopped after `stop` levels
    ctx.tree_set_stop_level(t, stop)
    st, rc = ctx.tree_run(t)
    assert rc == 0
    want = n_maps
    for _ in range(stop):
        want = (want + 1) // 2
    assert ctx.tree_node_count(t) == want and st["joins"] > 0
    with pytest.raises(api.LsfmError):
        ctx.tree_download(t)            # there is no final map yet
    with pytest.raises(api.LsfmError):
        ctx.tree_download_node(t, want)
    nodes = [ctx.tree_download_node(t, k) for k in range(want)]
    # a stop level is a property of the tree, not of one run: 0 gives the whole tree back
    ctx.tree_set_stop_level(t, 0)
    _, rc = ctx.tree_run(t)
    again = ctx.tree_download(t)
    ctx.tree_free(t)
    assert rc == 0 and np.array_equal(again["stno"], whole["stno"])
    # every pose of a node knows the local map it came from; the nodes partition the poses of the inputs
    org = np.concatenate([nd["pose_origin"] for nd in nodes])
    assert org.min() >= 0 and org.max() == n_maps - 1
    # through the files
    d = tmp_path / "nodes"
    d.mkdir()
    for k, nd in enumerate(nodes):
        api.write_localmap(d / f"localmap_{k + 1}.txt", nd, mono)
    back = api.read_localmaps(d, want, mono)
    for a, b in zip(nodes, back):
        for key in ("Ref", "FRef", "m", "n", "nU", "nW") + (("ScaP", "Fix", "Sign", "FScaP", "FFix") if mono else ()):
            assert a[key] == b[key], key
        for key in ("stno", "stVal", "U", "Ui", "Uj", "W", "photo", "feature", "V", "pose_origin"):
            assert np.array_equal(np.asarray(a[key]), np.asarray(b[key])), key   # %.17g: the identity
    t2 = ctx.tree_upload(back, mono)
    _, rc = ctx.tree_run(t2)
    assert rc == 0
    res = ctx.tree_download(t2)
    ctx.tree_free(t2)
    for key in ("Ref", "FRef", "m", "n", "nU", "nW"):
        assert res[key] == whole[key], key
    for key in ("stno", "Ui", "Uj", "photo", "feature", "pose_origin"):
        assert np.array_equal(res[key], whole[key]), key
    for key in ("stVal", "U", "W", "V"):
        a, b = np.asarray(res[key]), np.asarray(whole[key])
        assert np.max(np.abs(a - b)) <= RESUME_TOL * max(1.0, np.max(np.abs(b))), key


def test_a_stop_level_beyond_the_root_is_the_whole_tree(ctx):
    maps = [m.__dict__ for m in _sets(False, 7)]
    t = ctx.tree_upload(maps, False)
    _, rc = ctx.tree_run(t)
    whole = ctx.tree_download(t)
    ctx.tree_set_stop_level(t, 9)          # 7 maps: three levels
    st, rc2 = ctx.tree_run(t)
    assert rc == 0 and rc2 == 0 and st["levels"] == 3 and ctx.tree_node_count(t) == 1
    a, b = ctx.tree_download(t), ctx.tree_download_node(t, 0)
    ctx.tree_free(t)
    for g in (a, b):
        assert g["Ref"] == whole["Ref"] == g["FRef"] and np.array_equal(g["stno"], whole["stno"])
        assert np.max(np.abs(g["stVal"] - whole["stVal"])) <= RESUME_TOL * max(1.0, np.max(np.abs(whole["stVal"])))
    with pytest.raises(api.LsfmError):
        ctx.tree_set_stop_level(None, -1)


@pytest.mark.parametrize("mono", [False, True])
@pytest.mark.parametrize("n_maps,stop", [(13, 2), (10, 1), (21, 3)])
def test_nodes_of_a_level_equal_the_oracles_sub_trees(ctx, oracle, mono, n_maps, stop):
    """What a stopped run holds is what the reference's loop holds at that point (Imp.cpp:1997-2032): node k of level L is the join of
    local maps [k 2^L, (k + 1) 2^L) with its last re-anchoring still to come -- the oracle's tree over those maps without the final
    re-anchoring (orc_set_final_reanchor(0)).  A node of one map is that map."""
    from common import pose_param_err
    maps = _sets(mono, n_maps)
    dicts = [oracle.localmap_to_dict(m) for m in maps]
    t = ctx.tree_upload(dicts, mono)
    ctx.tree_set_stop_level(t, stop)
    _, rc = ctx.tree_run(t)
    assert rc == 0
    span = 1 << stop
    nn = (n_maps + span - 1) // span
    assert ctx.tree_node_count(t) == nn
    for k in range(nn):
        node = ctx.tree_download_node(t, k)
        sub = dicts[k * span:(k + 1) * span]
        exp, _, orc = oracle.divide_conquer(sub, mono, final_reanchor=False)
        assert orc == 0
        for key in ("Ref", "FRef", "m", "n", "nU", "nW") + (("ScaP", "Fix", "FScaP", "FFix") if mono else ()):
            assert node[key] == exp[key], (k, key, node[key], exp[key])
        for key in ("stno", "Ui", "Uj", "photo", "feature"):
            assert np.array_equal(node[key], exp[key]), (k, key)
        assert pose_param_err(node["stVal"], exp["stVal"], exp["stno"]) < 1e-6, k
        for key in ("U", "W", "V"):
            a, b = np.asarray(node[key]), np.asarray(exp[key])
            assert np.max(np.abs(a - b)) <= 1e-6 * max(1.0, np.max(np.abs(b))), (k, key)
        org = np.asarray(node["pose_origin"])
        assert org.min() >= k * span and org.max() < min((k + 1) * span, n_maps)
    ctx.tree_free(t)


def test_cli_stops_after_a_level_and_resumes_from_the_written_nodes(tmp_path):
    maps = synth.make_stereo_set(11, 6, 4, seed=5)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    full, part, nodes = str(tmp_path / "full.txt"), str(tmp_path / "part.txt"), tmp_path / "nodes"
    nodes.mkdir()
    subprocess.run([exe, "-path", str(d), "-num", "11", "-type", "Stereo", "-full", full], capture_output=True, text=True, check=True)
    r = subprocess.run([exe, "-path", str(d), "-num", "11", "-type", "Stereo", "-levels", "2", "-nodes", str(nodes)],
                       capture_output=True, text=True, check=True)
    assert "Stopped After Level 2: 3 Nodes Written" in r.stdout and "Generate Level 3" not in r.stdout
    assert sorted(os.listdir(nodes)) == ["localmap_1.txt", "localmap_2.txt", "localmap_3.txt"]
    r = subprocess.run([exe, "-path", str(nodes), "-num", "3", "-type", "Stereo", "-full", part, "-quiet", "1"],
                       capture_output=True, text=True, check=True)
    assert "Join Level" not in r.stdout and "Total Used Time:" in r.stdout
    a = np.array([[float(x) for x in l.split()] for l in open(full)])
    b = np.array([[float(x) for x in l.split()] for l in open(part)])
    assert np.array_equal(a[:, 0], b[:, 0])
    assert np.max(np.abs(a[:, 1] - b[:, 1])) <= RESUME_TOL * max(1.0, np.max(np.abs(a[:, 1])))
    # flags that make no sense alone are refused
    r = subprocess.run([exe, "-path", str(d), "-num", "11", "-type", "Stereo", "-levels", "2"], capture_output=True, text=True)
    assert r.returncode == 1 and "go together" in r.stderr


def test_cli_binary_cache_json_and_binary_state(tmp_path):
    """-cache: the first run parses the text files and leaves the cache, the second reads the cache (and says so in -json) and gives the
    same files byte for byte; a cache of another set size is not trusted.  -fullbin holds the -full numbers exactly."""
    import json
    maps = synth.make_stereo_set(10, 6, 4, seed=8)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    cache = str(tmp_path / "set.lsfmbin")
    outs = []
    for run in range(2):
        full, fb, js = (str(tmp_path / f"{n}{run}") for n in ("full", "fullbin", "json"))
        subprocess.run([exe, "-path", str(d), "-num", "10", "-type", "Stereo", "-cache", cache, "-full", full, "-fullbin", fb, "-json", js, "-quiet", "1"],
                       capture_output=True, text=True, check=True)
        outs.append((open(full).read(), open(fb, "rb").read(), json.load(open(js))))
    assert outs[0][2]["from_cache"] is False and outs[1][2]["from_cache"] is True
    assert outs[0][2]["maps"] == 10 and outs[0][2]["rc"] == 0 and outs[0][2]["joins"] == 9 and outs[0][2]["t_total_ms"] > 0
    assert set(outs[0][2]["phases_s"]) == {"read", "context", "upload", "join_tree", "download", "write"}
    # same inputs bit for bit: the two runs may differ by the order of their atomic sums only
    a = np.array([[float(x) for x in l.split()] for l in outs[0][0].splitlines()])
    b = np.array([[float(x) for x in l.split()] for l in outs[1][0].splitlines()])
    assert np.array_equal(a[:, 0], b[:, 0]) and np.max(np.abs(a[:, 1] - b[:, 1])) <= RESUME_TOL * max(1.0, np.max(np.abs(a[:, 1])))
    raw = outs[0][1]
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    assert n == len(a) and np.array_equal(np.frombuffer(raw[8:8 + 4 * n], np.int32), a[:, 0].astype(np.int32))
    assert np.array_equal(np.frombuffer(raw[8 + 4 * (n + (n & 1)):], np.float64), a[:, 1])      # %.17g == the doubles
    # asking for more maps than the cache holds: the text files are read (and the cache replaced)
    synth.write_set(str(d), synth.make_stereo_set(12, 6, 4, seed=8))
    js = str(tmp_path / "json2")
    r = subprocess.run([exe, "-path", str(d), "-num", "12", "-type", "Stereo", "-cache", cache, "-json", js, "-quiet", "1"], capture_output=True, text=True, check=True)
    assert "does not hold 12 Stereo maps" in r.stderr and json.load(open(js))["from_cache"] is False
    assert api.mapset_info(cache) == (12, False)


def test_cli_cache_is_not_trusted_once_the_text_files_changed(tmp_path):
    """-cache carries a stamp of what it was made from (resolved -path, size and modification time of every localmap_k.txt): a set that was
    regenerated under the same directory, with the same -num and -type, is parsed again instead of answered from the stale cache (advisor,
    round 4); text files that are gone cannot contradict the cache, which is then all there is."""
    import json
    import shutil
    import time
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    d = tmp_path / "set"
    cache = str(tmp_path / "set.lsfmbin")

    def run(tag):
        full, js = str(tmp_path / f"full_{tag}"), str(tmp_path / f"json_{tag}")
        r = subprocess.run([exe, "-path", str(d), "-num", "8", "-type", "Stereo", "-cache", cache, "-full", full, "-json", js, "-quiet", "1"],
                           capture_output=True, text=True, check=True)
        vals = np.array([[float(x) for x in l.split()] for l in open(full).read().splitlines()])
        return json.load(open(js))["from_cache"], vals, r.stderr

    synth.write_set(str(d), synth.make_stereo_set(8, 6, 4, seed=8))
    c0, a, _ = run("a")
    c1, b, _ = run("b")
    assert c0 is False and c1 is True
    time.sleep(0.05)
    synth.write_set(str(d), synth.make_stereo_set(8, 6, 4, seed=9))  # other values, same directory, same -num / -type
    c2, c, err = run("c")
    assert c2 is False and "not made from the text files" in err
    assert np.max(np.abs(c[:, 1] - a[:, 1])) > 1e-3  # the new set's map, not the cached one's
    c3, _, _ = run("d")
    assert c3 is True  # (the cache was replaced and stamped anew)
    shutil.rmtree(d)
    os.makedirs(d)
    c4, e, _ = run("e")
    assert c4 is True and np.max(np.abs(e[:, 1] - c[:, 1])) <= RESUME_TOL * max(1.0, np.max(np.abs(c[:, 1])))
