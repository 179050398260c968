"""The drop-in process boundary: linearsfm_amd/LinearSFM (HIP library behind the reference's command line and file
formats, LinearSFMImp.cpp:7989-8105) against the oracle's CLI on the same localmap_k.txt files."""
import os
import subprocess

import numpy as np
import pytest

from linearsfm_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _table(path):
    return np.array([[float(x) for x in line.split()] for line in open(path)])


@pytest.mark.parametrize("typ", ["Stereo", "Monocular"])
def test_cli_matches_oracle_cli(oracle, tmp_path, typ):
    mono = typ == "Monocular"
    maps = synth.make_mono_set(6, 8, 4, seed=41) if mono else synth.make_stereo_set(6, 6, 4, seed=41)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    ora = os.path.join(ROOT, "oracle", "lsfm_oracle")
    assert os.path.exists(exe), "build() must have produced the CLI"
    outs = {}
    for tag, binary in (("hip", exe), ("oracle", ora)):
        p, f, s = (str(tmp_path / f"{tag}_{n}.txt") for n in ("Pose", "Feature", "State"))
        r = subprocess.run([binary, "-path", str(d), "-num", "6", "-type", typ, "-p", p, "-f", f, "-st", s],
                           capture_output=True, text=True, check=True)
        outs[tag] = (r.stdout, _table(p), _table(f), _table(s))
    so, sh = outs["oracle"][0], outs["hip"][0]
    # same progress lines (LinearSFMImp.cpp:1952, 1995, 2072)
    strip = lambda t: [l for l in t.splitlines() if l and not l.startswith("Total Used Time")]
    assert strip(so) == strip(sh)
    assert "Total Used Time:" in sh
    for a, b in zip(outs["hip"][1:], outs["oracle"][1:]):
        assert a.shape == b.shape
        assert np.array_equal(a[:, 0], b[:, 0])            # ids, sorted
        assert np.max(np.abs(a[:, 1:] - b[:, 1:])) <= 2e-6  # "%lf": 6 decimals


@pytest.mark.parametrize("name", ["stereo_n5", "mono_n5"])
def test_cli_files_byte_for_byte_vs_reference_writers(tmp_path, name):
    """The command line on the input maps of a golden tree: -st / -p / -f files against the REAL reference writers' bytes.
    tests/golden/writers.npz holds what lmj_SaveStateVector / lmj_SavePoses_3DPF (LinearSFMImp.cpp:2102-2117, 7876-7967) wrote for
    the fixture's final state; the device's state (taken exactly from -fullbin) agrees with that state to 1e-9, so its files are
    (a) exactly what the pinned Python statement of the writers makes of the device's own state, and (b) the golden bytes except
    where a value sits within 1e-9 of a sixth-decimal rounding boundary (at most a line or two; normally none).  `-p` without
    `-f` writes nothing (Imp.cpp:2078: only together)."""
    from common import load_golden, get_map, reference_writer_bytes
    from linearsfm_amd import api
    z = load_golden(name + ".npz")
    mono = str(z["type"]) == "Monocular"
    N = int(z["N"])
    d = tmp_path / "set"
    os.makedirs(d)
    for k in range(N):
        api.write_localmap(str(d / f"localmap_{k + 1}.txt"), get_map(z, f"in{k}"), mono)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    fp, ff, fs, fb, lone = (str(tmp_path / x) for x in ("Pose.txt", "Feature.txt", "State.txt", "state.bin", "LonePose.txt"))
    typ = "Monocular" if mono else "Stereo"
    subprocess.run([exe, "-path", str(d), "-num", str(N), "-type", typ, "-p", fp, "-f", ff, "-st", fs, "-fullbin", fb],
                   capture_output=True, text=True, check=True)
    raw = open(fb, "rb").read()
    n = int(np.frombuffer(raw[:4], np.int32)[0])
    stno = np.frombuffer(raw[8:8 + 4 * n], np.int32)
    st = np.frombuffer(raw[8 + 4 * (n + (n & 1)):], np.float64)
    exp = get_map(z, "result")
    assert np.array_equal(stno, exp["stno"])
    assert np.max(np.abs(st - exp["stVal"]) / np.maximum(1.0, np.abs(exp["stVal"]))) < 1e-9
    got = tuple(open(p, "rb").read() for p in (fs, fp, ff))
    assert got == reference_writer_bytes(stno, st)                      # (a)
    w = np.load(os.path.join(ROOT, "tests", "golden", "writers.npz"))
    for g, key in zip(got, ("state", "both.pose", "both.feat")):        # (b)
        gl, rl = g.split(b"\n"), bytes(w[f"{name}.{key}"]).split(b"\n")
        assert len(gl) == len(rl)
        assert sum(a != b for a, b in zip(gl, rl)) <= 2, key
    subprocess.run([exe, "-path", str(d), "-num", str(N), "-type", typ, "-p", lone], capture_output=True, text=True, check=True)
    assert not os.path.exists(lone)
    # (c) where the compiled reference travelled with the snapshot (oracle/_ref/, built in the authoring container from the sources
    # where they lie; no file of /root/reference is read): the REAL writers on the device's own state, byte for byte
    ref_dump = os.path.join(ROOT, "oracle", "_ref", "ref_dump")
    if os.path.exists(ref_dump):
        fi = str(tmp_path / "in.bin")
        with open(fi, "wb") as f:
            for nm, a in (("stno", np.ascontiguousarray(stno)), ("stVal", np.ascontiguousarray(st))):
                f.write(f"{nm} {'f8' if a.dtype == np.float64 else 'i4'} {a.size}\n".encode())
                f.write(a.tobytes())
        rp, rf, rs = (str(tmp_path / x) for x in ("RefPose.txt", "RefFeature.txt", "RefState.txt"))
        subprocess.run([ref_dump, "save", "both", fi, rp, rf, rs], check=True)
        assert got == tuple(open(p, "rb").read() for p in (rs, rp, rf))


def test_cli_errors_like_the_reference(tmp_path):
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    r = subprocess.run([exe, "-num", "3", "-type", "Stereo"], capture_output=True, text=True)
    assert r.returncode == 0 and "LinerSFM Error: Please Input Right File Path:" in r.stdout   # Imp.cpp:8075-8076
    r = subprocess.run([exe, "-help"], capture_output=True, text=True)
    assert "Linear SFM Solution General Options" in r.stdout


def test_cli_level_scheduled_solves_and_info_export(oracle, tmp_path):
    """LSFM_LEVEL_SOLVE=1 forces the triangular solves that go by elimination-tree level (the path taken when a task's
    slice of the vector does not fit LDS); -info stores the final map with its information matrix, re-readable."""
    from linearsfm_amd import api
    maps = synth.make_stereo_set(12, 6, 4, seed=43)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    full = {}
    for tag, env in (("tasks", {}), ("levels", {"LSFM_LEVEL_SOLVE": "1"})):
        s, info = str(tmp_path / f"{tag}_full.txt"), str(tmp_path / f"{tag}_info.txt")
        subprocess.run([exe, "-path", str(d), "-num", "12", "-type", "Stereo", "-full", s, "-info", info],
                       capture_output=True, text=True, check=True, env=dict(os.environ, **env))
        full[tag] = (_table(s), api.read_localmap(info, False))
    exp, _, rc = oracle.divide_conquer([oracle.localmap_to_dict(m) for m in maps], False)
    assert rc == 0
    for tag in full:
        st, info = full[tag]
        assert np.array_equal(st[:, 0], exp["stno"])
        assert np.max(np.abs(st[:, 1] - exp["stVal"]) / np.maximum(1.0, np.abs(exp["stVal"]))) < 1e-6, tag
        # the exported map is the final map: same state, same information blocks
        assert np.array_equal(info["stno"], exp["stno"]) and np.array_equal(info["photo"], exp["photo"])
        for k in ("U", "W", "V"):
            assert np.max(np.abs(np.asarray(info[k]) - np.asarray(exp[k]))) / np.max(np.abs(np.asarray(exp[k]))) < 1e-6, (tag, k)


def test_cli_forward_substitution_inside_and_outside_the_factorisation(tmp_path):
    """The first preconditioner application of a level takes its forward substitution inside the supernodal factorisation
    (right-hand side as one more panel row of k_sn_panel); LSFM_NO_FUSED_FWD=1 runs it as launches of its own (k_sn_fwd), the
    path a second refinement step takes.  A path that revisits (lap=50) gives separators wide enough for supernode groups."""
    maps = synth.make_stereo_set(200, 20, 5, seed=77, lap=50)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    out = {}
    for tag, env in (("fused", {}), ("apart", {"LSFM_NO_FUSED_FWD": "1"})):
        s = str(tmp_path / f"{tag}_full.txt")
        subprocess.run([exe, "-path", str(d), "-num", "200", "-type", "Stereo", "-full", s], capture_output=True, text=True, check=True,
                       env=dict(os.environ, **env))
        out[tag] = _table(s)
    assert np.array_equal(out["fused"][:, 0], out["apart"][:, 0])
    assert np.max(np.abs(out["fused"][:, 1] - out["apart"][:, 1]) / np.maximum(1.0, np.abs(out["apart"][:, 1]))) < 1e-9


@pytest.mark.parametrize("typ,n", [("Stereo", 13), ("Monocular", 9)])
def test_rccl_allreduce_hook_from_cpp(tmp_path, typ, n):
    """liblsfm_rccl.so (RCCL behind lsfm_allreduce_fn, for C / C++ hosts) on one GPU: linearsfm_amd/lsfm_rccl_selftest joins the set
    as one tree and as two blocks + a top tree whose sums all go through ncclAllReduce on the library's stream (a communicator
    of one rank), and compares."""
    mono = typ == "Monocular"
    maps = synth.make_mono_set(n, 8, 4, seed=43, **synth.SPIRAL) if mono else synth.make_stereo_set(n, 8, 5, seed=43, lap=30, home=5)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "lsfm_rccl_selftest")
    assert os.path.exists(exe), "build() must have produced the self-test"
    r = subprocess.run([exe, "-path", str(d), "-num", str(n), "-type", typ], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all-reduces" in r.stdout


def test_cli_arenas_grow_when_a_run_exhausts_them(oracle, tmp_path):
    """The device arenas start at a fraction of the upper bound the upload asks for (a cold process paid seconds of hipMalloc for the
    whole bound) and are doubled when a run exhausts one -- the run starts over, the result is the same.  Here: a floor of 1 MiB and
    1/4096 of the bound, so that a small tree has to grow them several times."""
    maps = synth.make_stereo_set(40, 30, 5, seed=44, lap=20, home=5)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    full = {}
    for tag, env in (("grown", {"LSFM_ARENA_DIV": "4096", "LSFM_ARENA_MIN_MB": "1", "LSFM_DEBUG": "1"}), ("whole", {"LSFM_ARENA_DIV": "1"})):
        s = str(tmp_path / f"{tag}_full.txt")
        r = subprocess.run([exe, "-path", str(d), "-num", "40", "-type", "Stereo", "-full", s], capture_output=True, text=True, check=True,
                           env=dict(os.environ, **env))
        full[tag] = (_table(s), r.stderr)
    assert "arenas grown to" in full["grown"][1] and "arenas grown to" not in full["whole"][1]
    a, b = full["grown"][0], full["whole"][0]
    assert np.array_equal(a[:, 0], b[:, 0])
    assert np.max(np.abs(a[:, 1] - b[:, 1]) / np.maximum(1.0, np.abs(b[:, 1]))) < 1e-9


def test_cli_with_roctx_ranges_switched_on(tmp_path):
    """LSFM_ROCTX=1: the library looks up the roctx library at run time and brackets tree run / levels / stages; without a
    profiler attached the ranges go nowhere and the run is the run it was (the profile: profiles/r04_roctx_marker_stats.csv)."""
    maps = synth.make_stereo_set(6, 6, 4, seed=12)
    d = tmp_path / "set"
    synth.write_set(str(d), maps)
    exe = os.path.join(ROOT, "linearsfm_amd", "LinearSFM")
    outs = []
    for env in ({}, {"LSFM_ROCTX": "1"}):
        full = str(tmp_path / f"full{len(outs)}.txt")
        r = subprocess.run([exe, "-path", str(d), "-num", "6", "-type", "Stereo", "-full", full], capture_output=True, text=True,
                           env=dict(os.environ, **env))
        assert r.returncode == 0 and "no roctx library" not in r.stderr, r.stderr
        outs.append(_table(full))
    assert np.array_equal(outs[0][:, 0], outs[1][:, 0]) and np.max(np.abs(outs[0][:, 1] - outs[1][:, 1])) < 1e-9
