"""CPU-only tests of the host half of the solver (linearsfm_amd/csrc/lsfm_symbolic.cpp through lsfm_symbolic_analyse): the
ordering + symbolic block Cholesky that stands where the reference calls cholmod_amd / cholmod_analyze_p in every join
(Imp.cpp:2413, 2440 / 7081, 7112).  Checked against a brute-force symbolic elimination in Python sets."""
import numpy as np
import pytest

from common import golden_system, load_golden
from linearsfm_amd import api, synth


def _brute_force_fill(m, rowptr, colidx, perm):
    """pattern of L (sets of rows per column, new numbering) by eliminating the permuted graph column by column"""
    pinv = np.empty(m, np.int64)
    pinv[perm] = np.arange(m)
    below = [set() for _ in range(m)]
    for p in range(m):
        for k in range(rowptr[p], rowptr[p + 1]):
            q = int(colidx[k])
            if q != p:
                a, b = sorted((int(pinv[p]), int(pinv[q])))
                below[a].add(b)
    for j in range(m):
        rows = sorted(below[j])
        if rows:
            par = rows[0]
            below[par].update(r for r in rows[1:])
    return below


def _check(m, rowptr, colidx, origin=None):
    r = api.symbolic_analyse(rowptr, colidx, origin)
    perm, colptr, rowidx = r["perm"], r["colptr"], r["rowidx"]
    assert sorted(perm.tolist()) == list(range(m))
    assert colptr[0] == 0 and colptr[m] == len(rowidx) == r["info"][0]
    exp = _brute_force_fill(m, rowptr, colidx, perm)
    for j in range(m):
        col = rowidx[colptr[j]:colptr[j + 1]]
        assert col[0] == j and np.all(np.diff(col) > 0), j               # diagonal first, ascending
        assert set(col[1:].tolist()) == exp[j], j                        # exactly the fill of the elimination, no more, no less
    return r


@pytest.mark.parametrize("name", ["stereo_n5.npz", "stereo_n8.npz", "mono_n5.npz", "mono_n8.npz"])
def test_symbolic_on_reference_assembled_systems(oracle, name):
    """the patterns of all 22 systems the real reference assembled (block pattern = what pba_constructAuxCSS* hands to cholmod_amd)"""
    z = load_golden(name)
    for j in range(int(z["njoins"])):
        rowptr, colidx = z[f"join{j}.parts_in.rowptr"], z[f"join{j}.parts_in.colidx"]
        m = len(rowptr) - 1
        # (a Mono join drops every block of its reference pose: the device keeps that row's diagonal block as a placeholder)
        rows = [sorted(set([p]) | set(colidx[rowptr[p]:rowptr[p + 1]].tolist())) for p in range(m)]
        rp = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int32)
        _check(m, rp, np.concatenate(rows).astype(np.int32))


@pytest.mark.parametrize("m,band,hubs,seed", [(1, 0, 0, 0), (2, 1, 0, 0), (60, 2, 3, 1), (300, 5, 6, 2), (700, 12, 12, 3)])
def test_symbolic_on_schur_like_patterns(m, band, hubs, seed):
    rowptr, colidx, _ = synth.schur_like_matrix(m, band, hubs, seed=seed)
    r = _check(m, rowptr, colidx)
    assert r["info"][1] >= 1


def test_symbolic_batched_level_and_join_tree_origins():
    """A tree level is many independent systems in one matrix; origins that are not the pose positions (poses brought by local
    maps 0..N-1, several per map, as after Mono joins) must drive the dissection: the two halves of the join tree stay
    separate sub-trees below the top separator."""
    rng = np.random.default_rng(4)
    m, seg = 240, 60
    rows = [set([p]) for p in range(m)]
    for s0 in range(0, m, seg):
        for p in range(s0, s0 + seg):
            for d in (1, 2, 3):
                if p + d < s0 + seg:
                    rows[p].add(p + d)
        h = s0 + int(rng.integers(seg))
        for p in range(s0, s0 + seg):
            rows[min(h, p)].add(max(h, p))
    rowptr, colidx = [0], []
    for p in range(m):
        colidx += sorted(rows[p])
        rowptr.append(len(colidx))
    origin = (np.arange(m) // 3).astype(np.int32)
    r = _check(m, np.array(rowptr, np.int32), np.array(colidx, np.int32), origin)
    # no fill between the independent systems
    perm, colptr, rowidx = r["perm"], r["colptr"], r["rowidx"]
    for j in range(m):
        segs = set(int(perm[i]) // seg for i in rowidx[colptr[j]:colptr[j + 1]])
        assert len(segs) == 1, j


def test_symbolic_rejects_malformed_patterns():
    with pytest.raises(api.LsfmError):
        api.symbolic_analyse(np.array([0, 1, 2], np.int32), np.array([1, 1], np.int32))   # row 0 without its diagonal block
    with pytest.raises(api.LsfmError):
        api.symbolic_analyse(np.array([0, 2, 3], np.int32), np.array([0, 5, 1], np.int32))  # column out of range


def test_symbolic_large_system_same_on_one_thread_and_many(tmp_path):
    """One large system is analysed on several host threads (separator cover by origin groups, counting sorts over key ranges, row-range
    walks for the column counts and patterns: lsfm_symbolic.cpp): every output must be what ONE thread produces.  The thread count is
    read once per process (LSFM_SYM_THREADS), so either side runs in a process of its own."""
    import hashlib
    import os
    import subprocess
    import sys
    script = tmp_path / "run.py"
    script.write_text(
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, sys.argv[1])\n"
        "from linearsfm_amd import api, synth\n"
        "rowptr, colidx, _ = synth.schur_like_matrix(9000, 20, 16, seed=5)\n"
        "assert rowptr[-1] >= 200000\n"
        "r = api.symbolic_analyse(rowptr, colidx)\n"
        "h = hashlib.sha256()\n"
        "for k in ('perm', 'colptr', 'rowidx', 'info'): h.update(np.ascontiguousarray(r[k]).tobytes())\n"
        "print(h.hexdigest(), int(r['info'][0]))\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = {}
    for label, threads in (("one", "1"), ("many", "8")):
        env = dict(os.environ, LSFM_SYM_THREADS=threads)
        out[label] = subprocess.check_output([sys.executable, str(script), root], env=env, text=True).split()
    assert out["one"] == out["many"]
    assert int(out["one"][1]) > 200000
