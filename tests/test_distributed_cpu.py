"""world_size-2 / 3 / 4 gloo tests of the multi-GPU schedule (linearsfm_amd/distributed.py: blocks = sub-trees, pairwise
merge rounds on log2(G) ranks) on CPU.  The compute back end here is the oracle (tests only); on the GPU box the same
schedule (merge_schedule) drives the HIP library with device-resident hand-off (ShardedTree, tests/test_gpu_sharded.py)."""
import os
import socket

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

from linearsfm_amd import synth
from linearsfm_amd.distributed import merge_schedule, shard_bounds
from sharded_reference import sharded_divide_conquer


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_run_tree(maps, mono, final_reanchor):
    from oracle import pyoracle as po
    dicts = [po.localmap_to_dict(m) if not isinstance(m, dict) else m for m in maps]
    out, _, rc = po.divide_conquer(dicts, mono, final_reanchor=final_reanchor)
    assert rc == 0
    return out


def _make(n_maps, mono):
    return synth.make_mono_set(n_maps, 6, 4, seed=21) if mono else synth.make_stereo_set(n_maps, 5, 4, seed=21)


def _worker(rank, world, port, n_maps, mono, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = sharded_divide_conquer(_make(n_maps, mono), mono, _oracle_run_tree)
    if rank == 0:
        q.put({k: out[k] for k in ("stno", "stVal", "Ui", "Uj", "photo", "feature", "Ref", "FRef")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_maps,mono", [(2, 8, False), (2, 7, False), (3, 11, False), (2, 6, True), (4, 13, False), (4, 16, True),
                                               (4, 5, False)])
def test_subtree_sharding_equals_serial_tree(oracle, world, n_maps, mono):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_maps, mono, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = _oracle_run_tree(_make(n_maps, mono), mono, True)
    # identical tree shape => identical arithmetic: bitwise equal to the single-process order
    for k in ("stno", "Ui", "Uj", "photo", "feature"):
        assert np.array_equal(got[k], exp[k]), k
    assert got["Ref"] == exp["Ref"] and got["FRef"] == exp["FRef"]
    assert np.array_equal(got["stVal"], exp["stVal"])


def test_shard_bounds_are_subtrees():
    for n, w in ((3499, 8), (16, 4), (5, 2), (7, 3), (1, 2)):
        size, b = shard_bounds(n, w)
        assert size & (size - 1) == 0 and size * w >= n
        assert b[0][0] == 0 and max(h for _, h in b) == n
        assert all(lo % size == 0 or lo == n for lo, _ in b)


def test_merge_schedule_is_a_binary_tree():
    """Every rank but 0 sends exactly once, to the rank that merges it in that round; the last merge is on rank 0 and is
    re-anchored; the re-anchoring of an intermediate node follows the parity of its index at its level."""
    for world in (1, 2, 3, 4, 5, 8):
        nonempty = [True] * world
        sends, merges = {}, {}
        for r in range(world):
            for act in merge_schedule(r, world, nonempty):
                (sends if act[0] == "send" else merges).setdefault(r, []).append(act)
        assert sorted(sends) == list(range(1, world))
        for r, (a,) in sends.items():
            _, peer, j = a
            assert any(m[1] == r and m[2] == j for m in merges[peer])
        if world > 1:
            last = merges[0][-1]
            assert last[3] is True
            for r, acts in merges.items():
                for _, peer, j, reanchor in acts[:-1] if r == 0 else acts:
                    assert reanchor == ((r >> (j + 1)) % 2 == 1)


# ---- feature-sharded top levels (linearsfm_amd.distributed "P2") with the oracle as back end --------------------------------
def _oracle_run_slices(slices, mono, rank, world):
    from oracle import pyoracle as po
    fn = po.torch_reduce_fn()
    out, _, rc = po.divide_conquer(slices, mono, final_reanchor=True, comm=(rank, world, fn))
    assert rc == 0
    return out


def _worker_top(rank, world, port, n_maps, mono, q):
    from sharded_reference import sharded_divide_conquer_top
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    out = sharded_divide_conquer_top(_make(n_maps, mono), mono, _oracle_run_tree, _oracle_run_slices)
    if rank == 0:
        q.put({k: out[k] for k in ("stno", "stVal", "Ui", "Uj", "photo", "feature", "Ref", "FRef", "U", "W", "V")})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_maps,mono", [(2, 8, False), (2, 7, False), (3, 11, False), (2, 6, True), (4, 13, False), (4, 16, True),
                                               (4, 5, False)])
def test_feature_sharded_top_levels_equal_serial_tree(oracle, world, n_maps, mono):
    """Every rank joins its block, then ALL ranks evaluate the levels above on their slice of the features (label mod world); the
    sums over features that enter pose-side quantities (hub rows of the transformed U, S and E, the pattern of S) cross the
    ranks through gloo all-reduces inside the oracle.  Same structure as the serial tree; values equal up to the order of those
    sums."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_top, args=(r, world, port, n_maps, mono, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp = _oracle_run_tree(_make(n_maps, mono), mono, True)
    for k in ("stno", "Ui", "Uj", "photo", "feature"):
        assert np.array_equal(got[k], exp[k]), k
    assert got["Ref"] == exp["Ref"] and got["FRef"] == exp["FRef"]
    scale = np.maximum(1.0, np.abs(exp["stVal"]))
    assert np.max(np.abs(got["stVal"] - exp["stVal"]) / scale) < 1e-9
    for k in ("U", "W", "V"):  # the information matrix of the final map, put together from the slices
        a, b = np.asarray(got[k]).reshape(-1), np.asarray(exp[k]).reshape(-1)
        assert a.shape == b.shape and np.max(np.abs(a - b)) <= 1e-9 * max(1.0, np.max(np.abs(b))), k


def test_slices_of_a_map_add_up_to_the_map():
    from linearsfm_amd.distributed import joint_feature_order, merge_slices
    from sharded_reference import slice_map
    from oracle import pyoracle as po
    maps = [po.localmap_to_dict(m) for m in _make(6, False)]
    node = _oracle_run_tree(maps[:4], False, True)
    order = np.asarray(node["stno"])[6 * node["m"]::3]
    for G in (1, 2, 3, 5):
        parts = [slice_map(node, G, g) for g in range(G)]
        assert sum(p["n"] for p in parts) == node["n"]
        back = merge_slices(parts, order)
        for k in ("stno", "stVal", "V", "W", "photo", "feature", "FBlock"):
            assert np.array_equal(np.asarray(back[k]).reshape(-1), np.asarray(node[k]).reshape(-1)), (G, k)
    # the order of a joint map's features: End's, then Cur's unmatched ones
    a = np.asarray(maps[0]["stno"])[6 * maps[0]["m"]::3]
    b = np.asarray(maps[1]["stno"])[6 * maps[1]["m"]::3]
    j = _oracle_run_tree(maps[:2], False, False)
    assert np.array_equal(joint_feature_order([a, b]), np.asarray(j["stno"])[6 * j["m"]::3])
