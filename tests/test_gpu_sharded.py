"""The device-resident multi-GPU scheduler (linearsfm_amd.distributed.ShardedTree) on ONE GPU: two / four processes share
cuda:0, the packed sub-tree roots travel through torch.distributed (gloo here: two ranks of one RCCL communicator cannot
sit on the same device; with the nccl backend the same code sends the device buffers directly).  The result must equal the
single tree over all maps."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from common import feat_param_err, pose_param_err
from linearsfm_amd import synth

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make(n_maps, mono):
    return synth.make_mono_set(n_maps, 8, 4, seed=41, **synth.SPIRAL) if mono else synth.make_stereo_set(n_maps, 8, 5, seed=41, lap=30, home=5)


def _worker(rank, world, port, n_maps, mono, q, top="merge", backend="gloo", plans=True, solve="owned"):
    # (leaf sub-trees of at most 12 blocks instead of 90, so that sets this small have supernode groups and inter-block separators
    # above them: the distributed factorisation is then what runs, not its replicated fallback; read once, when the library loads)
    os.environ["LSFM_TASK_X"] = "12"
    import torch
    import torch.distributed as dist
    from linearsfm_amd import api
    from linearsfm_amd.distributed import ShardedTree, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    maps = _make(n_maps, mono)
    _, bounds = shard_bounds(n_maps, world)
    lo, hi = bounds[rank]
    ctx = api.Context(0)
    st = ShardedTree(ctx, maps[lo:hi], lo, n_maps, mono, top=top, shard_single=True, comm_bytes=64 << 20, solve=solve)
    st.set_plans(plans)
    outs = []
    for _ in range(3):  # repeated steps reuse the resident trees, their plans and the packed buffers
        dist.barrier()
        stats, rc = st.run()
        assert rc == 0
        if top == "shard" and world > 1 and n_maps >= 16:
            # the camera systems of the top levels were factored by block ownership (or, asked so, by every rank in full)
            assert (stats["dist_solves"] > 0) == (solve == "owned"), (solve, stats["dist_solves"])
        if top == "shard":
            o = st.download()  # collective: every rank hands its feature slice to rank 0
            if rank == 0:
                outs.append(o)
        elif rank == 0:
            outs.append(st.download())
    if rank == 0:
        q.put([{k: o[k] for k in ("stno", "stVal", "Ui", "Uj", "photo", "feature", "Ref", "FRef")} for o in outs])
    dist.barrier()
    st.close()
    ctx.close()
    dist.destroy_process_group()


# top = "merge": sub-tree sharding with pairwise merge rounds; "shard": the levels above the blocks feature-sharded over ALL ranks
# (three all-reduces per level through the library's lsfm_allreduce_fn hook; without plans also the union of the ranks' patterns)
# solve = "owned": the camera systems of the feature-sharded levels are factored by block ownership (rank r the columns of block r's
# poses, one exact integer all-reduce of the inter-block separators' accumulators, the separators by everybody); "replicated": by
# every rank in full (round 3)
@pytest.mark.parametrize("world,n_maps,mono,top,backend,plans,solve", [
    (2, 64, False, "merge", "gloo", True, "owned"), (4, 100, False, "merge", "gloo", True, "owned"), (2, 40, True, "merge", "gloo", True, "owned"),
    (3, 21, False, "merge", "gloo", True, "owned"),
    (2, 64, False, "shard", "gloo", True, "owned"), (4, 100, False, "shard", "gloo", True, "owned"), (2, 40, True, "shard", "gloo", True, "owned"),
    (3, 21, False, "shard", "gloo", True, "owned"), (4, 100, False, "shard", "gloo", False, "owned"), (4, 52, True, "shard", "gloo", False, "owned"),
    (4, 3, False, "shard", "gloo", True, "owned"), (4, 200, True, "shard", "gloo", True, "owned"),
    (2, 64, False, "shard", "gloo", True, "replicated"), (4, 52, True, "shard", "gloo", False, "replicated"),
    # one rank, RCCL: the library's sums go through torch's all_reduce on device pointers under the library's own stream
    (1, 48, False, "shard", "nccl", True, "owned"), (1, 24, True, "shard", "nccl", False, "owned")])
def test_sharded_tree_equals_single_tree(ctx, world, n_maps, mono, top, backend, plans, solve):
    maps = _make(n_maps, mono)
    single, _, rc = ctx.divide_conquer([dict(m.__dict__) for m in maps], mono)
    assert rc == 0
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, n_maps, mono, q, top, backend, plans, solve)) for r in range(world)]
    for p in procs:
        p.start()
    outs = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    for got in outs:
        assert np.array_equal(got["stno"], single["stno"])
        for k in ("Ui", "Uj", "photo", "feature"):
            assert np.array_equal(got[k], single[k]), k
        assert got["Ref"] == single["Ref"] and got["FRef"] == single["FRef"]
        # same tree shape, same joins; the elimination order of a merged system and the summation order of atomics may differ
        # (feature-sharded: the sums over features are taken slice by slice)
        tol = 1e-8 if mono else 1e-9
        assert pose_param_err(got["stVal"], single["stVal"], single["stno"]) < tol
        assert feat_param_err(got["stVal"], single["stVal"], single["stno"]) < tol


@pytest.mark.parametrize("mono,n_maps,nslices", [(False, 24, 3), (True, 12, 4), (False, 5, 1)])
def test_device_slice_packs_equal_the_numpy_slices(ctx, mono, n_maps, nslices):
    """lsfm_tree_export_slice_dev against linearsfm_amd.distributed.slice_map: the final map of a tree cut by feature label on the
    device, every pack uploaded again as a one-map tree and downloaded, must be the slice numpy cuts from the downloaded map --
    labels, state, V, W blocks and their order, run pointers -- and the slices must add up to the map."""
    import torch
    from linearsfm_amd.distributed import merge_slices, slice_map
    maps = _make(n_maps, mono)
    t = ctx.tree_upload([dict(m.__dict__) for m in maps], mono)
    packs = []
    try:
        _, rc = ctx.tree_run(t)
        assert rc == 0
        m, n, stno, stVal = ctx.tree_download_state(t)
        full = ctx.tree_download(t)
        assert (m, n) == (full["m"], full["n"]) and np.array_equal(stno, full["stno"]) and np.array_equal(stVal, full["stVal"])
        sizes = ctx.tree_export_slice_sizes(t, nslices)
        for g in range(nslices):
            buf = torch.empty(sizes[g], dtype=torch.uint8, device="cuda")
            ctx.tree_export_slice_dev(t, nslices, g, buf.data_ptr(), sizes[g])
            packs.append(buf)
    finally:
        ctx.tree_free(t)
    got = []
    for g, buf in enumerate(packs):
        ts = ctx.tree_upload_dev([buf.data_ptr()], mono)
        try:
            ctx.tree_set_final_reanchor(ts, False)
            _, rc = ctx.tree_run(ts)  # one map, nothing to join: the resident input is the result
            assert rc == 0
            d = ctx.tree_download(ts)
        finally:
            ctx.tree_free(ts)
        exp = slice_map(full, nslices, g)
        assert d["m"] == exp["m"] and d["n"] == exp["n"] and d["Ref"] == full["Ref"] and d["FRef"] == full["FRef"]
        for k in ("stno", "stVal", "Ui", "Uj", "U", "V", "W", "photo", "feature", "FBlock"):
            assert np.array_equal(np.asarray(d[k]).reshape(-1), np.asarray(exp[k]).reshape(-1)), (g, k)
        got.append(d)
    order = np.asarray(full["stno"])[6 * full["m"]::3]
    back = merge_slices(got, order)
    for k in ("stno", "stVal", "V", "W", "photo", "feature", "FBlock"):
        assert np.array_equal(np.asarray(back[k]).reshape(-1), np.asarray(full[k]).reshape(-1)), k


def test_comm_buffer_too_small_is_reported(ctx):
    """The arrays the library sums over the ranks live in the caller's buffer: one that cannot hold a camera system is an
    error of the call (LSFM_ERR_ARG with the size needed), not a write past its end."""
    from linearsfm_amd import api
    from linearsfm_amd.distributed import ShardedTree
    maps = _make(12, False)
    st = ShardedTree(ctx, maps, 0, len(maps), False, top="shard", shard_single=True, comm_bytes=4096)
    try:
        with pytest.raises(api.LsfmError, match="too small"):
            st.run()
    finally:
        st.close()
    # the context is usable afterwards
    out, _, rc = ctx.divide_conquer([dict(m.__dict__) for m in maps], False)
    assert rc == 0 and out["m"] == 12
