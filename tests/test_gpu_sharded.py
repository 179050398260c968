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


def _worker(rank, world, port, n_maps, mono, q, top="merge", backend="gloo", plans=True):
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
    st = ShardedTree(ctx, maps[lo:hi], lo, n_maps, mono, top=top, shard_single=True, comm_bytes=64 << 20)
    st.set_plans(plans)
    outs = []
    for _ in range(3):  # repeated steps reuse the resident trees, their plans and the packed buffers
        dist.barrier()
        _, rc = st.run()
        assert rc == 0
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
@pytest.mark.parametrize("world,n_maps,mono,top,backend,plans", [
    (2, 64, False, "merge", "gloo", True), (4, 100, False, "merge", "gloo", True), (2, 40, True, "merge", "gloo", True),
    (3, 21, False, "merge", "gloo", True),
    (2, 64, False, "shard", "gloo", True), (4, 100, False, "shard", "gloo", True), (2, 40, True, "shard", "gloo", True),
    (3, 21, False, "shard", "gloo", True), (4, 100, False, "shard", "gloo", False), (4, 52, True, "shard", "gloo", False),
    (4, 3, False, "shard", "gloo", True),
    # one rank, RCCL: the library's sums go through torch's all_reduce on device pointers under the library's own stream
    (1, 48, False, "shard", "nccl", True), (1, 24, True, "shard", "nccl", False)])
def test_sharded_tree_equals_single_tree(ctx, world, n_maps, mono, top, backend, plans):
    maps = _make(n_maps, mono)
    single, _, rc = ctx.divide_conquer([dict(m.__dict__) for m in maps], mono)
    assert rc == 0
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    port = _free_port()
    procs = [mpc.Process(target=_worker, args=(r, world, port, n_maps, mono, q, top, backend, plans)) for r in range(world)]
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
