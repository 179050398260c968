"""The distribution the tolerance of tests/test_gpu_sharded.py is derived from: the feature-sharded tree on `world` ranks (all on
ONE GPU, gloo) against the single tree over all maps, `runs` runs of the resident trees -- the sums over features are taken slice by
slice there, the tiles of K9 and of the transform hold other features, so the two evaluations differ by rounding that the top
systems' conditioning amplifies.  Prints per case the max, the 99.9th percentile and the median of pose_param_err / feat_param_err
(tests/common.py) over the runs.
usage: python tools/sharded_noise.py <runs> [case ...]      case = world,n_maps,mono(0|1),plans(0|1),solve   (default: the test's gloo "shard" cases)"""
import os
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)
import numpy as np
import torch.multiprocessing as mp

from linearsfm_amd import synth

DEFAULT = ["4,200,1,1,owned", "2,64,0,1,owned", "4,100,0,1,owned", "2,40,1,1,owned", "4,100,0,0,owned", "4,52,1,0,owned", "2,64,0,1,replicated",
           "4,52,1,0,replicated", "3,21,0,1,owned"]


def make(n_maps, mono):  # (tests/test_gpu_sharded.py _make)
    return synth.make_mono_set(n_maps, 8, 4, seed=41, **synth.SPIRAL) if mono else synth.make_stereo_set(n_maps, 8, 5, seed=41, lap=30, home=5)


def worker(rank, world, port, n_maps, mono, plans, solve, runs, q):
    os.environ["LSFM_TASK_X"] = "12"  # (as the test: leaf sub-trees small enough for these sets to have inter-block separators)
    import torch
    import torch.distributed as dist
    from linearsfm_amd import api
    from linearsfm_amd.distributed import ShardedTree, shard_bounds
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    maps = make(n_maps, mono)
    _, bounds = shard_bounds(n_maps, world)
    lo, hi = bounds[rank]
    ctx = api.Context(0)
    st = ShardedTree(ctx, maps[lo:hi], lo, n_maps, mono, top="shard", shard_single=True, comm_bytes=64 << 20, solve=solve)
    st.set_plans(plans)
    outs = []
    for _ in range(runs):
        dist.barrier()
        _, rc = st.run()
        assert rc == 0
        o = st.download(full=False)
        if rank == 0:
            outs.append(np.asarray(o["stVal"]).copy())
    if rank == 0:
        q.put(outs)
    dist.barrier()
    st.close()
    ctx.close()
    dist.destroy_process_group()


def main():
    from common import feat_param_err, pose_param_err
    from linearsfm_amd import api
    runs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    cases = sys.argv[2:] or DEFAULT
    import torch
    torch.cuda.init()
    ctx = api.Context(0)
    for case in cases:
        world, n_maps, mono, plans, solve = case.split(",")
        world, n_maps, mono, plans = int(world), int(n_maps), bool(int(mono)), bool(int(plans))
        maps = make(n_maps, mono)
        singles = []
        for _ in range(8):  # the single tree is not the same bits from run to run either: its own spread, for scale
            single, _, rc = ctx.divide_conquer([dict(m.__dict__) for m in maps], mono)
            assert rc == 0
            singles.append(single)
        single = singles[0]
        own = [max(pose_param_err(s["stVal"], single["stVal"], single["stno"]), feat_param_err(s["stVal"], single["stVal"], single["stno"])) for s in singles[1:]]
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        mpc = mp.get_context("spawn")
        q = mpc.Queue()
        procs = [mpc.Process(target=worker, args=(r, world, port, n_maps, mono, plans, solve, runs, q)) for r in range(world)]
        for p in procs:
            p.start()
        outs = q.get(timeout=3600)
        for p in procs:
            p.join(timeout=300)
        pe = np.array([pose_param_err(o, single["stVal"], single["stno"]) for o in outs])
        fe = np.array([feat_param_err(o, single["stVal"], single["stno"]) for o in outs])
        distinct = len(set(o.tobytes() for o in outs))
        print("case world=%d maps=%d %s plans=%d solve=%s runs=%d | pose err max %.3e p99.9 %.3e median %.3e | feature err max %.3e p99.9 %.3e median %.3e | "
              "distinct sharded states %d | single tree vs itself (7 runs) max %.3e"
              % (world, n_maps, "Mono" if mono else "Stereo", plans, solve, len(outs), pe.max(), np.percentile(pe, 99.9), np.median(pe), fe.max(),
                 np.percentile(fe, 99.9), np.median(fe), distinct, max(own) if own else 0.0), flush=True)
    ctx.close()


if __name__ == "__main__":
    main()
