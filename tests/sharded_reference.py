"""TEST SCAFFOLDING (moved out of linearsfm_amd/distributed.py in round 6): the multi-rank SCHEDULE of the sharded tree on a caller-supplied
CPU back end -- the tests pass the oracle -- with nodes moved as Python objects over torch.distributed (gloo), and the numpy slicer the
device's slice packs are checked against.  Nothing of the product imports this module."""
import numpy as np
import torch.distributed as dist

from linearsfm_amd.distributed import first_reanchor, joint_feature_order, merge_schedule, merge_slices, shard_bounds


def sharded_divide_conquer(maps, mono, run_tree, group=None):
    """maps: the FULL list of local maps (every rank passes the same list or at least its own slice filled in).
    run_tree(list of map dicts, mono, final_reanchor) -> map dict.  Returns the final map on rank 0, None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    _, bounds = shard_bounds(len(maps), world)
    nonempty = [hi > lo for lo, hi in bounds]
    lo, hi = bounds[rank]
    node = None
    if hi > lo:
        block = []
        for k, m in enumerate(maps[lo:hi]):
            d = dict(m if isinstance(m, dict) else m.__dict__)
            d.setdefault("pose_origin", np.full(int(d["m"]), lo + k, np.int32))  # local map index inside the WHOLE tree
            block.append(d)
        node = run_tree(block, mono, first_reanchor(rank, world))
    for act in merge_schedule(rank, world, nonempty):
        if act[0] == "send":
            if node is not None:  # an empty block has nothing to hand over, and its partner does not wait for it
                dist.send_object_list([node], dst=act[1], group=group)
            return None
        _, peer, _, reanchor = act
        parts = [node] if node is not None else []
        if peer is not None:
            box = [None]
            dist.recv_object_list(box, src=peer, group=group)
            parts.append(box[0])
        if parts:
            node = run_tree(parts, mono, reanchor)
    return node if rank == 0 else None


def sharded_divide_conquer_top(maps, mono, run_tree, run_slices, group=None):
    """The same tree with FEATURE-SHARDED top levels, on a caller-supplied CPU back end (the tests pass the oracle): rank r joins
    block r (run_tree), cuts the root into `world` slices by feature label, takes slice `rank` of every block and evaluates the
    top levels on them together with the other ranks -- run_slices(list of slice dicts, mono, rank, world) -> slice dict of the
    final map, with the sums over features taken across the ranks inside.  Returns the final map on rank 0, None elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    _, bounds = shard_bounds(len(maps), world)
    nonempty = [hi > lo for lo, hi in bounds]
    lo, hi = bounds[rank]
    slices, ids = None, None
    if hi > lo:
        block = []
        for k, m in enumerate(maps[lo:hi]):
            d = dict(m if isinstance(m, dict) else m.__dict__)
            d.setdefault("pose_origin", np.full(int(d["m"]), lo + k, np.int32))
            block.append(d)
        node = run_tree(block, mono, first_reanchor(rank, world))
        ids = np.asarray(node["stno"])[6 * int(node["m"])::3].copy()
        slices = [slice_map(node, world, g) for g in range(world)]
    everything = [None] * world
    dist.all_gather_object(everything, (slices, ids), group=group)  # (a test-sized all-to-all: every rank picks its column)
    mine = [everything[b][0][rank] for b in range(world) if nonempty[b]]
    piece = run_slices(mine, mono, rank, world)
    pieces = [None] * world if rank == 0 else None
    dist.gather_object(piece, pieces, dst=0, group=group)
    if rank != 0:
        return None
    return merge_slices(pieces, joint_feature_order([everything[b][1] for b in range(world) if nonempty[b]]))


def slice_map(d, nslices, g):
    """Slice g of a map dict: all poses and U blocks, the features with label % nslices == g in their order with V and W."""
    m, n = int(d["m"]), int(d["n"])
    stno, stVal = np.asarray(d["stno"]), np.asarray(d["stVal"])
    fid = stno[6 * m::3]
    keep = np.nonzero(fid % nslices == g)[0]
    newidx = np.full(n, -1, np.int64)
    newidx[keep] = np.arange(len(keep))
    feature = np.asarray(d["feature"])
    wsel = np.nonzero(newidx[feature] >= 0)[0] if len(feature) else np.zeros(0, np.int64)
    out = dict(d)
    out["n"] = len(keep)
    sel = (6 * m + 3 * keep[:, None] + np.arange(3)[None, :]).reshape(-1)
    out["stno"] = np.concatenate([stno[:6 * m], stno[sel]]).astype(np.int32)
    out["stVal"] = np.concatenate([stVal[:6 * m], stVal[sel]])
    out["V"] = np.asarray(d["V"]).reshape(-1, 9)[keep]
    out["W"] = np.asarray(d["W"]).reshape(-1, 18)[wsel]
    out["photo"] = np.asarray(d["photo"])[wsel].astype(np.int32)
    out["feature"] = newidx[feature[wsel]].astype(np.int32)
    out["nW"] = len(wsel)
    fb = np.zeros(len(keep), np.int32)
    if len(wsel):
        cnt = np.bincount(out["feature"], minlength=len(keep))
        fb = (np.cumsum(cnt) - cnt).astype(np.int32)
    out["FBlock"] = fb
    return out
