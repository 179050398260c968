"""Multi-GPU scheduling of the join tree: one process per GPU, torch.distributed for the hand-off of sub-tree roots.

P1, subtree sharding.  The reference pairs maps (2i, 2i+1) level by level (LinearSFMImp.cpp:1932-2035), so the tree
node of level k with index r covers the local maps [r*2^k, (r+1)*2^k): a block of 2^k consecutive local maps IS a
subtree.  Rank r joins block r with no communication (an even-indexed root is left in the frame of its last join, an odd-indexed
one is taken back to its first frame, exactly what the reference's loop does to an intermediate node), the roots are gathered and the remaining log2(G) levels run on
rank 0.  The tree shape -- and therefore every transform and join -- is identical to the single-process order.

`run_tree(maps, mono, final_reanchor) -> map dict` is the compute back end: `Context.divide_conquer` of the HIP library
in production; the CPU tests pass the oracle (tests only) to check the scheduling with the gloo backend.
"""
from __future__ import annotations

import torch.distributed as dist


def shard_bounds(n_maps, world):
    """Block size 2^k with 2^k * world >= n_maps, and the [lo, hi) range of every rank (trailing ranks may be empty)."""
    k = 0
    while (1 << k) * world < n_maps:
        k += 1
    size = 1 << k
    return size, [(min(r * size, n_maps), min((r + 1) * size, n_maps)) for r in range(world)]


def sharded_divide_conquer(maps, mono, run_tree, group=None):
    """maps: the FULL list of local maps (every rank passes the same list or at least its own slice filled in).
    Returns the final map on rank 0, None elsewhere."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if world == 1:
        return run_tree(maps, mono, True)
    _, bounds = shard_bounds(len(maps), world)
    lo, hi = bounds[rank]
    # the reference re-expresses a node in its first frame when it is produced with an odd index at its level
    # ((i+1)%2 == 0, LinearSFMImp.cpp:1997-2025): block r is node r of its level
    root = run_tree(maps[lo:hi], mono, rank % 2 == 1) if hi > lo else None
    roots = [None] * world if rank == 0 else None
    dist.gather_object(root, roots, dst=0, group=group)
    if rank != 0:
        return None
    roots = [r for r in roots if r is not None]
    return run_tree(roots, mono, True)
