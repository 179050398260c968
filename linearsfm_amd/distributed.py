"""Multi-GPU scheduling of ONE join tree: one process per GPU, torch.distributed for the hand-off of sub-tree roots.

The reference pairs maps (2i, 2i+1) level by level (LinearSFMImp.cpp:1932-2035), so the tree node of level k with index r
covers the local maps [r*2^k, (r+1)*2^k): a block of 2^k consecutive local maps IS a subtree.  With G ranks:

  1. rank r joins block r on its own GPU, no communication (the independent joins of the lower levels: the only
     parallelism the reference's loop offers, LinearSFMImp.cpp:1938-2033);
  2. log2(G) merge rounds: in round j the node held by rank a (a multiple of 2^(j+1)) is joined with the node of rank
     a + 2^j, on rank a.  The partner's node travels as ONE packed device buffer (lsfm_tree_export_dev ->
     torch.distributed send/recv: RCCL over xGMI with the nccl backend -> lsfm_tree_upload_dev / lsfm_tree_reload_dev);
     no host copy of the arrays, no pickling.  Half of the remaining ranks drop out every round; the last join runs on
     rank 0.

An odd-indexed node of a level is taken back to its first frame, an even-indexed one is left in the frame of its last
join, exactly as the reference's loop treats an intermediate node ((i+1)%2 == 0, LinearSFMImp.cpp:1997-2025); the final
map goes back to the first frame of the whole set (2039-2063).  The tree shape -- and therefore every transform and
join -- is identical to the single-process order.

`merge_schedule` is the schedule itself (pure Python, shared by both back ends); `ShardedTree` drives the HIP library;
`sharded_divide_conquer` runs the same schedule with a caller-supplied CPU back end (the tests pass the oracle) so that
the scheduling is covered by world-size-2/3/4 gloo tests without a GPU.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.distributed as dist


def shard_bounds(n_maps, world):
    """Block size 2^k with 2^k * world >= n_maps, and the [lo, hi) range of every rank (trailing ranks may be empty)."""
    k = 0
    while (1 << k) * world < n_maps:
        k += 1
    size = 1 << k
    return size, [(min(r * size, n_maps), min((r + 1) * size, n_maps)) for r in range(world)]


def merge_schedule(rank, world, nonempty):
    """What `rank` does after its own block, round by round.  nonempty[r]: block r holds maps (blocks fill from the left).
    Yields ("send", peer, round) -- hand the node to `peer` and stop -- or ("merge", peer_or_None, round, reanchor) -- join
    the own node (first) with the node received from `peer` (None: nothing to join, the node is carried), then take the
    result back to its first frame if `reanchor`."""
    rounds = 0
    while (1 << rounds) < world:
        rounds += 1
    for j in range(rounds):
        stride = 1 << j
        if rank % (2 * stride) == stride:
            yield ("send", rank - stride, j)
            return
        peer = rank + stride
        have = peer < world and nonempty[peer]
        last = j == rounds - 1
        # index of the joined node at its level decides whether it goes back to its first frame (odd: yes); the root always does
        reanchor = True if last else ((rank >> (j + 1)) % 2 == 1)
        yield ("merge", peer if have else None, j, reanchor)


def first_reanchor(rank, world):
    """Whether the root of block `rank` is re-anchored by the block's own run: odd-indexed nodes are; a lone rank owns the
    whole tree, whose root always is."""
    return True if world == 1 else rank % 2 == 1


# ---------------------------------------------------------------------------------------------------------------
# CPU back end (tests: the oracle) -- nodes are map dicts, moved as Python objects
# ---------------------------------------------------------------------------------------------------------------
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


def hip_run_tree(ctx):
    """`run_tree` back end over the HIP library for sharded_divide_conquer (host hand-off; ShardedTree is the device-resident
    scheduler): raises on a failure, refuses a result whose systems did not all converge."""
    def run(parts, mono, final_reanchor):
        out, stats, rc = ctx.divide_conquer(parts, mono, final_reanchor=final_reanchor)
        if rc != 0:
            raise RuntimeError(f"lsfm_divide_conquer: {stats['not_converged']} system(s) not converged "
                               f"(max relative residual {stats['max_rel_residual']:.3e})")
        return out
    return run


# ---------------------------------------------------------------------------------------------------------------
# HIP back end: nodes stay on the devices
# ---------------------------------------------------------------------------------------------------------------
class ShardedTree:
    """One join tree over all ranks of `group`.  Every rank uploads ITS block of local maps once (resident in HBM);
    run() joins the whole tree and leaves the final map on rank 0 (download()).  run() can be repeated: the block trees
    and the merge trees keep their allocations and their plans, the packed buffers are reused."""

    def __init__(self, ctx, maps_block, lo, n_total, mono, group=None, device=None):
        self.ctx, self.mono, self.group = ctx, bool(mono), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        _, self.bounds = shard_bounds(n_total, self.world)
        assert self.bounds[self.rank][0] == lo and self.bounds[self.rank][1] - lo == len(maps_block), \
            "maps_block must be exactly this rank's block (shard_bounds)"
        self.nonempty = [hi > a for a, hi in self.bounds]
        self.block_tree = None
        if maps_block:
            block = []
            for k, m in enumerate(maps_block):
                d = dict(m if isinstance(m, dict) else m.__dict__)
                d.setdefault("pose_origin", np.full(int(d["m"]), lo + k, np.int32))
                block.append(d)
            self.block_tree = ctx.tree_upload(block, self.mono)
            ctx.tree_set_final_reanchor(self.block_tree, first_reanchor(self.rank, self.world))
        self.merge_trees = {}   # round -> tree built from packed buffers
        self.out_bufs = {}      # round -> torch buffer the own node is packed into
        self.in_bufs = {}       # round -> torch buffer the partner's node arrives in
        self.in_host = {}       # round -> pinned host landing buffer (gloo: the payload is staged through the host)
        self.pending = {}       # round -> receives posted at the start of a run
        self.result = None
        self.plans = True
        self.gpu_direct = dist.is_initialized() and dist.get_backend(group) == "nccl"

    def set_plans(self, on):
        """on = False: every run of every tree of this rank analyses its structure again (what the reference's timed region
        contains: symbolic work in every join); True (default): later runs reuse what the first one recorded."""
        self.plans = bool(on)
        for t in list(self.merge_trees.values()) + ([self.block_tree] if self.block_tree is not None else []):
            self.ctx.tree_set_plans(t, self.plans)

    # -- transport of one packed node -------------------------------------------------------------------------------
    def _send(self, buf, dst):
        size = torch.tensor([buf.numel()], dtype=torch.int64, device=self.device if self.gpu_direct else "cpu")
        dist.send(size, dst=dst, group=self.group)
        dist.send(buf if self.gpu_direct else buf.cpu(), dst=dst, group=self.group)
        if self.gpu_direct:
            # the next run() packs into the same buffer on the library's own stream, which knows nothing of RCCL's: wait here
            torch.cuda.current_stream(self.device).synchronize()

    def _post_recv(self, src, slot):
        """Posts the receives of the node `src` will send in round `slot` NOW, into the buffer of the last run (same structure,
        same size: a resident tree is joined again and again), so that the transfer runs while this rank is still busy with
        its own block or its export.  First run of a slot: nothing is known yet, _recv does it all."""
        buf = self.in_bufs.get(slot)
        if buf is None:
            return
        size = torch.zeros(1, dtype=torch.int64, device=self.device if self.gpu_direct else "cpu")
        land = buf if self.gpu_direct else self.in_host.setdefault(slot, torch.empty(buf.numel(), dtype=torch.uint8).pin_memory())
        self.pending[slot] = (size, land, dist.irecv(size, src=src, group=self.group), dist.irecv(land, src=src, group=self.group))

    def _recv(self, src, slot):
        posted = self.pending.pop(slot, None)
        if posted is not None:
            size, land, w0, w1 = posted
            w0.wait()
            w1.wait()
            buf = self.in_bufs[slot]
            if int(size.item()) != buf.numel():
                raise RuntimeError("a sub-tree root changed its size between two runs of a resident tree")
            if not self.gpu_direct:
                buf.copy_(land, non_blocking=True)
            torch.cuda.current_stream(self.device).synchronize()  # the library reads the buffer on its own stream next
            return buf
        size = torch.zeros(1, dtype=torch.int64, device=self.device if self.gpu_direct else "cpu")
        dist.recv(size, src=src, group=self.group)
        n = int(size.item())
        buf = self.in_bufs.get(slot)
        if buf is None or buf.numel() != n:
            buf = self.in_bufs[slot] = torch.empty(n, dtype=torch.uint8, device=self.device)
        if self.gpu_direct:
            dist.recv(buf, src=src, group=self.group)
            torch.cuda.current_stream(self.device).synchronize()  # the library reads it on its own stream next
        else:
            host = torch.empty(n, dtype=torch.uint8)
            dist.recv(host, src=src, group=self.group)
            buf.copy_(host)
            torch.cuda.synchronize(self.device)
        return buf

    def _export(self, tree, slot):
        n = self.ctx.tree_export_size(tree)
        if n == 0:
            raise RuntimeError("tree has no exportable result")
        buf = self.out_bufs.get(slot)
        if buf is None or buf.numel() != n:
            buf = self.out_bufs[slot] = torch.empty(n, dtype=torch.uint8, device=self.device)
        self.ctx.tree_export_dev(tree, buf.data_ptr(), n)  # synchronises the library's stream
        return buf

    # -- one whole tree ---------------------------------------------------------------------------------------------
    def run(self):
        """Returns (stats of the last tree run on this rank or None, worst return code on this rank)."""
        ctx = self.ctx
        cur, stats, worst = None, None, 0
        # the partners' nodes may start travelling as soon as they exist: post every receive of this run whose buffer is known
        for act in merge_schedule(self.rank, self.world, self.nonempty):
            if act[0] == "merge" and act[1] is not None:
                self._post_recv(act[1], act[2])
        if self.block_tree is not None:
            stats, rc = ctx.tree_run(self.block_tree)
            worst = max(worst, rc)
            cur = self.block_tree
        self.result = None
        for act in merge_schedule(self.rank, self.world, self.nonempty):
            if act[0] == "send":
                _, peer, j = act
                if cur is not None:
                    self._send(self._export(cur, ("send", j)), peer)
                return stats, worst
            _, peer, j, reanchor = act
            parts = []
            if cur is not None:
                parts.append(self._export(cur, ("own", j)))
            if peer is not None:
                parts.append(self._recv(peer, j))
            if not parts:
                cur = None
                continue
            ptrs = [b.data_ptr() for b in parts]
            mt = self.merge_trees.get(j)
            if mt is None:
                mt = self.merge_trees[j] = ctx.tree_upload_dev(ptrs, self.mono)
                ctx.tree_set_plans(mt, self.plans)
            else:
                ctx.tree_reload_dev(mt, ptrs)
            ctx.tree_set_final_reanchor(mt, reanchor)
            stats, rc = ctx.tree_run(mt)
            worst = max(worst, rc)
            cur = mt
        if self.rank == 0:
            self.result = cur
        return stats, worst

    def download(self):
        """The final map (rank 0, after run())."""
        if self.result is None:
            raise RuntimeError("no result on this rank (rank 0 holds it after run())")
        return self.ctx.tree_download(self.result)

    def close(self):
        for t in list(self.merge_trees.values()) + ([self.block_tree] if self.block_tree is not None else []):
            self.ctx.tree_free(t)
        self.merge_trees, self.block_tree, self.result = {}, None, None
