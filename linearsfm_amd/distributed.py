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

That is sub-tree sharding ("P1"); its merge rounds leave half of the remaining GPUs idle each.  `top="shard"` (the default of
ShardedTree with more than one rank) replaces them by FEATURE-SHARDED top levels ("P2"): after its block, every rank cuts its
sub-tree root into G slices by feature label (feat_id % G; all poses and U blocks in every slice), slice g of every block goes
to rank g (one all-to-all of packed device buffers), and ALL ranks run the top log2(G) levels together, each on its slice of
the features -- the Schur loop (Imp.cpp:2244-2332), the feature part of the transform (1270-1917) and the back-substitution
(2980-3020) are sums over features.  Three all-reduces per level cross the GPUs (RCCL over xGMI with the nccl backend): the
pose rows of the transform's I C, the camera system (S, E) before it is factored, and the pose solution of rank 0 (the
replicated factorisations may differ in the last bit); see include/lsfm.h, lsfm_tree_set_comm.

`merge_schedule` is the schedule itself (pure Python, shared by both back ends); `ShardedTree` drives the HIP library;
`sharded_divide_conquer` runs the same schedule with a caller-supplied CPU back end (the tests pass the oracle) so that
the scheduling is covered by world-size-2/3/4 gloo tests without a GPU.
"""
from __future__ import annotations

import datetime

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
# feature-sharded top levels: the order and the assembly of the final map from its slices (ShardedTree.download)
# ---------------------------------------------------------------------------------------------------------------
def joint_feature_order(id_lists):
    """Feature labels of the map the reference's tree makes of maps with the given label lists, in its order: a join keeps
    End's features in their order and appends Cur's unmatched ones in theirs (Imp.cpp:2601-2660); an unpaired trailing map
    is carried."""
    level = [np.asarray(a, np.int64) for a in id_lists]
    while len(level) > 1:
        nxt = []
        for i in range(0, len(level) - 1, 2):
            E, Cu = level[i], level[i + 1]
            nxt.append(np.concatenate([E, Cu[~np.isin(Cu, E)]]))
        if len(level) % 2:
            nxt.append(level[-1])
        level = nxt
    return level[0]


def merge_slices(slices, order):
    """The whole map from its feature slices (map dicts; slices[0] supplies poses and U) with the features in `order` (labels)."""
    base = slices[0]
    m = int(base["m"])
    order = np.asarray(order, np.int64)
    n = len(order)
    srt = np.argsort(order, kind="stable")
    out = {k: base[k] for k in base}
    stno = np.concatenate([np.asarray(base["stno"])[:6 * m], np.repeat(order, 3)]).astype(np.int32)
    stVal = np.concatenate([np.asarray(base["stVal"])[:6 * m], np.zeros(3 * n)])
    V = np.zeros((n, 9))
    lens = np.zeros(n, np.int64)
    gpos_all = []
    for sl in slices:
        ids = np.asarray(sl["stno"])[6 * m::3].astype(np.int64)
        gp = srt[np.searchsorted(order[srt], ids)] if len(ids) else np.zeros(0, np.int64)
        assert np.array_equal(order[gp], ids), "a slice holds a feature the joint order does not know"
        gpos_all.append(gp)
        stVal[6 * m:].reshape(n, 3)[gp] = np.asarray(sl["stVal"])[6 * m:].reshape(-1, 3)
        V[gp] = np.asarray(sl["V"]).reshape(-1, 9)
        fe = np.asarray(sl["feature"])
        if len(fe):
            lens[gp] += np.bincount(fe, minlength=len(ids))
    assert sum(len(g) for g in gpos_all) == n, "the slices do not add up to the joint map"
    FB = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    nW = int(FB[-1])
    W = np.zeros((nW, 18)); photo = np.zeros(nW, np.int32); feature = np.zeros(nW, np.int32)
    for sl, gp in zip(slices, gpos_all):
        fe = np.asarray(sl["feature"])
        k = len(fe)
        if not k:
            continue
        fb = np.asarray(sl["FBlock"]).astype(np.int64)
        dest = FB[gp[fe]] + (np.arange(k) - fb[fe])
        W[dest] = np.asarray(sl["W"]).reshape(-1, 18)
        photo[dest] = np.asarray(sl["photo"])
        feature[dest] = gp[fe]
    out.update(n=n, stno=stno, stVal=stVal, V=V, W=W, photo=photo, feature=feature, nW=nW, FBlock=FB[:-1].astype(np.int32))
    return out


# ---------------------------------------------------------------------------------------------------------------
# HIP back end: nodes stay on the devices
# ---------------------------------------------------------------------------------------------------------------
class ShardedTree:
    """One join tree over all ranks of `group`.  Every rank uploads ITS block of local maps once (resident in HBM);
    run() joins the whole tree and leaves the final map on rank 0 (download()).  run() can be repeated: the block trees
    and the merge trees keep their allocations and their plans, the packed buffers are reused."""

    def __init__(self, ctx, maps_block, lo, n_total, mono, group=None, device=None, top="shard", comm_bytes=0, shard_single=False, solve="owned"):
        """top: "shard" -- the levels above the blocks run feature-sharded on ALL ranks (all-reduces per level); "merge" --
        pairwise merge rounds on half of the remaining ranks each (no collective in the data path).  comm_bytes: size of the
        device buffer the all-reduced arrays live in (0: 512 MiB; it must hold the largest camera system, 288 bytes per block).
        shard_single: take the feature-sharded path with ONE rank too (tests: the whole collective path on one GPU)."""
        assert top in ("shard", "merge") and solve in ("owned", "replicated")
        # solve: how the camera systems of the feature-sharded top levels are factored -- "owned": rank r factors the columns of
        # block r's poses, the inter-block separators are summed and factored by everybody (lsfm_tree_set_comm_blocks);
        # "replicated": every rank factors everything (round 3)
        self.solve = solve
        self.ctx, self.mono, self.group = ctx, bool(mono), group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.block_maps, self.bounds = shard_bounds(n_total, self.world)
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
        self.top = top if (self.world > 1 or shard_single) else "merge"
        if self.block_tree is not None:
            # (feature-sharded top: an odd-indexed root goes back to its first frame inside the top tree's first level, on all ranks)
            ctx.tree_set_final_reanchor(self.block_tree, first_reanchor(self.rank, self.world) if self.top == "merge" else False)
        self.comm_bytes = int(comm_bytes) or (512 << 20)
        self.comm_buf = None    # device memory the all-reduced arrays live in (feature-sharded top)
        self.comm_cb = None     # keeps the ctypes callback alive
        self.comm_error = None
        self.top_tree = None
        self.slice_out, self.slice_in, self.slice_sizes = {}, {}, None
        self.block_ids = None   # feature labels of this rank's sub-tree root, in its order (for the order of the final map)
        self.order = None
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
        for t in list(self.merge_trees.values()) + [t for t in (self.block_tree, self.top_tree) if t is not None]:
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

    # -- feature-sharded top levels -----------------------------------------------------------------------------------
    def _allreduce_callback(self):
        """The function the library calls for every sum that crosses the GPUs (include/lsfm.h lsfm_allreduce_fn): `count`
        8-byte elements at `offset` of self.comm_buf, in place.  nccl: torch's all_reduce under the library's own stream
        (ExternalStream) -- ordered on the device, the host does not wait.  gloo (tests: several ranks on one GPU): staged
        through the host."""
        from . import api
        ext = torch.cuda.ExternalStream(int(self.ctx.stream() or 0), device=self.device)
        f64, i64 = self.comm_buf.view(torch.float64), self.comm_buf.view(torch.int64)

        def cb(user, offset, count, dtype, stream):
            try:
                v = (i64 if dtype == api.LSFM_DTYPE_I64 else f64)[offset // 8: offset // 8 + count]
                if self.gpu_direct:
                    with torch.cuda.stream(ext):
                        dist.all_reduce(v, group=self.group)
                elif not dist.is_initialized():
                    pass
                else:
                    ext.synchronize()
                    h = v.cpu()
                    dist.all_reduce(h, group=self.group)
                    v.copy_(h)
                    torch.cuda.synchronize(self.device)
                return 0
            except Exception as e:  # an exception must not unwind through the C frames
                self.comm_error = e
                return 1
        return api.ALLREDUCE_FN(cb)

    def _exchange_slices(self, sizes):
        """Slice g of this rank's root goes to rank g; slice `rank` of every other block arrives.  sizes[src][dst] bytes."""
        G, me = self.world, self.rank
        reqs, staged = [], []
        for b in range(G):
            if b == me or not self.nonempty[b]:
                continue
            n = int(sizes[b][me])
            buf = self.slice_in.get(b)
            if buf is None or buf.numel() != n:
                buf = self.slice_in[b] = torch.empty(n, dtype=torch.uint8, device=self.device)
            if self.gpu_direct:
                reqs.append(dist.P2POp(dist.irecv, buf, b, self.group))
            else:
                host = torch.empty(n, dtype=torch.uint8).pin_memory()
                staged.append((buf, host))
                reqs.append(dist.P2POp(dist.irecv, host, b, self.group))
        if self.block_tree is not None:
            for g in range(G):
                if g != me:
                    reqs.append(dist.P2POp(dist.isend, self.slice_out[g] if self.gpu_direct else self.slice_out[g].cpu(), g, self.group))
        if reqs:
            for w in dist.batch_isend_irecv(reqs):
                w.wait()
        for buf, host in staged:
            buf.copy_(host, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()  # the library reads the buffers on its own stream next

    @staticmethod
    def _add_stats(a, b):
        """Stats of two tree runs of one step as one record: times and counts add up, the residual is the larger one."""
        if a is None or b is None:
            return dict(b or a or {})
        out = {}
        for k in set(a) | set(b):
            x, y = a.get(k, 0), b.get(k, 0)
            out[k] = max(x, y) if k in ("max_rel_residual", "spmv_nnzb_upper_last", "spmv_rows_last", "upload_ms", "s_digest", "factor_digest") else x + y
        return out

    def _run_shard(self):
        import time
        ctx, G, me = self.ctx, self.world, self.rank
        stats, worst = None, 0
        t0 = time.perf_counter()
        if self.block_tree is not None:
            stats, rc = ctx.tree_run(self.block_tree)
            worst = max(worst, rc)
            if self.block_ids is None:  # structure: fetched once
                m, n, stno, _ = ctx.tree_download_state(self.block_tree)
                self.block_ids = stno[6 * m::3].copy()
            mine = ctx.tree_export_slice_sizes(self.block_tree, G)
            for g in range(G):
                buf = self.slice_out.get(g)
                if buf is None or buf.numel() != mine[g]:
                    buf = self.slice_out[g] = torch.empty(mine[g], dtype=torch.uint8, device=self.device)
                ctx.tree_export_slice_dev(self.block_tree, G, g, buf.data_ptr(), mine[g])  # synchronises the library's stream
        else:
            mine = [0] * G
        t1 = time.perf_counter()
        if self.slice_sizes is None:  # structure: exchanged once
            t = torch.tensor(mine, dtype=torch.int64, device=self.device if self.gpu_direct else "cpu")
            allt = [torch.zeros_like(t) for _ in range(G)]
            if dist.is_initialized():
                dist.all_gather(allt, t, group=self.group)
            else:
                allt = [t]
            self.slice_sizes = [[int(v) for v in a.tolist()] for a in allt]
        elif [int(v) for v in mine] != self.slice_sizes[me]:
            raise RuntimeError("the slices of this rank's sub-tree root changed their sizes between two runs of a resident tree")
        self._exchange_slices(self.slice_sizes)
        parts = [(self.slice_out[me] if b == me else self.slice_in[b]) for b in range(G) if self.nonempty[b]]
        ptrs = [b.data_ptr() for b in parts]
        if self.top_tree is None:
            self.top_tree = ctx.tree_upload_dev(ptrs, self.mono)
            ctx.tree_set_plans(self.top_tree, self.plans)
            # (not zeroed: the library zeroes every region it carves out, on its own stream -- a fill on torch's stream would race it)
            self.comm_buf = torch.empty(self.comm_bytes, dtype=torch.uint8, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()
            self.comm_cb = self._allreduce_callback()
            ctx.tree_set_comm(self.top_tree, me, G, self.comm_cb, self.comm_buf.data_ptr(), self.comm_bytes)
            ctx.tree_set_comm_blocks(self.top_tree, self.block_maps if self.solve == "owned" else 0)
        else:
            ctx.tree_reload_dev(self.top_tree, ptrs)
        self.comm_error = None
        t2 = time.perf_counter()
        try:
            top_stats, rc = ctx.tree_run(self.top_tree)
        except Exception:
            if self.comm_error is not None:
                raise self.comm_error
            raise
        t3 = time.perf_counter()
        worst = max(worst, rc)
        self.result = self.top_tree
        stats = self._add_stats(stats, top_stats)
        # host wall clock of the step's phases on this rank: own block (+ cutting its root into slices), all-to-all of the slices
        # (+ unpacking them), the feature-sharded top levels
        stats["phase_block_ms"], stats["phase_exchange_ms"], stats["phase_top_ms"] = 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)
        return stats, worst

    def _download_shard(self, full):
        """Rank 0: the final map put together from the ranks' feature slices, in the order the single tree leaves it in
        (full=False: state vector only -- stno, stVal, m, n).  Collective: every rank calls it."""
        ctx, G = self.ctx, self.world
        if self.order is None:
            ids = [None] * G
            if dist.is_initialized():
                dist.all_gather_object(ids, self.block_ids, group=self.group)
            else:
                ids = [self.block_ids]
            self.order = joint_feature_order([a for a, ne in zip(ids, self.nonempty) if ne])
        if full:
            piece = ctx.tree_download(self.top_tree)
        else:
            m, n, stno, stVal = ctx.tree_download_state(self.top_tree)
            piece = dict(m=m, n=n, stno=stno, stVal=stVal, V=np.zeros((n, 9)), W=np.zeros((0, 18)), photo=np.zeros(0, np.int32),
                         feature=np.zeros(0, np.int32), FBlock=np.zeros(n, np.int32))
        pieces = [None] * G if self.rank == 0 else None
        if dist.is_initialized():
            dist.gather_object(piece, pieces, dst=0, group=self.group)
        else:
            pieces = [piece]
        if self.rank != 0:
            return None
        return merge_slices(pieces, self.order)

    # -- one whole tree ---------------------------------------------------------------------------------------------
    def run(self):
        """Returns (stats of the last tree run on this rank or None, worst return code on this rank)."""
        if self.top == "shard":
            return self._run_shard()
        try:
            return self._run_merge()
        finally:
            # whatever ended the run: no receive posted by it may be left behind (the next run posts its own, and untagged
            # messages would then meet the wrong buffers)
            for slot in list(self.pending):
                _, _, w0, w1 = self.pending.pop(slot)
                for w in (w0, w1):
                    try:
                        # (bounded: a peer that failed as well never sends, and an endless wait here would hide the exception
                        # that ended the run)
                        w.wait(timeout=datetime.timedelta(seconds=30))
                    except Exception:
                        pass

    def _run_merge(self):
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

    def download(self, full=True):
        """The final map (rank 0, after run()).  With the feature-sharded top every rank holds a slice of the features: the call
        is then collective (every rank makes it, rank 0 gets the map, the others None); full=False gathers the state vector only."""
        if self.top == "shard":
            return self._download_shard(full)
        if self.result is None:
            raise RuntimeError("no result on this rank (rank 0 holds it after run())")
        return self.ctx.tree_download(self.result)

    def close(self):
        for t in list(self.merge_trees.values()) + [t for t in (self.block_tree, self.top_tree) if t is not None]:
            self.ctx.tree_free(t)
        self.merge_trees, self.block_tree, self.top_tree, self.result = {}, None, None, None
        self.comm_cb = None
