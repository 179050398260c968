"""Sub-tree lanes on ONE GPU: the join tree cut at a power-of-two boundary into K blocks (linearsfm_amd.distributed.shard_bounds --
node k of level L covers maps [k 2^L, (k + 1) 2^L), so the blocks are sub-trees of the one tree), every block on its own
context (own streams, own arenas) driven by its own host thread, the roots handed over in device memory and merged pairwise
(merge_schedule).  What two independent sub-trees in flight at once make of a chip whose kernels are latency-bound.

    python tools/lanes_probe.py [config] [K ...]           e.g.  python tools/lanes_probe.py nc3500 1 2 4

Prints per K: ms per tree (analysing runs / repeat runs), and the largest difference of the final state from K = 1."""
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from linearsfm_amd import api, synth  # noqa: E402
from linearsfm_amd.distributed import first_reanchor, merge_schedule, shard_bounds  # noqa: E402


class Lanes:
    def __init__(self, config, K):
        typ, N = synth.CONFIGS[config][0], synth.CONFIGS[config][1]
        self.mono, self.K = typ == "Monocular", K
        _, self.bounds = shard_bounds(N, K)
        self.nonempty = [hi > lo for lo, hi in self.bounds]
        self.ctx, self.tree = [], []
        for k, (lo, hi) in enumerate(self.bounds):
            ctx = api.Context(0)
            ctx.set_pcg(1e-12, 0)
            tree = None
            if hi > lo:
                _, block = synth.make_config(config, N, seed=0, only=(lo, hi))
                dicts = []
                for j, m in enumerate(block):
                    d = dict(m if isinstance(m, dict) else m.__dict__)
                    d.setdefault("pose_origin", np.full(int(d["m"]), lo + j, np.int32))
                    dicts.append(d)
                tree = ctx.tree_upload(dicts, self.mono)
                ctx.tree_set_final_reanchor(tree, first_reanchor(k, K))
            self.ctx.append(ctx)
            self.tree.append(tree)
        self.merge = [dict() for _ in range(K)]
        self.bufs = [dict() for _ in range(K)]
        self.mail = {}            # (dst, round) -> tensor
        self.cv = threading.Condition()
        self.plans = False
        self.result = None
        self.start = threading.Barrier(K + 1)
        self.done = threading.Barrier(K + 1)
        self.stop = False
        self.err = []
        self.threads = [threading.Thread(target=self._loop, args=(k,), daemon=True) for k in range(K)]
        for t in self.threads:
            t.start()

    def _export(self, k, tree, slot):
        ctx = self.ctx[k]
        n = ctx.tree_export_size(tree)
        buf = self.bufs[k].get(slot)
        if buf is None or buf.numel() != n:
            buf = self.bufs[k][slot] = torch.empty(n, dtype=torch.uint8, device="cuda:0")
        ctx.tree_export_dev(tree, buf.data_ptr(), n)
        return buf

    def _run_lane(self, k):
        ctx, cur = self.ctx[k], None
        if self.tree[k] is not None:
            ctx.tree_set_plans(self.tree[k], self.plans)
            ctx.tree_run(self.tree[k])
            cur = self.tree[k]
        for act in merge_schedule(k, self.K, self.nonempty):
            if act[0] == "send":
                _, peer, j = act
                if cur is not None:
                    buf = self._export(k, cur, ("send", j))
                    with self.cv:
                        self.mail[(peer, j)] = buf
                        self.cv.notify_all()
                return
            _, peer, j, reanchor = act
            parts = []
            if cur is not None:
                parts.append(self._export(k, cur, ("own", j)))
            if peer is not None:
                with self.cv:
                    while (k, j) not in self.mail:
                        self.cv.wait()
                    parts.append(self.mail.pop((k, j)))
            if not parts:
                cur = None
                continue
            ptrs = [b.data_ptr() for b in parts]
            mt = self.merge[k].get(j)
            if mt is None:
                mt = self.merge[k][j] = ctx.tree_upload_dev(ptrs, self.mono)
            else:
                ctx.tree_reload_dev(mt, ptrs)
            ctx.tree_set_plans(mt, self.plans)
            ctx.tree_set_final_reanchor(mt, reanchor)
            ctx.tree_run(mt)
            cur = mt
        if k == 0:
            self.result = cur

    def _loop(self, k):
        while True:
            self.start.wait()
            if self.stop:
                return
            try:
                self._run_lane(k)
            except Exception as e:  # noqa: BLE001
                self.err.append((k, repr(e)))
            self.done.wait()

    def run(self):
        self.start.wait()
        self.done.wait()
        if self.err:
            raise RuntimeError(self.err)

    def timed(self, steps, plans):
        self.plans = plans
        self.run()
        self.run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            self.run()
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / steps

    def state(self):
        return self.ctx[0].tree_download_state(self.result)

    def close(self):
        self.stop = True
        self.start.wait()


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else "nc3500"
    Ks = [int(v) for v in sys.argv[2:]] or [1, 2, 4]
    ref = None
    for K in Ks:
        L = Lanes(config, K)
        a = L.timed(10, False)
        st = L.state()
        r = L.timed(10, True)
        vals = np.asarray(st[-1] if isinstance(st, (tuple, list)) else st["stVal"], dtype=np.float64)
        if ref is None:
            ref = vals
        diff = float(np.max(np.abs(vals - ref) / np.maximum(1.0, np.abs(ref)))) if vals.shape == ref.shape else float("nan")
        print(f"{config}: {K} lane(s): analysing {a:.2f} ms per tree, repeat {r:.2f} ms; state vs the first line {diff:.2e}", flush=True)
        L.close()


if __name__ == "__main__":
    main()
