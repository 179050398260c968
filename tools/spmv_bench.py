#!/usr/bin/env python3
"""K10a stand-alone: the 6x6-block symmetric SpMV of the CG (k_spmv) on Schur-like matrices of growing size, to show the
kernel's streaming rate once the matrix no longer fits the caches (at the NC3500-like size S is 15 MB and the SpMV
runs from L2 / Infinity Cache).  Pattern: pose chain with `band` neighbours + `hubs` dense hub rows (like S at the top
of a join tree).  Bytes = nnzb*(288+4) + 4(m+1) + 96 m on the upper-block storage.
usage: python tools/spmv_bench.py [m ...]"""
import json
import sys

import numpy as np

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from linearsfm_amd import api, synth  # noqa: E402


def matrix(m, band=12, hubs=12, seed=0):
    return synth.schur_like_matrix(m, band, hubs, seed)


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [3499, 16384, 65536, 262144]
    ctx = api.Context(0)
    out = []
    for m in sizes:
        rowptr, colidx, val = matrix(m)
        x = np.random.default_rng(1).normal(size=6 * m)
        _, ms, by = ctx.spmv_bench(rowptr, colidx, val, x, reps=20)
        out.append(dict(m=m, nnzb_upper=int(len(colidx)), matrix_MB=len(colidx) * 288 / 1e6, avg_launch_ms=ms,
                        algorithmic_GBps=by / (ms * 1e-3) / 1e9, frac_of_8TBps=by / (ms * 1e-3) / 8e12))
        print(json.dumps(out[-1]))
    ctx.close()


if __name__ == "__main__":
    main()
