#!/usr/bin/env python3
"""Local-map ingestion: the product's reader (one piece per file, hand tokeniser, host threads)
against the fscanf loop of the reference (oracle port of Imp.cpp:3044-3132), same files, bit-identical arrays.
usage: python tools/reader_bench.py [maps=256] [new_per_frame=130] [vis=5]"""
import ctypes as C
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from linearsfm_amd import api, synth  # noqa: E402
from oracle import pyoracle as po  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    npf = int(sys.argv[2]) if len(sys.argv) > 2 else 130
    vis = int(sys.argv[3]) if len(sys.argv) > 3 else 5
    po.build()
    d = tempfile.mkdtemp()
    synth.write_set(d, synth.make_stereo_set(N, new_per_frame=npf, vis=vis, seed=0))
    mb = sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / 1e6
    L = api.lib()
    res = dict(maps=N, text_MB=mb, host_cores=os.cpu_count())
    t = time.time()
    for k in range(N):
        po.read_map(os.path.join(d, f"localmap_{k + 1}.txt"), False)
    res["fscanf_port_s"] = time.time() - t
    for th in (1, 8, 32):
        arr = (api.LsfmMap * N)()
        bad = C.c_int()
        t = time.time()
        assert L.lsfm_read_localmaps(d.encode(), 1, N, 0, th, arr, C.byref(bad)) == 0
        res[f"lsfm_read_localmaps_{th}_threads_s"] = time.time() - t
        for k in range(N):
            L.lsfm_map_release(C.byref(arr[k]))
    print(json.dumps(res))


if __name__ == "__main__":
    main()
