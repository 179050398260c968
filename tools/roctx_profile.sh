#!/bin/bash
# roctx ranges of a tree run next to its kernels: LSFM_ROCTX=1 + rocprofv3 --marker-trace (no counters in this pass)
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT" || exit 1
export LSFM_ROCTX=1
rocprofv3 --marker-trace --kernel-trace --stats -d gpurun_out/roctx -o run -- python3 bench.py --steps 2 --warmup 1 > gpurun_out/roctx_bench.log 2> gpurun_out/roctx_bench.err
ls -R gpurun_out/roctx | head -30
