"""The library's own fill and small-copy kernels (lsfm_prims.hip: k_fill_words / k_fill_bytes / k_copy_words, CopyBatch reading the
pinned ring) stand where the runtime's hipMemsetAsync / hipMemcpyAsync stood: every accumulator of the path is cleared and every index
table arrives through them.  lsfm_selftest_prims drives them with random offsets, lengths and bytes -- unaligned heads and tails
included -- and compares with the host; here through the C ABI, with the runtime's paths (LSFM_RUNTIME_FILL / _COPY) as a cross-check in
a process of their own."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_own_fill_and_copy_kernels_match_the_host(ctx, seed):
    ctx.selftest_prims(96, seed)


def test_runtime_fill_and_copy_paths_still_work():
    code = ("import sys; sys.path.insert(0, '.')\n"
            "from linearsfm_amd import api\n"
            "c = api.Context(0); c.selftest_prims(48, 7); print('ok')\n")
    env = dict(os.environ, LSFM_RUNTIME_FILL="1", LSFM_RUNTIME_COPY="1")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
