"""Where a round (256 W blocks) of k_tr_entries spends its clocks, phase by phase, over one NC3500-like tree (lane 0 of every
work-group; the timers themselves cost a few per cent).  Needs  make -C linearsfm_amd/csrc clean; make -C linearsfm_amd/csrc K9_TIMING=1"""
import ctypes as C, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from linearsfm_amd import api, synth
_, maps = synth.make_config('nc3500', seed=0)
ctx = api.Context(0); L = api.lib()
t = ctx.tree_upload([dict(m if isinstance(m, dict) else m.__dict__) for m in maps], False)
ctx.tree_run(t); ctx.tree_run(t)
out = (C.c_ulonglong * 16)()
L.lsfm_debug_tr(out, 1); ctx.tree_run(t); L.lsfm_debug_tr(out, 0)
v = np.array(list(out), dtype=np.float64)
names = ['tile set-up', 'round set-up', 'own block (index, W, D^T W D, store)', 'prefetch issue', '-', 'C_k, W^T C_k, C_f', 'barrier waits', 'feature sums', 'pose rows (LDS atomics)', 'flush']
tot = v[:10].sum()
print('tiles', v[12], 'rounds', v[11], 'clocks per round', tot / max(v[11], 1))
for n, x in zip(names, v[:10]):
    print(f'  {n:40s} {100*x/tot:5.1f} %  {x/max(v[11],1):9.0f} per round')
