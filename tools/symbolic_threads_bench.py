"""Host-side symbolic analysis of a level with 16 independent systems of 1 024 poses each: LSFM_SYM_THREADS=1|8 python tools/symbolic_threads_bench.py
(same result whatever the thread count: the digest printed last)."""
import sys, ctypes as C, hashlib
sys.path.insert(0, '.')
import numpy as np
from linearsfm_amd import api, synth
nsys, msys = 16, 1024
rp1, ci1, _ = synth.schur_like_matrix(msys, band=30, hubs=10, seed=3)
rp = [0]; ci = []
for g in range(nsys):
    ci.append(ci1 + g*msys); rp += list(rp1[1:] + rp[-1])
rp = np.array(rp, np.int32); ci = np.concatenate(ci).astype(np.int32)
m = nsys*msys
origin = np.arange(m, dtype=np.int32)
L=api.lib(); P=C.POINTER
perm=np.zeros(m,np.int32); colptr=np.zeros(m+1,np.int32); cap=30_000_000
rowidx=np.zeros(cap,np.int32); info=np.zeros(8,np.int32); ms=C.c_double()
rc=L.lsfm_symbolic_analyse(m, rp.ctypes.data_as(P(C.c_int)), ci.ctypes.data_as(P(C.c_int)), origin.ctypes.data_as(P(C.c_int)), 3, perm.ctypes.data_as(P(C.c_int)), colptr.ctypes.data_as(P(C.c_int)), rowidx.ctypes.data_as(P(C.c_int)), cap, info.ctypes.data_as(P(C.c_int)), C.byref(ms))
h=hashlib.sha1(perm.tobytes()+colptr.tobytes()+rowidx[:colptr[m]].tobytes()).hexdigest()[:16]
print(rc, len(ci), info[:6], round(ms.value,2), h)
