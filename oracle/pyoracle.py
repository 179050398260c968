"""ORACLE -- TEST INFRASTRUCTURE ONLY: ctypes binding of oracle/liblsfm_oracle.so (see lsfm_oracle.h).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


class OrcMap(C.Structure):
    _fields_ = [("r", C.c_int), ("Ref", C.c_int), ("FRef", C.c_int),
                ("m", C.c_int), ("n", C.c_int), ("nU", C.c_int), ("nW", C.c_int),
                ("ScaP", C.c_int), ("Fix", C.c_int), ("Sign", C.c_int), ("FScaP", C.c_int), ("FFix", C.c_int),
                ("stno", C.POINTER(C.c_int)), ("stVal", C.POINTER(C.c_double)),
                ("U", C.POINTER(C.c_double)), ("Ui", C.POINTER(C.c_int)), ("Uj", C.POINTER(C.c_int)),
                ("W", C.POINTER(C.c_double)), ("photo", C.POINTER(C.c_int)), ("feature", C.POINTER(C.c_int)),
                ("V", C.POINTER(C.c_double)), ("FBlock", C.POINTER(C.c_int))]


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE, "liblsfm_oracle.so", "lsfm_oracle"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "liblsfm_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        P = C.POINTER
        dp, ip = P(C.c_double), P(C.c_int)
        L.orc_read_map.argtypes = [C.c_char_p, C.c_int, P(OrcMap)]
        L.orc_write_map.argtypes = [C.c_char_p, C.c_int, P(OrcMap)]
        L.orc_map_free.argtypes = [P(OrcMap)]
        L.orc_map_copy.argtypes = [P(OrcMap), P(OrcMap)]
        L.orc_transform_stereo.argtypes = [P(OrcMap), C.c_int, P(OrcMap)]
        L.orc_transform_mono.argtypes = [P(OrcMap), C.c_int, C.c_int, C.c_int, P(OrcMap)]
        L.orc_join_assemble_stereo.argtypes = [P(OrcMap), P(OrcMap), P(OrcMap), P(dp), P(dp)]
        L.orc_join_assemble_mono.argtypes = [P(OrcMap), P(OrcMap), P(OrcMap), P(dp), P(dp), ip]
        L.orc_solve_stereo.argtypes = [dp, dp, dp, dp, dp, dp, ip, ip, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                       P(C.c_long)]
        L.orc_solve_mono.argtypes = [dp, dp, dp, dp, dp, dp, ip, ip, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, P(C.c_long)]
        L.orc_schur.argtypes = [dp, dp, dp, dp, dp, ip, ip, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                P(ip), P(ip), P(dp), P(dp), P(dp)]
        L.orc_join_stereo.argtypes = [P(OrcMap), P(OrcMap), P(OrcMap)]
        L.orc_join_mono.argtypes = [P(OrcMap), P(OrcMap), P(OrcMap)]
        L.orc_divide_conquer.argtypes = [P(OrcMap), C.c_int, C.c_int, P(OrcMap), C.c_int, dp]
        L.orc_divide_conquer_omp.argtypes = [P(OrcMap), C.c_int, C.c_int, P(OrcMap), C.c_int, dp]
        L.orc_set_match_hash.argtypes = [C.c_int]
        L.orc_set_final_reanchor.argtypes = [C.c_int]
        L.orc_set_extended.argtypes = [C.c_int]
        L.orc_set_comm.argtypes = [C.c_int, C.c_int, REDUCE_FN]
        L.orc_schur_csc.argtypes = [dp, ip, ip, C.c_int, C.c_int, C.c_int, P(ip), P(ip), P(dp)]
        L.orc_schur_csc.restype = C.c_int
        L.orc_solve_features.argtypes = [dp, dp, dp, dp, dp, C.c_int, ip, ip, C.c_int]
        L.orc_gn_polish.argtypes = [P(OrcMap), C.c_int, C.c_int, P(OrcMap), C.c_int, dp, dp, ip]
        L.orc_gn_objective.argtypes = [P(OrcMap), C.c_int, C.c_int, P(OrcMap), dp, dp]
        L.orc_gn_hessian_times.argtypes = [P(OrcMap), C.c_int, C.c_int, P(OrcMap), dp, dp]
        L.free = C.CDLL(None).free
        L.free.argtypes = [C.c_void_p]
        _LIB = L
    return _LIB


_KEEP = {}

# orc_reduce_fn (lsfm_oracle.h): sums `count` 8-byte elements at buf over the processes, in place (dtype 0: double, 1: int64)
REDUCE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_long, C.c_int)


def torch_reduce_fn(group=None):
    """An orc_reduce_fn over torch.distributed (gloo on CPU): the feature-sharded evaluation of a tree by several processes."""
    import torch
    import torch.distributed as dist

    def fn(buf, count, dtype):
        if count <= 0:
            return
        a = np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_int64 if dtype == 1 else C.c_double)), shape=(count,))
        t = torch.from_numpy(a)
        dist.all_reduce(t, group=group)  # in place: t shares the C buffer
    return REDUCE_FN(fn)


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def map_to_dict(g: OrcMap):
    r = 6 * g.m + 3 * g.n
    return dict(Ref=g.Ref, FRef=g.FRef, m=g.m, n=g.n, nU=g.nU, nW=g.nW, ScaP=g.ScaP, Fix=g.Fix, Sign=g.Sign,
                FScaP=g.FScaP, FFix=g.FFix,
                stno=_arr(g.stno, r, np.int32), stVal=_arr(g.stVal, r, np.float64),
                U=_arr(g.U, 36 * g.nU, np.float64).reshape(-1, 36), Ui=_arr(g.Ui, g.nU, np.int32),
                Uj=_arr(g.Uj, g.nU, np.int32),
                W=_arr(g.W, 18 * g.nW, np.float64).reshape(-1, 18), photo=_arr(g.photo, g.nW, np.int32),
                feature=_arr(g.feature, g.nW, np.int32),
                V=_arr(g.V, 9 * g.n, np.float64).reshape(-1, 9), FBlock=_arr(g.FBlock, g.n, np.int32))


def dict_to_map(d) -> OrcMap:
    """Builds an OrcMap whose arrays are malloc'ed by libc (the oracle frees consumed maps)."""
    L = lib()
    g = OrcMap()
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]

    def put(a, ctype, dtype):
        a = np.ascontiguousarray(a, dtype=dtype).reshape(-1)
        p = libc.malloc(max(a.nbytes, 8))
        C.memmove(p, a.ctypes.data, a.nbytes)
        return C.cast(p, C.POINTER(ctype))
    g.m, g.n = int(d["m"]), int(d["n"])
    g.nU, g.nW = int(len(d["Ui"])), int(len(d["photo"]))
    g.r = 6 * g.m + 3 * g.n
    g.Ref, g.FRef = int(d["Ref"]), int(d.get("FRef", d["Ref"]))
    g.ScaP, g.Fix, g.Sign = int(d.get("ScaP", 0)), int(d.get("Fix", 0)), int(d.get("Sign", 1))
    g.FScaP, g.FFix = int(d.get("FScaP", g.ScaP)), int(d.get("FFix", g.Fix))
    g.stno = put(d["stno"], C.c_int, np.int32)
    g.stVal = put(d["stVal"], C.c_double, np.float64)
    g.U = put(d["U"], C.c_double, np.float64)
    g.Ui = put(d["Ui"], C.c_int, np.int32)
    g.Uj = put(d["Uj"], C.c_int, np.int32)
    g.W = put(d["W"], C.c_double, np.float64)
    g.photo = put(d["photo"], C.c_int, np.int32)
    g.feature = put(d["feature"], C.c_int, np.int32)
    g.V = put(d["V"], C.c_double, np.float64)
    g.FBlock = put(d["FBlock"], C.c_int, np.int32)
    return g


def localmap_to_dict(lm):
    return dict(Ref=lm.Ref, FRef=lm.FRef, m=lm.m, n=lm.n, ScaP=lm.ScaP, Fix=lm.Fix, Sign=lm.Sign, FScaP=lm.FScaP,
                FFix=lm.FFix, stno=lm.stno, stVal=lm.stVal, U=lm.U, Ui=lm.Ui, Uj=lm.Uj, W=lm.W, photo=lm.photo,
                feature=lm.feature, V=lm.V, FBlock=lm.FBlock)


def read_map(path, mono):
    g = OrcMap()
    rc = lib().orc_read_map(path.encode(), int(mono), C.byref(g))
    if rc:
        raise IOError(f"orc_read_map({path}) -> {rc}")
    return g


def transform(d, mono, Ref, ScaP=0, Fix=0):
    L = lib()
    gi = dict_to_map(d)
    go = OrcMap()
    if mono:
        L.orc_transform_mono(C.byref(gi), Ref, ScaP, Fix, C.byref(go))
    else:
        L.orc_transform_stereo(C.byref(gi), Ref, C.byref(go))
    out = map_to_dict(go)
    L.orc_map_free(C.byref(gi))
    L.orc_map_free(C.byref(go))
    return out


def join_assemble(dEnd, dCur, mono):
    """Returns (joint dict [stVal zero], eP, eF, solve_args or None, End dict after in-place angle wrap, Cur dict)."""
    L = lib()
    ge, gc, gj = dict_to_map(dEnd), dict_to_map(dCur), OrcMap()
    eP, eF = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    sa = None
    if mono:
        sa_c = (C.c_int * 5)()
        L.orc_join_assemble_mono(C.byref(ge), C.byref(gc), C.byref(gj), C.byref(eP), C.byref(eF), sa_c)
        sa = list(sa_c)
    else:
        L.orc_join_assemble_stereo(C.byref(ge), C.byref(gc), C.byref(gj), C.byref(eP), C.byref(eF))
    j = map_to_dict(gj)
    ePa, eFa = _arr(eP, 6 * gj.m, np.float64), _arr(eF, 3 * gj.n, np.float64)
    L.free(eP)
    L.free(eF)
    de, dc = map_to_dict(ge), map_to_dict(gc)
    for g in (ge, gc, gj):
        L.orc_map_free(C.byref(g))
    return j, ePa, eFa, sa, de, dc


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t))


def solve(j, eP, eF, mono, sa=None, extended=False):
    """Schur + direct solve + back-substitution on assembled joint arrays; returns (stVal, rc, stats).
    extended: the same statements in long double (orc_set_extended), result rounded to fp64."""
    L = lib()
    L.orc_set_extended(int(extended))
    m, n = int(j["m"]), int(j["n"])
    st = np.zeros(6 * m + 3 * n)
    U = np.ascontiguousarray(j["U"], np.float64); W = np.ascontiguousarray(j["W"], np.float64)
    V = np.ascontiguousarray(j["V"], np.float64)
    Ui = np.ascontiguousarray(j["Ui"], np.int32); Uj = np.ascontiguousarray(j["Uj"], np.int32)
    ph = np.ascontiguousarray(j["photo"], np.int32); fe = np.ascontiguousarray(j["feature"], np.int32)
    eP = np.ascontiguousarray(eP, np.float64); eF = np.ascontiguousarray(eF, np.float64)
    stats = (C.c_long * 2)()
    d, i = C.c_double, C.c_int
    if mono:
        rc = L.orc_solve_mono(_p(st, d), _p(eF, d), _p(eP, d), _p(U, d), _p(W, d), _p(V, d), _p(Ui, i), _p(Uj, i),
                              _p(ph, i), _p(fe, i), m, n, len(Ui), len(ph), sa[0], sa[1], sa[2], sa[3], sa[4], stats)
    else:
        rc = L.orc_solve_stereo(_p(st, d), _p(eF, d), _p(eP, d), _p(U, d), _p(W, d), _p(V, d), _p(Ui, i), _p(Uj, i),
                                _p(ph, i), _p(fe, i), m, n, len(Ui), len(ph), stats)
    return st, rc, (stats[0], stats[1])


def schur(j, eP, eF, accumulate_u):
    """S (block CRS upper; diagonal blocks upper-triangle only), E, Vinv."""
    L = lib()
    m, n = int(j["m"]), int(j["n"])
    U = np.ascontiguousarray(j["U"], np.float64); W = np.ascontiguousarray(j["W"], np.float64)
    V = np.ascontiguousarray(j["V"], np.float64)
    Ui = np.ascontiguousarray(j["Ui"], np.int32); Uj = np.ascontiguousarray(j["Uj"], np.int32)
    ph = np.ascontiguousarray(j["photo"], np.int32); fe = np.ascontiguousarray(j["feature"], np.int32)
    eP = np.ascontiguousarray(eP, np.float64); eF = np.ascontiguousarray(eF, np.float64)
    d, i = C.c_double, C.c_int
    rp, ci = C.POINTER(i)(), C.POINTER(i)()
    S, E, IV = C.POINTER(d)(), C.POINTER(d)(), C.POINTER(d)()
    L.orc_schur(_p(eF, d), _p(eP, d), _p(U, d), _p(W, d), _p(V, d), _p(Ui, i), _p(Uj, i), _p(ph, i), _p(fe, i),
                m, n, len(Ui), len(ph), int(accumulate_u), C.byref(rp), C.byref(ci), C.byref(S), C.byref(E),
                C.byref(IV))
    rowptr = _arr(rp, m + 1, np.int32)
    nuis = int(rowptr[m])
    out = (rowptr, _arr(ci, nuis, np.int32), _arr(S, 36 * nuis, np.float64).reshape(-1, 6, 6),
           _arr(E, 6 * m, np.float64), _arr(IV, 9 * n, np.float64).reshape(-1, 3, 3))
    for p in (rp, ci, S, E, IV):
        L.free(p)
    return out


def solve_features(j, IV, eF, dpa):
    """pba_solveFeatures alone: feature values from given pose values (Imp.cpp:2980-3020)."""
    L = lib()
    n = int(j["n"])
    W = np.ascontiguousarray(j["W"], np.float64); IV = np.ascontiguousarray(IV, np.float64)
    eF = np.ascontiguousarray(eF, np.float64); dpa = np.ascontiguousarray(dpa, np.float64)
    ph = np.ascontiguousarray(j["photo"], np.int32); fe = np.ascontiguousarray(j["feature"], np.int32)
    dpb = np.zeros(3 * n)
    d, i = C.c_double, C.c_int
    L.orc_solve_features(_p(W, d), _p(IV, d), _p(eF, d), _p(dpa, d), _p(dpb, d), n, _p(ph, i), _p(fe, i), len(ph))
    return dpb


def schur_csc(S, rowptr, colidx, m, skipblk=-1, skipfix=-1):
    """Scalar CSC (upper) of a block-CRS S as handed to CHOLMOD (pba_constructCSSLM / GN): (Sp, Si, Sx)."""
    L = lib()
    S = np.ascontiguousarray(S, np.float64); rowptr = np.ascontiguousarray(rowptr, np.int32)
    colidx = np.ascontiguousarray(colidx, np.int32)
    d, i = C.c_double, C.c_int
    Sp, Si, Sx = C.POINTER(i)(), C.POINTER(i)(), C.POINTER(d)()
    ns = L.orc_schur_csc(_p(S, d), _p(rowptr, i), _p(colidx, i), int(m), int(skipblk), int(skipfix), C.byref(Sp), C.byref(Si),
                         C.byref(Sx))
    sp = _arr(Sp, ns + 1, np.int32)
    out = (sp, _arr(Si, int(sp[ns]), np.int32), _arr(Sx, int(sp[ns]), np.float64))
    for p in (Sp, Si, Sx):
        L.free(p)
    return out


def divide_conquer(dicts, mono, verbose=False, match_hash=True, final_reanchor=True, threads=0, extended=False, comm=None):
    """Full hierarchical join of a list of map dicts; returns (final map dict, timing[4], rc).
    threads > 0: the independent joins of a level on that many host threads (same result, timing[0] only).
    extended: every solve of the tree in long double (transform and assembly stay fp64: they are pinned to the reference).
    comm = (rank, world, REDUCE_FN): feature-sharded evaluation -- `dicts` are this process's slices of the maps (orc_set_comm)."""
    L = lib()
    if comm is not None:
        assert threads == 0 and not extended
        L.orc_set_comm(int(comm[0]), int(comm[1]), comm[2])
    else:
        L.orc_set_comm(0, 1, C.cast(None, REDUCE_FN))
    L.orc_set_extended(int(extended))
    L.orc_set_match_hash(int(match_hash))
    L.orc_set_final_reanchor(int(final_reanchor))
    N = len(dicts)
    arr = (OrcMap * N)()
    for k, d in enumerate(dicts):
        arr[k] = dict_to_map(d)
    out = OrcMap()
    timing = (C.c_double * 4)()
    if threads > 0:
        rc = L.orc_divide_conquer_omp(arr, N, int(mono), C.byref(out), int(threads), timing)
    else:
        rc = L.orc_divide_conquer(arr, N, int(mono), C.byref(out), int(verbose), timing)
    if comm is not None:
        L.orc_set_comm(0, 1, C.cast(None, REDUCE_FN))
    res = map_to_dict(out)
    L.orc_map_free(C.byref(out))
    return res, list(timing), rc


def _gn_maps(dicts, G):
    N = len(dicts)
    arr = (OrcMap * N)()
    for k, d in enumerate(dicts):
        arr[k] = dict_to_map(d)
    g = dict_to_map(G)
    return arr, g


def _gn_free(arr, g):
    L = lib()
    for k in range(len(arr)):
        L.orc_map_free(C.byref(arr[k]))
    L.orc_map_free(C.byref(g))


def gn_objective(dicts, mono, G, want_grad=True):
    """F(x) = sum_k ||x^_k - f_k(x)||^2_{I_k} and b = sum_k J_k^T I_k r_k (= -grad F / 2) at the global state G (lsfm_gn.inc)."""
    L = lib()
    arr, g = _gn_maps(dicts, G)
    F = C.c_double(0.0)
    grad = np.zeros(6 * g.m + 3 * g.n) if want_grad else None
    rc = L.orc_gn_objective(arr, len(dicts), int(mono), C.byref(g), C.byref(F), _p(grad, C.c_double) if want_grad else None)
    _gn_free(arr, g)
    if rc:
        raise ValueError(f"orc_gn_objective -> {rc}")
    return F.value, grad


def gn_hessian_times(dicts, mono, G, v):
    """H v for the step matrix H = sum_k J_k^T I_k J_k at the global state G."""
    L = lib()
    arr, g = _gn_maps(dicts, G)
    v = np.ascontiguousarray(v, np.float64)
    y = np.zeros_like(v)
    rc = L.orc_gn_hessian_times(arr, len(dicts), int(mono), C.byref(g), _p(v, C.c_double), _p(y, C.c_double))
    _gn_free(arr, g)
    if rc:
        raise ValueError(f"orc_gn_hessian_times -> {rc}")
    return y


def gn_polish(dicts, mono, G, iters, extended=False):
    """`iters` Gauss-Newton steps of the map-joining objective from the global state G (a map dict: stno, stVal, Ref, Mono: ScaP, Fix).
    Returns (stVal, obj[iters + 1], gnorm[iters + 1], halvings[iters], rc).  extended: the steps' solves in long double (the twin)."""
    L = lib()
    L.orc_set_extended(int(extended))
    L.orc_set_comm(0, 1, C.cast(None, REDUCE_FN))
    arr, g = _gn_maps(dicts, G)
    obj, gn, hv = np.zeros(iters + 1), np.zeros(iters + 1), np.zeros(max(iters, 1), np.int32)
    rc = L.orc_gn_polish(arr, len(dicts), int(mono), C.byref(g), int(iters), _p(obj, C.c_double), _p(gn, C.c_double), _p(hv, C.c_int))
    st = _arr(g.stVal, 6 * g.m + 3 * g.n, np.float64)
    _gn_free(arr, g)
    L.orc_set_extended(0)
    if rc < 0:
        raise ValueError(f"orc_gn_polish -> {rc}")
    return st, obj, gn, hv[:iters], rc
