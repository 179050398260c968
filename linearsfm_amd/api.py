"""ctypes binding of liblsfm_hip.so (C ABI: include/lsfm.h).

This module is plumbing only: every numerical step runs in the hand-written HIP library.  There is no CPU
fallback -- importing works anywhere, but creating a Context without a usable MI355X (or without the built
library) raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "liblsfm_hip.so")
_LIB = None

LSFM_OK = 0
LSFM_NOT_CONVERGED = 1


class LsfmMap(C.Structure):
    _fields_ = [("Ref", C.c_int), ("FRef", C.c_int), ("m", C.c_int), ("n", C.c_int), ("nU", C.c_int), ("nW", C.c_int),
                ("ScaP", C.c_int), ("Fix", C.c_int), ("Sign", C.c_int), ("FScaP", C.c_int), ("FFix", C.c_int),
                ("stno", C.POINTER(C.c_int)), ("stVal", C.POINTER(C.c_double)),
                ("U", C.POINTER(C.c_double)), ("Ui", C.POINTER(C.c_int)), ("Uj", C.POINTER(C.c_int)),
                ("W", C.POINTER(C.c_double)), ("photo", C.POINTER(C.c_int)), ("feature", C.POINTER(C.c_int)),
                ("V", C.POINTER(C.c_double)), ("FBlock", C.POINTER(C.c_int)), ("pose_origin", C.POINTER(C.c_int))]


class LsfmStats(C.Structure):
    _fields_ = [("t_total_ms", C.c_double), ("t_transform_ms", C.c_double), ("t_join_ms", C.c_double),
                ("t_schur_ms", C.c_double), ("t_pcg_ms", C.c_double), ("t_backsub_ms", C.c_double),
                ("pcg_iterations", C.c_long), ("spmv_launches", C.c_long), ("spmv_ms", C.c_double),
                ("spmv_bytes", C.c_double), ("spmv_nnzb_upper_last", C.c_long), ("spmv_rows_last", C.c_long),
                ("max_rel_residual", C.c_double), ("levels", C.c_int), ("joins", C.c_int), ("transforms", C.c_int),
                ("not_converged", C.c_int), ("schur_launches", C.c_long), ("trf_launches", C.c_long),
                ("schur_ms", C.c_double), ("schur_bytes", C.c_double), ("trf_ms", C.c_double), ("trf_bytes", C.c_double),
                ("schur_flops", C.c_double), ("upload_ms", C.c_double), ("attempts", C.c_int),
                ("s_digest", C.c_ulonglong), ("factor_digest", C.c_ulonglong), ("dist_solves", C.c_int), ("dist_work_total", C.c_double), ("dist_work_shared", C.c_double), ("refactor_mismatch", C.c_int), ("s_rebuild_mismatch", C.c_int), ("small_levels", C.c_int), ("t_small_ms", C.c_double)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


class LsfmError(RuntimeError):
    pass


# include/lsfm.h lsfm_allreduce_fn: (user, offset_bytes, count, dtype, hip_stream) -> 0 on success
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_size_t, C.c_int, C.c_void_p)
LSFM_DTYPE_F64, LSFM_DTYPE_I64 = 0, 1


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LsfmError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        P = C.POINTER
        dp, ip, vp = P(C.c_double), P(C.c_int), C.c_void_p
        L.lsfm_context_create.argtypes = [C.c_int, C.c_size_t, P(vp)]
        L.lsfm_context_destroy.argtypes = [vp]
        L.lsfm_context_destroy.restype = None
        L.lsfm_set_pcg.argtypes = [vp, C.c_double, C.c_int]
        L.lsfm_set_precision.argtypes = [vp, C.c_int]
        L.lsfm_set_small_solve.argtypes = [vp, C.c_int]
        L.lsfm_set_spmv_variant.argtypes = [vp, C.c_int]
        L.lsfm_last_error.argtypes = [vp]
        L.lsfm_last_error.restype = C.c_char_p
        L.lsfm_stream.argtypes = [vp]
        L.lsfm_stream.restype = vp
        L.lsfm_map_release.argtypes = [P(LsfmMap)]
        L.lsfm_map_release.restype = None
        L.lsfm_transform_stereo.argtypes = [vp, P(LsfmMap), C.c_int, P(LsfmMap)]
        L.lsfm_transform_mono.argtypes = [vp, P(LsfmMap), C.c_int, C.c_int, C.c_int, P(LsfmMap)]
        L.lsfm_join_stereo.argtypes = [vp, P(LsfmMap), P(LsfmMap), P(LsfmMap), dp, dp]
        L.lsfm_join_mono.argtypes = [vp, P(LsfmMap), P(LsfmMap), P(LsfmMap), dp, dp]
        L.lsfm_solve_stereo.argtypes = [vp, dp, dp, dp, dp, dp, dp, ip, ip, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.lsfm_solve_mono.argtypes = [vp, dp, dp, dp, dp, dp, dp, ip, ip, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, dp]
        L.lsfm_tree_upload.argtypes = [vp, P(LsfmMap), C.c_int, C.c_int, P(vp)]
        L.lsfm_tree_run.argtypes = [vp, vp, P(LsfmStats)]
        L.lsfm_tree_set_final_reanchor.argtypes = [vp, C.c_int]
        L.lsfm_tree_set_plans.argtypes = [vp, C.c_int]
        L.lsfm_tree_export_size.argtypes = [vp, vp]
        L.lsfm_tree_export_size.restype = C.c_size_t
        L.lsfm_tree_export_dev.argtypes = [vp, vp, vp, C.c_size_t]
        L.lsfm_packed_size.argtypes = [vp]
        L.lsfm_packed_size.restype = C.c_size_t
        L.lsfm_tree_upload_dev.argtypes = [vp, P(vp), C.c_int, C.c_int, P(vp)]
        L.lsfm_tree_reload_dev.argtypes = [vp, vp, P(vp), C.c_int]
        L.lsfm_tree_set_comm.argtypes = [vp, C.c_int, C.c_int, ALLREDUCE_FN, vp, vp, C.c_size_t]
        L.lsfm_tree_set_comm_blocks.argtypes = [vp, C.c_int]
        L.lsfm_tree_export_slice_sizes.argtypes = [vp, vp, C.c_int, P(C.c_size_t)]
        L.lsfm_tree_export_slice_dev.argtypes = [vp, vp, C.c_int, C.c_int, vp, C.c_size_t]
        L.lsfm_tree_download.argtypes = [vp, vp, P(LsfmMap)]
        L.lsfm_tree_set_stop_level.argtypes = [vp, C.c_int]
        L.lsfm_tree_node_count.argtypes = [vp, vp]
        L.lsfm_tree_download_node.argtypes = [vp, vp, C.c_int, P(LsfmMap)]
        L.lsfm_tree_download_state.argtypes = [vp, vp, ip, ip, ip, dp, C.c_size_t]
        L.lsfm_tree_free.argtypes = [vp, vp]
        L.lsfm_tree_free.restype = None
        L.lsfm_divide_conquer.argtypes = [vp, P(LsfmMap), C.c_int, C.c_int, P(LsfmMap), P(LsfmStats)]
        L.lsfm_read_localmap.argtypes = [C.c_char_p, C.c_int, P(LsfmMap)]
        L.lsfm_write_localmap.argtypes = [C.c_char_p, C.c_int, P(LsfmMap)]
        L.lsfm_read_localmaps.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, P(LsfmMap), P(C.c_int)]
        L.lsfm_write_mapset.argtypes = [C.c_char_p, P(LsfmMap), C.c_int, C.c_int]
        L.lsfm_mapset_info.argtypes = [C.c_char_p, P(C.c_int), P(C.c_int)]
        L.lsfm_read_mapset.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, P(LsfmMap)]
        L.lsfm_save_state_bin.argtypes = [C.c_char_p, P(C.c_double), P(C.c_int), C.c_int]
        L.lsfm_save_state.argtypes = [C.c_char_p, dp, ip, C.c_int]
        L.lsfm_save_poses.argtypes = [C.c_char_p, C.c_char_p, ip, dp, C.c_int]
        L.lsfm_schur_pattern.argtypes = [vp, ip, ip, ip, ip, C.c_int, C.c_int, C.c_int, C.c_int, ip, ip, C.c_int, ip]
        L.lsfm_symbolic_analyse.argtypes = [C.c_int, ip, ip, ip, C.c_int, ip, ip, ip, C.c_int, ip, dp]
        L.lsfm_inverse_v.argtypes = [vp, dp, C.c_int, C.c_int]
        L.lsfm_gn_polish.argtypes = [vp, P(LsfmMap), C.c_int, C.c_int, P(LsfmMap), C.c_int, dp, dp, ip]
        L.lsfm_solve_features.argtypes = [vp, dp, dp, dp, dp, dp, dp, C.c_int, C.c_int, ip, ip]
        L.lsfm_spmv_bench.argtypes = [vp, C.c_int, ip, ip, dp, dp, dp, C.c_int, dp, dp]
        L.lsfm_wstream_bench.argtypes = [vp, C.c_longlong, C.c_int, C.c_int, dp]
        L.lsfm_selftest_prims.argtypes = [vp, C.c_int, C.c_uint]
        _LIB = L
    return _LIB


EXPORTS = ["lsfm_context_create", "lsfm_context_destroy", "lsfm_set_pcg", "lsfm_set_precision", "lsfm_set_small_solve", "lsfm_set_spmv_variant", "lsfm_last_error", "lsfm_stream",
           "lsfm_map_release", "lsfm_transform_stereo", "lsfm_transform_mono", "lsfm_join_stereo", "lsfm_join_mono",
           "lsfm_solve_stereo", "lsfm_solve_mono", "lsfm_tree_upload", "lsfm_tree_run", "lsfm_tree_set_final_reanchor",
           "lsfm_tree_download", "lsfm_tree_set_stop_level", "lsfm_tree_node_count", "lsfm_tree_download_node", "lsfm_tree_download_state", "lsfm_tree_set_plans", "lsfm_tree_export_size", "lsfm_tree_export_dev", "lsfm_packed_size",
           "lsfm_tree_upload_dev", "lsfm_tree_reload_dev", "lsfm_tree_set_comm", "lsfm_tree_set_comm_blocks", "lsfm_tree_export_slice_sizes", "lsfm_tree_export_slice_dev",
           "lsfm_tree_free", "lsfm_divide_conquer", "lsfm_read_localmap", "lsfm_read_localmaps", "lsfm_write_localmap", "lsfm_write_mapset", "lsfm_mapset_info", "lsfm_mapset_stamp", "lsfm_read_mapset", "lsfm_save_state_bin", "lsfm_save_state", "lsfm_save_poses", "lsfm_gn_polish",
           "lsfm_spmv_bench", "lsfm_wstream_bench", "lsfm_selftest_prims", "lsfm_schur_pattern", "lsfm_symbolic_analyse", "lsfm_inverse_v", "lsfm_solve_features"]


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype).reshape(-1)


def _ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


class HostMap:
    """A map in the reference layout whose arrays are owned by numpy; `.c` is the lsfm_map view."""

    def __init__(self, d):
        g = d if isinstance(d, dict) else d.__dict__
        self.stno = _c(g["stno"], np.int32); self.stVal = _c(g["stVal"], np.float64)
        self.U = _c(g["U"], np.float64); self.Ui = _c(g["Ui"], np.int32); self.Uj = _c(g["Uj"], np.int32)
        self.W = _c(g["W"], np.float64); self.photo = _c(g["photo"], np.int32); self.feature = _c(g["feature"], np.int32)
        self.V = _c(g["V"], np.float64); self.FBlock = _c(g["FBlock"], np.int32)
        c = LsfmMap()
        c.Ref = int(g["Ref"]); c.FRef = int(g.get("FRef", g["Ref"])); c.m = int(g["m"]); c.n = int(g["n"])
        c.nU = len(self.Ui); c.nW = len(self.photo)
        c.ScaP = int(g.get("ScaP", 0)); c.Fix = int(g.get("Fix", 0)); c.Sign = int(g.get("Sign", 1))
        c.FScaP = int(g.get("FScaP", c.ScaP)); c.FFix = int(g.get("FFix", c.Fix))
        c.stno = _ptr(self.stno, C.c_int); c.stVal = _ptr(self.stVal, C.c_double)
        c.U = _ptr(self.U, C.c_double); c.Ui = _ptr(self.Ui, C.c_int); c.Uj = _ptr(self.Uj, C.c_int)
        c.W = _ptr(self.W, C.c_double); c.photo = _ptr(self.photo, C.c_int); c.feature = _ptr(self.feature, C.c_int)
        c.V = _ptr(self.V, C.c_double); c.FBlock = _ptr(self.FBlock, C.c_int)
        if g.get("pose_origin") is not None:
            self.pose_origin = _c(g["pose_origin"], np.int32)
            c.pose_origin = _ptr(self.pose_origin, C.c_int)
        self.c = c


def _arr(ptr, n, dtype):
    if n == 0:
        return np.zeros(0, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def map_to_dict(g: LsfmMap, release=True):
    r = 6 * g.m + 3 * g.n
    d = dict(Ref=g.Ref, FRef=g.FRef, m=g.m, n=g.n, nU=g.nU, nW=g.nW, ScaP=g.ScaP, Fix=g.Fix, Sign=g.Sign,
             FScaP=g.FScaP, FFix=g.FFix,
             stno=_arr(g.stno, r, np.int32), stVal=_arr(g.stVal, r, np.float64),
             U=_arr(g.U, 36 * g.nU, np.float64).reshape(-1, 36), Ui=_arr(g.Ui, g.nU, np.int32), Uj=_arr(g.Uj, g.nU, np.int32),
             W=_arr(g.W, 18 * g.nW, np.float64).reshape(-1, 18), photo=_arr(g.photo, g.nW, np.int32),
             feature=_arr(g.feature, g.nW, np.int32), V=_arr(g.V, 9 * g.n, np.float64).reshape(-1, 9),
             FBlock=_arr(g.FBlock, g.n, np.int32))
    if g.pose_origin:
        d["pose_origin"] = _arr(g.pose_origin, g.m, np.int32)
    if release:
        lib().lsfm_map_release(C.byref(g))
    return d


class Context:
    """One per GPU.  Raises LsfmError when no HIP device is usable (the library has no CPU path)."""

    def __init__(self, device=0, arena_bytes=0):
        self._h = C.c_void_p()
        rc = lib().lsfm_context_create(int(device), int(arena_bytes), C.byref(self._h))
        if rc != 0:
            self._h = None
            raise LsfmError(f"lsfm_context_create failed (rc={rc}): no usable HIP device -- there is no CPU fallback")

    def close(self):
        if self._h:
            lib().lsfm_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc < 0:
            raise LsfmError(f"{what} failed (rc={rc}): {lib().lsfm_last_error(self._h).decode()}")
        return rc

    def set_pcg(self, rel_tol=1e-12, max_steps=0):
        """Stopping rule of the refinement (relative residual) and the most steps a system may take (0: the default, 50)."""
        self._check(lib().lsfm_set_pcg(self._h, float(rel_tol), int(max_steps)), "lsfm_set_pcg")

    def set_precision(self, mixed):
        """False: fp64 throughout.  True: the Cholesky preconditioner kept and applied in fp32, residual correction in fp64."""
        self._check(lib().lsfm_set_precision(self._h, 1 if mixed else 0), "lsfm_set_precision")

    def set_small_solve(self, max_poses=5):
        """Levels whose camera systems have at most `max_poses` poses are solved by the one-launch dense path (0: none; at most 16;
        default 5); the others, and everything in feature-sharded runs, by the sparse pipeline.  The tests compare the two."""
        self._check(lib().lsfm_set_small_solve(self._h, int(max_poses)), "lsfm_set_small_solve")

    def set_spmv_variant(self, variant):
        """0: by size (default); 1: always the kernel that streams the upper blocks once; 2: always the row-sorted list."""
        self._check(lib().lsfm_set_spmv_variant(self._h, int(variant)), "lsfm_set_spmv_variant")

    def stream(self):
        return lib().lsfm_stream(self._h)

    def tree_reload_dev(self, tree, dev_ptrs):
        """New values (same structure) for the resident inputs of a tree made by tree_upload_dev; plans are kept."""
        arr = (C.c_void_p * len(dev_ptrs))(*[C.c_void_p(int(p)) for p in dev_ptrs])
        self._check(lib().lsfm_tree_reload_dev(self._h, tree, arr, len(dev_ptrs)), "lsfm_tree_reload_dev")

    # ---- the reference's three scheduler-facing methods -------------------------------------------------
    def transform(self, d, mono, Ref, ScaP=0, Fix=0):
        hm = HostMap(d)
        out = LsfmMap()
        if mono:
            rc = lib().lsfm_transform_mono(self._h, C.byref(hm.c), int(Ref), int(ScaP), int(Fix), C.byref(out))
        else:
            rc = lib().lsfm_transform_stereo(self._h, C.byref(hm.c), int(Ref), C.byref(out))
        self._check(rc, "lsfm_transform")
        return map_to_dict(out)

    def join(self, dEnd, dCur, mono):
        """Returns (joint dict with solved state, eP, eF, rc)."""
        he, hc = HostMap(dEnd), HostMap(dCur)
        out = LsfmMap()
        m = he.c.m + hc.c.m - (2 if mono else 0)
        eP = np.zeros(6 * m)
        eF = np.zeros(3 * (he.c.n + hc.c.n))
        fn = lib().lsfm_join_mono if mono else lib().lsfm_join_stereo
        rc = self._check(fn(self._h, C.byref(he.c), C.byref(hc.c), C.byref(out), _ptr(eP, C.c_double), _ptr(eF, C.c_double)),
                         "lsfm_join")
        j = map_to_dict(out)
        return j, eP, eF[:3 * j["n"]], rc

    def solve(self, j, eP, eF, mono, sa=None, x0=None):
        m, n = int(j["m"]), int(j["n"])
        st = np.zeros(6 * m + 3 * n)
        U = _c(j["U"], np.float64); W = _c(j["W"], np.float64); V = _c(j["V"], np.float64)
        Ui = _c(j["Ui"], np.int32); Uj = _c(j["Uj"], np.int32); ph = _c(j["photo"], np.int32); fe = _c(j["feature"], np.int32)
        eP = _c(eP, np.float64); eF = _c(eF, np.float64)
        x0p = _ptr(_c(x0, np.float64), C.c_double) if x0 is not None else None
        d, i = C.c_double, C.c_int
        if mono:
            rc = lib().lsfm_solve_mono(self._h, _ptr(st, d), _ptr(eF, d), _ptr(eP, d), _ptr(U, d), _ptr(W, d), _ptr(V, d),
                                       _ptr(Ui, i), _ptr(Uj, i), _ptr(ph, i), _ptr(fe, i), m, n, len(Ui), len(ph),
                                       sa[0], sa[1], sa[2], sa[3], sa[4], x0p)
        else:
            rc = lib().lsfm_solve_stereo(self._h, _ptr(st, d), _ptr(eF, d), _ptr(eP, d), _ptr(U, d), _ptr(W, d), _ptr(V, d),
                                         _ptr(Ui, i), _ptr(Uj, i), _ptr(ph, i), _ptr(fe, i), m, n, len(Ui), len(ph), x0p)
        self._check(rc, "lsfm_solve")
        return st, rc

    # ---- the scheduler --------------------------------------------------------------------------------
    def tree_upload(self, maps, mono):
        hms = [HostMap(m) for m in maps]
        arr = (LsfmMap * len(hms))(*[h.c for h in hms])
        t = C.c_void_p()
        self._check(lib().lsfm_tree_upload(self._h, arr, len(hms), int(mono), C.byref(t)), "lsfm_tree_upload")
        return t

    def tree_run(self, tree):
        st = LsfmStats()
        rc = self._check(lib().lsfm_tree_run(self._h, tree, C.byref(st)), "lsfm_tree_run")
        return st.as_dict(), rc

    def tree_download(self, tree):
        out = LsfmMap()
        self._check(lib().lsfm_tree_download(self._h, tree, C.byref(out)), "lsfm_tree_download")
        return map_to_dict(out)

    # ---- level checkpoint / resume (include/lsfm.h) ------------------------------------------------------
    def tree_set_stop_level(self, tree, levels):
        """The next runs of `tree` end after `levels` tree levels (0: the whole tree)."""
        if lib().lsfm_tree_set_stop_level(tree, int(levels)):
            raise LsfmError("lsfm_tree_set_stop_level: bad argument")

    def tree_node_count(self, tree):
        return int(lib().lsfm_tree_node_count(self._h, tree))

    def tree_download_node(self, tree, k):
        """Node k of the level the last run ended at, as a map dict that tree_upload takes back (FRef, FScaP, FFix, pose_origin)."""
        out = LsfmMap()
        self._check(lib().lsfm_tree_download_node(self._h, tree, int(k), C.byref(out)), "lsfm_tree_download_node")
        return map_to_dict(out)

    def tree_download_state(self, tree):
        """(m, n, stno, stVal) of the final map: the state vector without the information blocks."""
        m, n = C.c_int(0), C.c_int(0)
        self._check(lib().lsfm_tree_download_state(self._h, tree, C.byref(m), C.byref(n), None, None, 0), "lsfm_tree_download_state")
        r = 6 * m.value + 3 * n.value
        stno = np.zeros(r, np.int32); stVal = np.zeros(r)
        self._check(lib().lsfm_tree_download_state(self._h, tree, C.byref(m), C.byref(n), _ptr(stno, C.c_int), _ptr(stVal, C.c_double), r),
                    "lsfm_tree_download_state")
        return m.value, n.value, stno, stVal

    def tree_free(self, tree):
        lib().lsfm_tree_free(self._h, tree)

    def tree_set_plans(self, tree, on):
        lib().lsfm_tree_set_plans(tree, int(on))

    def tree_set_final_reanchor(self, tree, on):
        lib().lsfm_tree_set_final_reanchor(tree, int(on))

    # ---- device-resident hand-off of a tree node (multi-GPU sub-tree sharding) ---------------------------
    def tree_export_size(self, tree):
        return int(lib().lsfm_tree_export_size(self._h, tree))

    def tree_export_dev(self, tree, dev_ptr, cap):
        """Packs the final map of a finished tree into caller-owned device memory (e.g. a torch uint8 CUDA tensor)."""
        self._check(lib().lsfm_tree_export_dev(self._h, tree, C.c_void_p(int(dev_ptr)), int(cap)), "lsfm_tree_export_dev")

    def tree_upload_dev(self, dev_ptrs, mono):
        """A new tree whose resident inputs are the packed maps at the given device addresses (no host copy)."""
        arr = (C.c_void_p * len(dev_ptrs))(*[C.c_void_p(int(p)) for p in dev_ptrs])
        t = C.c_void_p()
        self._check(lib().lsfm_tree_upload_dev(self._h, arr, len(dev_ptrs), int(mono), C.byref(t)), "lsfm_tree_upload_dev")
        return t

    # ---- feature-sharded joins (the top of the tree over several GPUs) -----------------------------------
    def tree_export_slice_sizes(self, tree, nslices):
        sizes = (C.c_size_t * nslices)()
        self._check(lib().lsfm_tree_export_slice_sizes(self._h, tree, int(nslices), sizes), "lsfm_tree_export_slice_sizes")
        return [int(v) for v in sizes]

    def tree_export_slice_dev(self, tree, nslices, slice_, dev_ptr, cap):
        """Pack `slice_` (features with feat_id % nslices == slice_, all poses and U blocks) of a finished tree's final map."""
        self._check(lib().lsfm_tree_export_slice_dev(self._h, tree, int(nslices), int(slice_), C.c_void_p(int(dev_ptr)), int(cap)),
                    "lsfm_tree_export_slice_dev")

    def tree_set_comm(self, tree, rank, world, fn, dev_ptr, dev_bytes):
        """fn: an ALLREDUCE_FN instance (the caller keeps it alive as long as the tree runs); None / world <= 1: off."""
        cb = fn if fn is not None else C.cast(None, ALLREDUCE_FN)
        self._check(lib().lsfm_tree_set_comm(tree, int(rank), int(world), cb, None, C.c_void_p(int(dev_ptr) if dev_ptr else None),
                                             int(dev_bytes)), "lsfm_tree_set_comm")

    def tree_set_comm_blocks(self, tree, block_maps):
        """block_maps = 2^k > 0: the pose-side factorisation of the feature-sharded tree is distributed by block ownership; 0: replicated."""
        self._check(lib().lsfm_tree_set_comm_blocks(tree, int(block_maps)), "lsfm_tree_set_comm_blocks")

    def divide_conquer(self, maps, mono, final_reanchor=True):
        t = self.tree_upload(maps, mono)
        if not final_reanchor:
            lib().lsfm_tree_set_final_reanchor(t, 0)
        try:
            stats, rc = self.tree_run(t)
            out = self.tree_download(t)
        finally:
            self.tree_free(t)
        return out, stats, rc

    def gn_polish(self, maps, mono, G, iters):
        """lsfm_gn_polish: `iters` Gauss-Newton steps of the map-joining objective over all local maps from the global state G (a map
        dict: stno, stVal, m, n, Ref; Mono: ScaP, Fix; e.g. what divide_conquer returned).  No reference counterpart (parity unpinned).
        Returns (stVal, obj[iters + 1], gnorm[iters + 1], halvings[iters], rc)."""
        hms = [HostMap(d) for d in maps]
        arr = (LsfmMap * len(hms))(*[h.c for h in hms])
        x = LsfmMap()
        stno = _c(G["stno"], np.int32)
        st = np.array(np.asarray(G["stVal"], np.float64), copy=True)
        x.m, x.n, x.Ref, x.FRef = int(G["m"]), int(G["n"]), int(G["Ref"]), int(G.get("FRef", G["Ref"]))
        x.ScaP, x.Fix, x.Sign = int(G.get("ScaP", 0)), int(G.get("Fix", 0)), int(G.get("Sign", 1))
        x.stno, x.stVal = _ptr(stno, C.c_int), _ptr(st, C.c_double)
        org = None
        if G.get("pose_origin") is not None:
            org = _c(G["pose_origin"], np.int32)
            x.pose_origin = _ptr(org, C.c_int)
        obj, gn, hv = np.zeros(iters + 1), np.zeros(iters + 1), np.zeros(max(iters, 1), np.int32)
        rc = lib().lsfm_gn_polish(self._h, arr, len(hms), int(mono), C.byref(x), int(iters), _ptr(obj, C.c_double), _ptr(gn, C.c_double), _ptr(hv, C.c_int))
        if rc < 0:
            self._check(rc, "lsfm_gn_polish")
        return st, obj, gn, hv[:iters], rc

    def inverse_v(self, V):
        """lsfm_inverse_v (the reference's pba_inverseV, Imp.cpp:3022): V^-1 of the 3x3 feature blocks, [n, 9]."""
        IV = np.array(np.asarray(V, np.float64).reshape(-1), copy=True)
        n = IV.size // 9
        self._check(lib().lsfm_inverse_v(self._h, _ptr(IV, C.c_double), 0, n), "lsfm_inverse_v")
        return IV.reshape(n, 9)

    def solve_features(self, j, IV, eb, dpa):
        """lsfm_solve_features (the reference's pba_solveFeatures, Imp.cpp:2980): the features' back-substitution for the given pose
        values dpa[6m] on the system of joint-map dict j."""
        m, n = int(j["m"]), int(j["n"])
        W = _c(j["W"], np.float64); ph = _c(j["photo"], np.int32); fe = _c(j["feature"], np.int32)
        cnt = np.bincount(fe, minlength=n).astype(np.int32)
        IV = _c(IV, np.float64); eb = _c(eb, np.float64); dpa = _c(dpa, np.float64)
        dpb = np.zeros(3 * n)
        self._check(lib().lsfm_solve_features(self._h, _ptr(W, C.c_double), _ptr(IV, C.c_double), None, _ptr(eb, C.c_double), _ptr(dpa, C.c_double),
                                              _ptr(dpb, C.c_double), m, n, _ptr(cnt, C.c_int), _ptr(ph, C.c_int)), "lsfm_solve_features")
        return dpb

    def schur_pattern(self, j):
        """Upper block pattern (rowptr, colidx) of the camera system of a joint map dict, as the device builds it."""
        m, n = int(j["m"]), int(j["n"])
        Ui = _c(j["Ui"], np.int32); Uj = _c(j["Uj"], np.int32); ph = _c(j["photo"], np.int32); fe = _c(j["feature"], np.int32)
        cap = m * (m + 1) // 2
        rowptr = np.zeros(m + 1, np.int32)
        colidx = np.zeros(max(cap, 1), np.int32)
        nnzb = C.c_int(0)
        self._check(lib().lsfm_schur_pattern(self._h, _ptr(Ui, C.c_int), _ptr(Uj, C.c_int), _ptr(ph, C.c_int), _ptr(fe, C.c_int), m, n,
                                             len(Ui), len(ph), _ptr(rowptr, C.c_int), _ptr(colidx, C.c_int), cap, C.byref(nnzb)),
                    "lsfm_schur_pattern")
        return rowptr, colidx[:nnzb.value].copy()

    def spmv_bench(self, rowptr, colidx, val, x, reps=20):
        rowptr = _c(rowptr, np.int32); colidx = _c(colidx, np.int32); val = _c(val, np.float64); x = _c(x, np.float64)
        m = len(rowptr) - 1
        y = np.zeros(6 * m)
        ms, by = C.c_double(), C.c_double()
        self._check(lib().lsfm_spmv_bench(self._h, m, _ptr(rowptr, C.c_int), _ptr(colidx, C.c_int), _ptr(val, C.c_double),
                                          _ptr(x, C.c_double), _ptr(y, C.c_double), int(reps), C.byref(ms), C.byref(by)),
                    "lsfm_spmv_bench")
        return y, ms.value, by.value


def _wstream(self, nblocks, mode, reps=10):
    """ms per launch of the W access-pattern copy (lsfm_wstream_bench)."""
    ms = C.c_double()
    self._check(lib().lsfm_wstream_bench(self._h, int(nblocks), int(mode), int(reps), C.byref(ms)), "lsfm_wstream_bench")
    return ms.value


Context.wstream_bench = _wstream


def _selftest_prims(self, cases=64, seed=1):
    """The library's own fill / small-copy kernels against the host (lsfm_selftest_prims); raises LsfmError on the first mismatch."""
    self._check(lib().lsfm_selftest_prims(self._h, int(cases), int(seed)), "lsfm_selftest_prims")


Context.selftest_prims = _selftest_prims


def symbolic_analyse(rowptr, colidx, origin=None, reps=1):
    """Host-only: ordering + symbolic block Cholesky of a camera system's upper block pattern (no device needed).
    Returns dict(perm, colptr, rowidx, info, ms)."""
    rowptr = _c(rowptr, np.int32); colidx = _c(colidx, np.int32)
    m = len(rowptr) - 1
    org = _c(origin, np.int32) if origin is not None else None
    info = np.zeros(8, np.int32)
    perm = np.zeros(m, np.int32); colptr = np.zeros(m + 1, np.int32)
    ms = C.c_double()
    lib().lsfm_symbolic_analyse(m, _ptr(rowptr, C.c_int), _ptr(colidx, C.c_int), _ptr(org, C.c_int) if org is not None else None, 1,
                                _ptr(perm, C.c_int), _ptr(colptr, C.c_int), None, 0, _ptr(info, C.c_int), None)
    rowidx = np.zeros(max(int(info[0]), 1), np.int32)
    rc = lib().lsfm_symbolic_analyse(m, _ptr(rowptr, C.c_int), _ptr(colidx, C.c_int), _ptr(org, C.c_int) if org is not None else None, int(reps),
                                     _ptr(perm, C.c_int), _ptr(colptr, C.c_int), _ptr(rowidx, C.c_int), len(rowidx), _ptr(info, C.c_int), C.byref(ms))
    if rc:
        raise LsfmError(f"lsfm_symbolic_analyse failed (rc={rc}): malformed pattern")
    return dict(perm=perm, colptr=colptr, rowidx=rowidx[:int(info[0])], info=info, ms=ms.value)


LSFM_NODE_MAGIC = 1279870541  # first token of the tree-node trailer of a local-map file (lsfm_io.cpp)


def read_localmap(path, mono):
    g = LsfmMap()
    rc = lib().lsfm_read_localmap(str(path).encode(), int(mono), C.byref(g))
    if rc:
        raise LsfmError(f"lsfm_read_localmap({path}) failed (rc={rc})")
    return map_to_dict(g)


def write_localmap(path, d, mono):
    """A map dict (local map, joint map or final map with its information matrix) in the local-map text format."""
    hm = HostMap(d)
    rc = lib().lsfm_write_localmap(str(path).encode(), int(mono), C.byref(hm.c))
    if rc:
        raise LsfmError(f"lsfm_write_localmap({path}) failed (rc={rc})")


def write_mapset(path, maps, mono):
    """Binary cache of a set of maps (list of map dicts): one file, read back bit for bit by read_mapset."""
    hms = [HostMap(d) for d in maps]
    arr = (LsfmMap * len(hms))(*[h.c for h in hms])
    rc = lib().lsfm_write_mapset(str(path).encode(), arr, len(hms), int(mono))
    if rc:
        raise LsfmError(f"lsfm_write_mapset({path}) failed (rc={rc})")


def mapset_info(path):
    """(N, mono) of a binary cache; None when the file is missing or not a cache."""
    n, mono = C.c_int(0), C.c_int(0)
    if lib().lsfm_mapset_info(str(path).encode(), C.byref(n), C.byref(mono)):
        return None
    return n.value, bool(mono.value)


def read_mapset(path, mono, first=0, count=None, threads=0):
    info = mapset_info(path)
    if info is None:
        raise LsfmError(f"{path}: not a map-set cache")
    if count is None:
        count = info[0] - first
    arr = (LsfmMap * max(count, 1))()
    rc = lib().lsfm_read_mapset(str(path).encode(), int(mono), int(first), int(count), int(threads), arr)
    if rc:
        raise LsfmError(f"lsfm_read_mapset({path}) failed (rc={rc})")
    return [map_to_dict(arr[k]) for k in range(count)]


def read_localmaps(directory, count, mono, first=1, threads=0):
    """localmap_<first>.txt ... of a directory, parsed on `threads` host threads (0: one per core)."""
    arr = (LsfmMap * count)()
    bad = C.c_int(0)
    rc = lib().lsfm_read_localmaps(str(directory).encode(), int(first), int(count), int(mono), int(threads), arr, C.byref(bad))
    if rc:
        raise LsfmError(f"lsfm_read_localmaps({directory}) failed at localmap_{bad.value}.txt (rc={rc})")
    return [map_to_dict(arr[k]) for k in range(count)]
