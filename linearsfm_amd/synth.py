"""Seeded synthetic local-map sets in the reference's ``localmap_k.txt`` layout.

The real datasets (RS90_C, RS468_C, NC3500_C ...) are Google-Drive links only
(/root/reference/DataForC/*/...Dataset.txt:1), so every measurement and golden vector is made from
local maps written in the reference's own input format:

    Stereo  (lmj_readInformationStereo, LinearSFMImp.cpp:3044-3132)
        Ref, r, r x (stno stVal), m, n, nU, 36*nU U, nU Ui, nU Uj, nW, 18*nW W, nW photo, nW feature,
        9*n V, n FBlock
    Mono    (lmj_readInformationMono,   LinearSFMImp.cpp:6660-6754)
        Ref, ScaP, Fix, Sign, then as above from ``r`` on.

State labels: ``stno <= 0`` -> pose with id ``-stno`` (6 scalars tx ty tz alpha beta gamma), ``stno > 0`` ->
feature id (3 scalars).  Rotation convention R = Rx(gamma) Ry(beta) Rz(alpha), x_cam = R (x - t)
(LinearSFMImp.cpp:132-143, 421-455).

A Stereo local map k holds frames k, k+1 (Ref = k; state = pose k+1 and the points seen in both);
a Mono local map holds frames k, k+1, k+2 (Ref = k, ScaP = k+1; all three poses are in the state, the Ref
pose and the scale pose's ``Fix`` translation scalar are gauge-fixed and carry zero information rows).
The information matrix is sum J^T Sigma^-1 J of the observations, evaluated at the (perturbed) estimate.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

__all__ = ["LocalMap", "make_stereo_set", "make_mono_set", "write_localmap", "read_localmap", "write_set",
           "CONFIGS", "make_config"]


@dataclass
class LocalMap:
    mono: bool
    Ref: int
    stno: np.ndarray      # int32 [6m+3n]
    stVal: np.ndarray     # f64   [6m+3n]
    m: int
    n: int
    U: np.ndarray         # f64 [nU,36] row-major 6x6
    Ui: np.ndarray        # int32 [nU]
    Uj: np.ndarray
    W: np.ndarray         # f64 [nW,18] row-major 6x3 (pose rows x feature cols)
    photo: np.ndarray     # int32 [nW]
    feature: np.ndarray   # int32 [nW] (sorted)
    V: np.ndarray         # f64 [n,9]
    FBlock: np.ndarray    # int32 [n]
    ScaP: int = 0
    Fix: int = 0
    Sign: int = 1
    FRef: int = field(default=-1)
    FScaP: int = 0
    FFix: int = 0

    def __post_init__(self):
        if self.FRef < 0:
            self.FRef = self.Ref
            self.FScaP = self.ScaP
            self.FFix = self.Fix

    @property
    def nU(self):
        return int(self.Ui.shape[0])

    @property
    def nW(self):
        return int(self.photo.shape[0])


# ----------------------------------------------------------------------------------------------
# rotation helpers (same convention as the reference)
# ----------------------------------------------------------------------------------------------
def rot_ypr(a, b, g):
    """R = Rx(g) Ry(b) Rz(a), row-major 3x3 (LinearSFMImp.cpp:132-143)."""
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    return np.array([[cb * ca, cb * sa, -sb],
                     [sg * sb * ca - cg * sa, sg * sb * sa + cg * ca, sg * cb],
                     [cg * sb * ca + sg * sa, cg * sb * sa - sg * ca, cg * cb]])


def _rot_ypr_many(ang):
    """rot_ypr for an [n,3] array of (a, b, g): [n,3,3], the same expressions element by element."""
    a, b, g = ang[:, 0], ang[:, 1], ang[:, 2]
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    R = np.empty((ang.shape[0], 3, 3))
    R[:, 0, 0] = cb * ca; R[:, 0, 1] = cb * sa; R[:, 0, 2] = -sb
    R[:, 1, 0] = sg * sb * ca - cg * sa; R[:, 1, 1] = sg * sb * sa + cg * ca; R[:, 1, 2] = sg * cb
    R[:, 2, 0] = cg * sb * ca + sg * sa; R[:, 2, 1] = cg * sb * sa - sg * ca; R[:, 2, 2] = cg * cb
    return R


def ypr_from_rot(R):
    """Inverse of rot_ypr away from cos(beta)=0 (LinearSFMImp.cpp:162-177)."""
    b = np.arctan2(-R[0, 2], np.hypot(R[0, 0], R[0, 1]))
    a = np.arctan2(R[0, 1], R[0, 0])
    g = np.arctan2(R[1, 2], R[2, 2])
    return a, b, g


def _drot(a, b, g):
    """dR/da, dR/db, dR/dg by the product rule on Rx Ry Rz."""
    ca, sa, cb, sb, cg, sg = np.cos(a), np.sin(a), np.cos(b), np.sin(b), np.cos(g), np.sin(g)
    RG = np.array([[1, 0, 0], [0, cg, sg], [0, -sg, cg]])
    RB = np.array([[cb, 0, -sb], [0, 1, 0], [sb, 0, cb]])
    RA = np.array([[ca, sa, 0], [-sa, ca, 0], [0, 0, 1]])
    dG = np.array([[0, 0, 0], [0, -sg, cg], [0, -cg, -sg]])
    dB = np.array([[-sb, 0, -cb], [0, 0, 0], [cb, 0, -sb]])
    dA = np.array([[-sa, ca, 0], [-ca, -sa, 0], [0, 0, 0]])
    return RG @ RB @ dA, RG @ dB @ RA, dG @ RB @ RA


# ----------------------------------------------------------------------------------------------
# world
# ----------------------------------------------------------------------------------------------
GOLDEN_ANGLE = 2.399963229728653


def _world(n_frames, new_per_frame, vis, seed, lap=0, home=10, revisit=0.4, depth=(4.0, 12.0), turn=GOLDEN_ANGLE, strip=0, spacing=3.5, skip=0):
    """Camera path + points.  Camera moves in the x-y plane looking along +z, points in a slab above it.

    lap = 0: an open path that never returns (dead reckoning only; global coordinates of a long set are then so weakly
    determined that two fp64 evaluations of the same join tree differ visibly -- kept for the small fixtures).
    lap = P > 0: the path is a sequence of closed laps of P frames that all leave from and return to the same place
    (circles through the origin, every lap in a new direction), like the real sets, whose trajectories revisit
    (README.txt:58-60 names sets of up to 3499 local maps that the reference solved).  Points first seen within
    `home` frames of a lap boundary are re-observed, with probability `revisit`, by the frames at the same phase of the
    NEXT lap: the same feature id then appears in local maps that are a whole lap apart, and the join that brings the two
    laps together closes the loop.
    skip = K > 0 (with lap): lap j also meets lap j + 2^r again for every r <= K with 2^r | j (a skip list over the laps: the
    re-observed points of such a lap are dealt over its link lengths), so that every lap is O(log) links away from the
    first one however long the set is -- a chain of laps linked to their neighbours only lets the dead-reckoning error of
    a long set grow until two fp64 evaluations of the reference path disagree in the 5th digit (Mono, 2048 maps).
    strip = L > 0 (instead of lap): an aerial block -- parallel flight strips of L frames, flown to and fro, `spacing` apart
    (the points of a frame spread +-2.5 across the track: 3.5 leaves 30 % side overlap).  A point whose offset across the
    track puts it inside the NEXT strip's footprint is seen again by the frames of that strip that pass the same place, so
    every strip shares features with its neighbour along its whole length (a grid, not a chain with closures).  The
    camera keeps its orientation on the way back (a nadir camera flown backwards): poses stay away from the +-pi yaw seam,
    where two fp64 evaluations may legitimately print angles 2 pi apart."""
    rng1 = np.random.default_rng(seed + 1)
    rng2 = np.random.default_rng(seed + 2)
    i = np.arange(n_frames)
    step = 0.5
    heading = 0.6 * np.sin(2 * np.pi * i / 257.0) + 0.25 * np.sin(2 * np.pi * i / 61.0)
    pos = np.zeros((n_frames, 3))
    if strip > 0:
        js, ps = i // strip, i % strip
        pos[:, 0] = step * np.where(js % 2 == 0, ps, strip - 1 - ps)
        pos[:, 1] = spacing * js
    elif lap > 0:
        R = step / (2.0 * np.sin(np.pi / lap))             # chord between consecutive frames = step
        psi = turn * (i // lap)
        phi = psi + np.pi + 2 * np.pi * (i % lap) / lap
        pos[:, 0] = R * (np.cos(psi) + np.cos(phi))
        pos[:, 1] = R * (np.sin(psi) + np.sin(phi))
    else:
        pos[1:, 0] = np.cumsum(step * np.cos(heading[:-1]))
        pos[1:, 1] = np.cumsum(step * np.sin(heading[:-1]))
    pos[:, 2] = 0.3 * np.sin(2 * np.pi * i / 97.0)

    def smooth(scale, period, phase):
        return scale * np.sin(2 * np.pi * i / period + phase) + 0.3 * scale * np.sin(2 * np.pi * i / (period / 3.7) + 2 * phase)

    ph = rng1.uniform(0, 2 * np.pi, 3)
    ang = np.stack([heading * 0.4 + smooth(0.1, 143.0, ph[0]), smooth(0.15, 89.0, ph[1]), smooth(0.15, 113.0, ph[2])], 1)
    Rw = _rot_ypr_many(ang)                                            # world -> camera

    # points: `new_per_frame` start at every frame s (also vis-1 frames before frame 0 so the first maps are full)
    starts = np.repeat(np.arange(-(vis - 1), n_frames), new_per_frame)
    npts = starts.shape[0]
    mid = np.clip(starts + (vis - 1) / 2.0, 0, n_frames - 1)
    c_mid = np.stack([np.interp(mid, i, pos[:, d]) for d in range(3)], 1)
    off = np.stack([rng2.uniform(-3, 3, npts), rng2.uniform(-2.5, 2.5, npts), rng2.uniform(depth[0], depth[1], npts)], 1)
    pts = c_mid + off
    # second visibility window of the re-observed points (NEVER = none)
    starts2 = np.full(npts, NEVER)
    if lap > 0:
        rng4 = np.random.default_rng(seed + 4)
        ph_s = np.mod(starts, lap)
        near = (starts >= 0) & ((ph_s < home) | (ph_s >= lap - home))
        pick = near & (rng4.uniform(0, 1, npts) < revisit)
        dist = np.ones(npts, np.int64)
        if skip > 0:
            jl = np.maximum(starts, 0) // lap
            tz = np.zeros(npts, np.int64)                    # trailing zeros of the lap index, capped at `skip`
            for r in range(1, skip + 1):
                tz[(jl % (1 << r) == 0) & (jl > 0)] = r
            rsel = (rng4.uniform(0, 1, npts) * (tz + 1)).astype(np.int64)
            dist = 1 << np.minimum(rsel, tz)
            dist[(jl + dist) * lap + lap > n_frames] = 1     # no such lap: the neighbour then
        starts2[pick] = starts[pick] + lap * dist[pick]
    if strip > 0:
        # frame s = j L + p sees the point at the along-track places of its next vis frames; the next strip passes them in
        # reverse order with its frames (j + 1) L + (L - p - vis) ...
        js, ps = starts // strip, starts % strip
        pick = (starts >= 0) & (ps + vis <= strip) & ((js + 1) * strip + strip <= n_frames) & (off[:, 1] >= spacing - 2.5)
        starts2[pick] = (js[pick] + 1) * strip + strip - ps[pick] - vis
    return pos, Rw, starts, starts2, pts


NEVER = -(1 << 40)


class _Visibility:
    """Which points every frame of a window [first, last] sees: `starts` is sorted (points are created frame by frame), and so
    are the second windows of the re-observed points, so a query is two binary searches instead of a pass over all points
    (the pass made the generation of a 65 536-map set quadratic: 17 minutes)."""

    def __init__(self, starts, starts2, vis):
        self.starts, self.vis = starts, vis
        self.idx2 = np.nonzero(starts2 != NEVER)[0]
        self.s2 = starts2[self.idx2]
        if np.any(np.diff(self.s2) < 0):                    # (aerial strips: the second windows run against the first ones)
            o = np.argsort(self.s2, kind="stable")
            self.idx2, self.s2 = self.idx2[o], self.s2[o]
        assert np.all(np.diff(starts) >= 0)

    def __call__(self, first, last):
        lo, hi = last - self.vis + 1, first                 # start <= first and start + vis - 1 >= last
        a = np.arange(np.searchsorted(self.starts, lo, "left"), np.searchsorted(self.starts, hi, "right"))
        b = self.idx2[np.searchsorted(self.s2, lo, "left"):np.searchsorted(self.s2, hi, "right")]
        if b.size == 0:
            return a
        return np.union1d(a, b)                             # ascending point index, like a pass over all points


def _visible(starts, starts2, vis, first, last):
    """points seen by every frame of [first, last] (one query; the generators keep a _Visibility)"""
    return _Visibility(starts, starts2, vis)(first, last)


def _rel_pose(pos, Rw, k, j):
    """pose of frame j expressed in frame k: (t, R) with x_j = R (x_k - t)."""
    t = Rw[k] @ (pos[j] - pos[k])
    R = Rw[j] @ Rw[k].T
    return t, R


# ----------------------------------------------------------------------------------------------
# Stereo
# ----------------------------------------------------------------------------------------------
def make_stereo_set(n_maps, new_per_frame=130, vis=5, seed=0, noise=1e-3, first_id=1, lap=0, home=10, revisit=0.4, depth=(4.0, 12.0),
                    turn=GOLDEN_ANGLE, only=None, strip=0, spacing=3.5, skip=0):
    """n_maps Stereo local maps over n_maps+1 frames (ids first_id..).  ~new_per_frame*(vis-1) features per map.
    lap/home/revisit: see _world (lap = 0: open path).  only = (lo, hi): just the maps lo..hi-1 of the set (each map draws
    its noise from its own generator, so a slice equals the corresponding part of the whole set)."""
    n_frames = n_maps + 1
    pos, Rw, starts, starts2, pts = _world(n_frames, new_per_frame, vis, seed, lap, home, revisit, depth, turn, strip, spacing, skip)
    sinv = np.diag(1.0 / np.array([0.01, 0.01, 0.03]) ** 2)
    seen = _Visibility(starts, starts2, vis)
    maps = []
    for k in range(*(only or (0, n_maps))):
        rng3 = np.random.default_rng([seed + 3, k])
        # points visible in frame k and k+1
        sel = seen(k, k + 1)
        n = sel.shape[0]
        t, R = _rel_pose(pos, Rw, k, k + 1)
        a, b, g = ypr_from_rot(R)
        pose = np.array([t[0], t[1], t[2], a, b, g]) + rng3.normal(0, noise, 6)
        X = (Rw[k] @ (pts[sel] - pos[k]).T).T + rng3.normal(0, noise, (n, 3))
        # information at the estimate
        Re = rot_ypr(*pose[3:])
        dRA, dRB, dRG = _drot(*pose[3:])
        d = X - pose[:3]                                   # [n,3]
        Jp = np.zeros((n, 3, 6))
        Jp[:, :, 0:3] = -Re
        Jp[:, :, 3] = d @ dRA.T
        Jp[:, :, 4] = d @ dRB.T
        Jp[:, :, 5] = d @ dRG.T
        Jx = np.broadcast_to(Re, (n, 3, 3))
        U = np.einsum('nai,ab,nbj->ij', Jp, sinv, Jp).reshape(1, 36)
        W = np.einsum('nai,ab,nbj->nij', Jp, sinv, Jx).reshape(n, 18)
        V = (sinv[None] + np.einsum('nai,ab,nbj->nij', Jx, sinv, Jx)).reshape(n, 9)
        stno = np.concatenate([np.full(6, -(first_id + k + 1)), np.repeat(sel + 1, 3)]).astype(np.int32)
        stVal = np.concatenate([pose, X.reshape(-1)])
        maps.append(LocalMap(mono=False, Ref=first_id + k, stno=stno, stVal=stVal, m=1, n=n,
                             U=U, Ui=np.zeros(1, np.int32), Uj=np.zeros(1, np.int32),
                             W=W, photo=np.zeros(n, np.int32), feature=np.arange(n, dtype=np.int32),
                             V=V, FBlock=np.arange(n, dtype=np.int32)))
    return maps


# ----------------------------------------------------------------------------------------------
# Mono
# ----------------------------------------------------------------------------------------------
def make_mono_set(n_maps, new_per_frame=300, vis=4, seed=0, noise=1e-3, first_id=1, lap=0, home=10, revisit=0.4, depth=(4.0, 12.0),
                  turn=GOLDEN_ANGLE, only=None, strip=0, spacing=3.5, skip=0):
    """n_maps Mono local maps over n_maps+2 frames; map k = frames k,k+1,k+2 (Ref=k, ScaP=k+1).
    lap/home/revisit: see _world (lap = 0: open path); only = (lo, hi): just that slice of the set."""
    n_frames = n_maps + 2
    pos, Rw, starts, starts2, pts = _world(n_frames, new_per_frame, vis, seed, lap, home, revisit, depth, turn, strip, spacing, skip)
    w = 1.0 / (1e-3) ** 2
    seen = _Visibility(starts, starts2, vis)
    maps = []
    for k in range(*(only or (0, n_maps))):
        rng3 = np.random.default_rng([seed + 3, k])
        sel = seen(k, k + 2)
        n = sel.shape[0]
        t1, R1 = _rel_pose(pos, Rw, k, k + 1)
        t2, R2 = _rel_pose(pos, Rw, k, k + 2)
        Fix = int(np.argmax(np.abs(t1)))
        scale = abs(t1[Fix])
        Sign = 1 if t1[Fix] >= 0 else -1
        p1 = np.concatenate([t1 / scale, ypr_from_rot(R1)]) + rng3.normal(0, noise, 6)
        p2 = np.concatenate([t2 / scale, ypr_from_rot(R2)]) + rng3.normal(0, noise, 6)
        p1[Fix] = Sign
        X = (Rw[k] @ (pts[sel] - pos[k]).T).T / scale + rng3.normal(0, noise, (n, 3))
        poses = [np.zeros(6), p1, p2]
        Ublk = np.zeros((3, 6, 6))
        Wblk = np.zeros((3, n, 6, 3))
        V = np.zeros((n, 3, 3))
        for c, p in enumerate(poses):
            Re = rot_ypr(*p[3:])
            dRA, dRB, dRG = _drot(*p[3:])
            d = X - p[:3]
            Y = d @ Re.T                                     # camera coords [n,3]
            iz = 1.0 / Y[:, 2]
            Jpi = np.zeros((n, 2, 3))
            Jpi[:, 0, 0] = iz
            Jpi[:, 0, 2] = -Y[:, 0] * iz * iz
            Jpi[:, 1, 1] = iz
            Jpi[:, 1, 2] = -Y[:, 1] * iz * iz
            dY = np.zeros((n, 3, 6))
            dY[:, :, 0:3] = -Re
            dY[:, :, 3] = d @ dRA.T
            dY[:, :, 4] = d @ dRB.T
            dY[:, :, 5] = d @ dRG.T
            Jp = np.einsum('nab,nbj->naj', Jpi, dY)          # [n,2,6]
            Jx = np.einsum('nab,bj->naj', Jpi, Re)           # [n,2,3]
            Ublk[c] = w * np.einsum('nai,naj->ij', Jp, Jp)
            Wblk[c] = w * np.einsum('nai,naj->nij', Jp, Jx)
            V += w * np.einsum('nai,naj->nij', Jx, Jx)
        # gauge: Ref pose (block 0) and scalar Fix of the scale pose carry no information
        Ublk[1][Fix, :] = 0.0
        Ublk[1][:, Fix] = 0.0
        Wblk[1][:, Fix, :] = 0.0
        U = np.stack([Ublk[1].reshape(36), Ublk[2].reshape(36)])
        Ui = np.array([1, 2], np.int32)
        W = np.stack([Wblk[1], Wblk[2]], 1).reshape(2 * n, 18)          # per feature: pose 1 then pose 2
        photo = np.tile(np.array([1, 2], np.int32), n)
        feature = np.repeat(np.arange(n, dtype=np.int32), 2)
        ids = [first_id + k, first_id + k + 1, first_id + k + 2]
        stno = np.concatenate([np.repeat(-np.array(ids), 6), np.repeat(sel + 1, 3)]).astype(np.int32)
        stVal = np.concatenate([poses[0], p1, p2, X.reshape(-1)])
        maps.append(LocalMap(mono=True, Ref=ids[0], ScaP=ids[1], Fix=Fix, Sign=Sign, stno=stno, stVal=stVal,
                             m=3, n=n, U=U, Ui=Ui, Uj=Ui.copy(), W=W, photo=photo, feature=feature,
                             V=V.reshape(n, 9), FBlock=(2 * np.arange(n)).astype(np.int32)))
    return maps


# ----------------------------------------------------------------------------------------------
# I/O in the reference's text format
# ----------------------------------------------------------------------------------------------
def _fmt(a):
    return " ".join(repr(float(x)) for x in np.asarray(a, dtype=np.float64).reshape(-1))


def _fmti(a):
    return " ".join(str(int(x)) for x in np.asarray(a).reshape(-1))


def write_localmap(path, lm: LocalMap):
    with open(path, "w") as f:
        f.write(f"{lm.Ref}\n")
        if lm.mono:
            f.write(f"{lm.ScaP}\n{lm.Fix}\n{lm.Sign}\n")
        r = 6 * lm.m + 3 * lm.n
        f.write(f"{r}\n")
        f.write("\n".join(f"{int(s)} {float(v)!r}" for s, v in zip(lm.stno, lm.stVal)) + "\n")
        f.write(f"{lm.m}\n{lm.n}\n{lm.nU}\n")
        f.write(_fmt(lm.U) + "\n" + _fmti(lm.Ui) + "\n" + _fmti(lm.Uj) + "\n")
        f.write(f"{lm.nW}\n")
        f.write(_fmt(lm.W) + "\n" + _fmti(lm.photo) + "\n" + _fmti(lm.feature) + "\n")
        f.write(_fmt(lm.V) + "\n" + _fmti(lm.FBlock) + "\n")


def read_localmap(path, mono) -> LocalMap:
    tok = open(path).read().split()
    it = iter(tok)
    nxt = lambda: next(it)
    Ref = int(nxt())
    ScaP = Fix = 0
    Sign = 1
    if mono:
        ScaP, Fix, Sign = int(nxt()), int(nxt()), int(nxt())
    r = int(nxt())
    stno = np.empty(r, np.int32)
    stVal = np.empty(r)
    for i in range(r):
        stno[i] = int(nxt())
        stVal[i] = float(nxt())
    m, n, nU = int(nxt()), int(nxt()), int(nxt())
    U = np.array([float(nxt()) for _ in range(36 * nU)]).reshape(nU, 36)
    Ui = np.array([int(nxt()) for _ in range(nU)], np.int32)
    Uj = np.array([int(nxt()) for _ in range(nU)], np.int32)
    nW = int(nxt())
    W = np.array([float(nxt()) for _ in range(18 * nW)]).reshape(nW, 18)
    photo = np.array([int(nxt()) for _ in range(nW)], np.int32)
    feature = np.array([int(nxt()) for _ in range(nW)], np.int32)
    V = np.array([float(nxt()) for _ in range(9 * n)]).reshape(n, 9)
    FBlock = np.array([int(nxt()) for _ in range(n)], np.int32)
    return LocalMap(mono=mono, Ref=Ref, ScaP=ScaP, Fix=Fix, Sign=Sign, stno=stno, stVal=stVal, m=m, n=n, U=U,
                    Ui=Ui, Uj=Uj, W=W, photo=photo, feature=feature, V=V, FBlock=FBlock)


def write_set(dirpath, maps):
    os.makedirs(dirpath, exist_ok=True)
    for k, lm in enumerate(maps):
        write_localmap(os.path.join(dirpath, f"localmap_{k + 1}.txt"), lm)


# name -> (type, N maps, new features per frame, vis, path)   (BASELINE.md section 2).  path: keyword arguments of _world --
# the stand-ins revisit like the real sets do (README.txt:58-60), which is what makes their global coordinates well
# determined.  Stereo: "flower" -- laps of 120 frames in ever new directions that all start and end at the same place, the
# frames within 10 frames of a lap boundary re-observe 40 % of the points their predecessors of the previous lap saw.
# Mono (scale is observable only through shared points, so an open monocular chain drifts far more): laps of 40 frames,
# every lap slightly turned against the previous one, half of ALL points re-observed one lap later.  Lap boundaries do not
# coincide with the power-of-two blocks of the join tree.
FLOWER = dict(lap=120, home=10, revisit=0.4)
SPIRAL = dict(lap=40, home=20, revisit=0.5, turn=0.15)
# AP_Vaihingen-like (BASELINE.json configs[4]; the real set is "Aerial Photogrammetric Vaihingen Monocular",
# /root/reference/DataForC/AP_Vaihingen_C/...Dataset.txt:1, a link only): a monocular block of parallel strips with side overlap
# -- 12 strips of 20 frames.  (The block is kept at a size whose global coordinates are well determined: with 30 % side overlap
# the strips hang on each other through a third of their points only, and at 24 strips, or 40 frames a strip, two fp64
# evaluations of the reference path already differ by 1e-5 on the far corner -- measured with the oracle and its
# long-double twin; at this size by 5e-8.)
AERIAL = dict(strip=20, spacing=3.5)
CONFIGS = {
    "rs90":     ("Monocular", 88, 300, 4, SPIRAL),
    "rs468":    ("Monocular", 466, 300, 4, SPIRAL),
    "nc3500":   ("Stereo", 3499, 130, 5, FLOWER),
    "aerial":   ("Monocular", 238, 150, 4, AERIAL),
    "synth16k": ("Monocular", 16384, 64, 4, SPIRAL),
    "synth64k": ("Stereo", 65536, 64, 5, FLOWER),
}


def make_config(name, n_maps=None, seed=0, new_per_frame=None, vis=None, path=None, only=None):
    """The named stand-in set (optionally shortened / thinned, or just the slice `only` = (lo, hi) of it): returns
    (type, list of LocalMap)."""
    typ, N, npf, cvis, cpath = CONFIGS[name]
    gen = make_stereo_set if typ == "Stereo" else make_mono_set
    return typ, gen(n_maps or N, new_per_frame or npf, vis or cvis, seed, only=only, **(cpath if path is None else path))


def schur_like_matrix(m, band=12, hubs=12, seed=0):
    """Upper-block CSR pattern + values of a Schur-like 6x6-block symmetric matrix for the stand-alone SpMV measurements
    (bench.py --config spmv-stream, tools/spmv_bench.py): a pose chain with `band` neighbours plus `hubs` dense hub columns,
    like S at the top of a join tree.  Returns (rowptr int32 [m+1], colidx int32 [nnzb] sorted within a row, val [nnzb, 36])."""
    rng = np.random.default_rng(seed)
    hub = np.sort(rng.choice(m, size=min(hubs, m), replace=False))
    p = np.arange(m)
    nb = np.minimum(m, p + band + 1) - p                     # chain part of row p: columns p .. p+band
    cnt_h = np.zeros(m, np.int64)
    for h in hub:                                            # hub h lies beyond the band of the rows p < h - band
        cnt_h[:max(0, h - band)] += 1
    rowptr = np.concatenate([[0], np.cumsum(nb + cnt_h)]).astype(np.int64)
    assert rowptr[-1] < 2**31
    colidx = np.empty(rowptr[-1], np.int32)
    base = rowptr[:-1]
    for d in range(band + 1):
        rows = p[p + d < m]
        colidx[base[rows] + d] = rows + d
    fill = nb.copy()
    for h in hub:                                            # ascending hubs keep a row's columns sorted
        rows = p[:max(0, h - band)]
        colidx[base[rows] + fill[rows]] = h
        fill[rows] += 1
    val = rng.normal(size=(len(colidx), 36))
    return rowptr.astype(np.int32), colidx, val
