"""Synthetic BA / pose-graph workloads.

``generate_ba`` / ``generate_pg`` call the C generator compiled into libsoslam_ba.so
(stereo_orb_slam_amd/csrc/synth.c).  ``numpy_*`` are the Python mirror of the same counter-based
generator: identical 64-bit streams, floats equal to float32 rounding - tests/test_synth.py compares them.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np

from . import _lib

SEED = 20241004
GOLD = np.uint64(0x9E3779B97F4A7C15)
ST_TRACK, ST_PX, ST_PY, ST_PZ, ST_NOISE, ST_OSEL, ST_OVAL, ST_POSE, ST_DEPTH = 1, 2, 3, 4, 5, 6, 7, 8, 9
MAX_TRACK = 64


# ---- numpy mirror of the raw generator ---------------------------------------------------------------

def _mix64(z: np.ndarray) -> np.ndarray:
    z = z.astype(np.uint64)
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def numpy_u64(seed: int, stream: int, index) -> np.ndarray:
    index = np.asarray(index, dtype=np.uint64)
    with np.errstate(over="ignore"):
        k = _mix64(np.uint64(seed) + GOLD * np.uint64(stream + 1))
        return _mix64(k + GOLD * (index + np.uint64(1)))


def numpy_uniform(seed: int, stream: int, index) -> np.ndarray:
    return (numpy_u64(seed, stream, index) >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def numpy_normal(seed: int, stream: int, index) -> np.ndarray:
    index = np.asarray(index, dtype=np.uint64)
    u1 = numpy_uniform(seed, stream, np.uint64(2) * index)
    u2 = numpy_uniform(seed, stream, np.uint64(2) * index + np.uint64(1))
    return np.sqrt(-2.0 * np.log(1.0 - u1)) * np.cos(2.0 * np.pi * u2)


# ---- problems -----------------------------------------------------------------------------------------

@dataclass
class BaProblem:
    """A BA problem in the reference's container conventions (see include/soslam_synth.h)."""
    poses_wc: np.ndarray      # [n_cam,4,4] float32 camera->world
    points: np.ndarray        # [n_pt,3] float32
    obs_cam: np.ndarray       # [n_obs] uint32, frame-major
    obs_pt: np.ndarray        # [n_obs] uint32
    obs_uv: np.ndarray        # [n_obs,4] float32
    proj_l: np.ndarray        # [12] float64
    proj_r: np.ndarray        # [12] float64
    true_poses_wc: np.ndarray | None = None
    true_points: np.ndarray | None = None
    name: str = ""
    cam_fixed: np.ndarray = field(default=None)  # [n_cam] uint8; default: first camera fixed

    def __post_init__(self):
        if self.cam_fixed is None:
            f = np.zeros(len(self.poses_wc), np.uint8)
            f[0] = 1  # /root/reference/src/bundle_adjuster.cpp:113
            self.cam_fixed = f

    @property
    def n_cam(self): return len(self.poses_wc)
    @property
    def n_pt(self): return len(self.points)
    @property
    def n_obs(self): return len(self.obs_cam)

    def poses_cw(self) -> np.ndarray:
        """world->camera [angle-axis | t] doubles, through the reference's float32 conversion."""
        L = _lib.lib()
        out = np.zeros((self.n_cam, 6))
        T = np.ascontiguousarray(self.poses_wc, np.float32)
        for i in range(self.n_cam):
            L.soslam_pose_from_global_matrix(T[i].ctypes.data_as(C.c_void_p), out[i].ctypes.data_as(C.c_void_p))
        return out

    def points_f64(self) -> np.ndarray:
        return np.ascontiguousarray(self.points, np.float64)

    def covisibility_pairs(self) -> np.ndarray:
        """Unique camera pairs (a < b) that observe a common point: the job-wide block pattern of the reduced
        camera system, which every rank of a sharded job must share (soslam_ba_set_covisibility)."""
        order = np.lexsort((self.obs_cam, self.obs_pt))
        pt, cam = self.obs_pt[order].astype(np.int64), self.obs_cam[order].astype(np.int64)
        keys = []
        d = 1
        while True:
            same = pt[d:] == pt[:-d]
            if not same.any():
                break
            a, b = cam[:-d][same], cam[d:][same]
            keys.append(np.unique(np.minimum(a, b) * self.n_cam + np.maximum(a, b)))
            d += 1
        if not keys:
            return np.zeros((0, 2), np.uint32)
        k = np.unique(np.concatenate(keys))
        return np.stack([k // self.n_cam, k % self.n_cam], -1).astype(np.uint32)

    def shard(self, rank: int, world: int) -> "BaProblem":
        """Rank's share of a job sharded by point: all cameras, a contiguous
        range of points renumbered from zero, and the observations of those points."""
        b = (rank * self.n_pt) // world
        e = ((rank + 1) * self.n_pt) // world
        keep = (self.obs_pt >= b) & (self.obs_pt < e)
        return BaProblem(self.poses_wc, self.points[b:e], np.ascontiguousarray(self.obs_cam[keep]),
                         np.ascontiguousarray(self.obs_pt[keep] - np.uint32(b)), np.ascontiguousarray(self.obs_uv[keep]),
                         self.proj_l, self.proj_r, name=f"{self.name}[{rank}/{world}]", cam_fixed=self.cam_fixed)


def ba_params(config: int | None = None, **kw) -> _lib.SynthBaParams:
    p = _lib.SynthBaParams()
    _lib.check(_lib.lib().soslam_synth_ba_config(config if config is not None else 1, C.byref(p)), "soslam_synth_ba_config")
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def generate_ba(config: int | None = None, with_truth: bool = False, **kw) -> BaProblem:
    """BASELINE.json configs[0..2] (config=1,2,3) or custom parameters via keywords."""
    L = _lib.lib()
    p = ba_params(config, **kw)
    n_obs = C.c_uint32()
    _lib.check(L.soslam_synth_ba_count(C.byref(p), C.byref(n_obs)), "soslam_synth_ba_count")
    n = n_obs.value
    poses = np.zeros((p.n_cam, 4, 4), np.float32)
    pts = np.zeros((p.n_pt, 3), np.float32)
    oc, op = np.zeros(n, np.uint32), np.zeros(n, np.uint32)
    uv = np.zeros((n, 4), np.float32)
    pl, pr = np.zeros(12), np.zeros(12)
    tp = np.zeros((p.n_cam, 4, 4)) if with_truth else None
    tx = np.zeros((p.n_pt, 3)) if with_truth else None
    _lib.check(L.soslam_synth_ba_generate(C.byref(p), _lib.ptr(poses), _lib.ptr(pts), _lib.ptr(oc), _lib.ptr(op),
                                          _lib.ptr(uv), _lib.ptr(pl), _lib.ptr(pr), _lib.ptr(tp), _lib.ptr(tx)),
               "soslam_synth_ba_generate")
    return BaProblem(poses, pts, oc, op, uv, pl, pr, tp, tx, name=f"synth-ba-{p.n_cam}c-{p.n_pt}p-{n}o")


def _rot(axis: str, a: np.ndarray) -> np.ndarray:
    c, s, o, z = np.cos(a), np.sin(a), np.ones_like(a), np.zeros_like(a)
    if axis == "x":
        m = [[o, z, z], [z, c, -s], [z, s, c]]
    elif axis == "y":
        m = [[c, z, s], [z, o, z], [-s, z, c]]
    else:
        m = [[c, -s, z], [s, c, z], [z, z, o]]
    return np.stack([np.stack(r, -1) for r in m], -2)


def numpy_ba_observations(p: _lib.SynthBaParams):
    """Vectorised mirror of the observation part of soslam_synth_ba_generate.

    Returns (obs_cam, obs_pt, obs_uv float32) in the C generator's frame-major order."""
    seed, nc, npt = int(p.seed), int(p.n_cam), int(p.n_pt)
    j = np.arange(npt, dtype=np.uint64)
    if p.track_mode == 0:
        c0 = (j * np.uint64(nc - p.track_len + 1)) // np.uint64(npt)
        length = np.full(npt, p.track_len, np.int64)
    else:
        c0 = (j * np.uint64(nc)) // np.uint64(npt)
        q = 1.0 / p.track_len
        length = 1 + np.floor(np.log(1.0 - numpy_uniform(seed, ST_TRACK, j)) / np.log(1.0 - q)).astype(np.int64)
    c0 = c0.astype(np.int64)
    length = np.minimum(np.minimum(length, nc - c0), MAX_TRACK)
    # truth path (sequential sums, same order as the C loop)
    k = np.arange(nc)
    s = p.spacing * k
    yaw = p.curvature * s
    R = _rot("y", yaw) @ _rot("x", 0.02 * np.sin(0.1 * s)) @ _rot("z", 0.015 * np.cos(0.07 * s))
    x = np.concatenate([[0.0], np.cumsum(p.spacing * np.sin(yaw))[:-1]])
    z = np.concatenate([[0.0], np.cumsum(p.spacing * np.cos(yaw))[:-1]])
    pos = np.stack([x, 0.05 * np.sin(0.05 * s), z], -1)
    zlo = np.maximum(6.0, p.spacing * (length - 1) + 4.0)
    loc = np.stack([-15.0 + 30.0 * numpy_uniform(seed, ST_PX, j), -2.0 + 5.0 * numpy_uniform(seed, ST_PY, j),
                    zlo + (60.0 - zlo) * numpy_uniform(seed, ST_PZ, j)], -1)
    X = np.einsum("nij,nj->ni", R[c0], loc) + pos[c0]
    # expand (point, k) pairs
    pj = np.repeat(np.arange(npt), length)
    kk = np.arange(length.sum()) - np.repeat(np.cumsum(length) - length, length)
    cam = c0[pj] + kk
    d = X[pj] - pos[cam]
    pc = np.einsum("nji,nj->ni", R[cam], d)
    fx, cx, cy, tx = (float(np.float32(v)) for v in (718.856, 607.1928, 185.2157, -386.1448))
    uv = np.stack([(fx * pc[:, 0] + cx * pc[:, 2]) / pc[:, 2], (fx * pc[:, 1] + cy * pc[:, 2]) / pc[:, 2],
                   (fx * pc[:, 0] + cx * pc[:, 2] + tx) / pc[:, 2], (fx * pc[:, 1] + cy * pc[:, 2]) / pc[:, 2]], -1)
    idx = (pj.astype(np.uint64) * np.uint64(MAX_TRACK) + kk.astype(np.uint64))
    comp = idx[:, None] * np.uint64(4) + np.arange(4, dtype=np.uint64)[None, :]
    uv = uv + p.pixel_sigma * numpy_normal(seed, ST_NOISE, comp)
    outlier = numpy_uniform(seed, ST_OSEL, idx) < p.outlier_frac
    uv = uv + outlier[:, None] * p.outlier_px * (2.0 * numpy_uniform(seed, ST_OVAL, comp) - 1.0)
    order = np.lexsort((pj, cam))  # frame-major, points ascending inside a frame
    return cam[order].astype(np.uint32), pj[order].astype(np.uint32), uv[order].astype(np.float32)


@dataclass
class PgProblem:
    est: np.ndarray       # [n,7] tx ty tz qx qy qz qw
    e_from: np.ndarray    # [m] uint32
    e_to: np.ndarray      # [m] uint32
    meas: np.ndarray      # [m,7]
    fixed: np.ndarray     # [n] uint8
    info: np.ndarray      # [36] row-major 6x6, diag(.01,.01,.01,1,1,1) (/root/reference/src/pose_graph_optimizer.cpp:23-26)
    true_est: np.ndarray | None = None
    name: str = ""


def generate_pg(config: int = 5, with_truth: bool = False, **kw) -> PgProblem:
    L = _lib.lib()
    p = _lib.SynthPgParams()
    _lib.check(L.soslam_synth_pg_config(config, C.byref(p)), "soslam_synth_pg_config")
    for k, v in kw.items():
        setattr(p, k, v)
    ne = C.c_uint32()
    _lib.check(L.soslam_synth_pg_count(C.byref(p), C.byref(ne)), "soslam_synth_pg_count")
    m = ne.value
    est = np.zeros((p.n_node, 7))
    ef, et = np.zeros(m, np.uint32), np.zeros(m, np.uint32)
    meas = np.zeros((m, 7))
    te = np.zeros((p.n_node, 7)) if with_truth else None
    _lib.check(L.soslam_synth_pg_generate(C.byref(p), _lib.ptr(est), _lib.ptr(ef), _lib.ptr(et), _lib.ptr(meas), _lib.ptr(te)),
               "soslam_synth_pg_generate")
    fixed = np.zeros(p.n_node, np.uint8)
    fixed[0] = 1  # /root/reference/src/pose_graph_optimizer.cpp:118-121
    info = np.diag([0.01, 0.01, 0.01, 1.0, 1.0, 1.0]).reshape(36)
    return PgProblem(est, ef, et, meas, fixed, info, te, name=f"synth-pg-{p.n_node}n-{m}e")
