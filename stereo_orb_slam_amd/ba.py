"""Thin Python handle over the C ABI of include/soslam_ba.h (tests, bench.py, smoke).

The C++ host shim (stereo_orb_slam_amd/host/bundle_adjuster.h) is the drop-in for the reference's
``BundleAdjuster``; this module only lets Python drive the same entry points.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import BaIteration, BaOptions, BaSummary


def default_options(**kw) -> BaOptions:
    o = BaOptions()
    _lib.lib().soslam_ba_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def summary_dict(s: BaSummary) -> dict:
    d = {k: getattr(s, k) for k, _ in BaSummary._fields_ if k not in ("stage_ms", "stage_calls")}
    d["termination_name"] = _lib.TERM_NAMES[s.termination] if 0 <= s.termination < len(_lib.TERM_NAMES) else "?"
    d["stage_ms"] = {n: s.stage_ms[i] for i, n in enumerate(_lib.STAGE_NAMES)}
    d["stage_calls"] = {n: s.stage_calls[i] for i, n in enumerate(_lib.STAGE_NAMES)}
    return d


class BundleAdjustment:
    """One device-resident BA problem.  Mirrors the call order of BundleAdjuster::Optimize
    (/root/reference/src/bundle_adjuster.cpp:39-133): projection, problem, state, solve, read back."""

    def __init__(self, options: BaOptions | None = None):
        self._L = _lib.lib()
        self._h = C.c_void_p()
        self.options = options or default_options()
        _lib.check(self._L.soslam_ba_create(C.byref(self.options), C.byref(self._h)), "soslam_ba_create")
        self._keep = []
        self.n_cam = self.n_pt = self.n_obs = self.n_free = 0

    def set_options(self, options):
        """Replace the options of the live handle (device and stream stay)."""
        _lib.check(self._L.soslam_ba_set_options(self._h, C.byref(options)), "soslam_ba_set_options")
        self.options = options

    def close(self):
        if self._h:
            self._L.soslam_ba_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_projection(self, proj_l, proj_r):
        pl = np.ascontiguousarray(proj_l, np.float64).reshape(12)
        pr = np.ascontiguousarray(proj_r, np.float64).reshape(12)
        _lib.check(self._L.soslam_ba_set_projection(self._h, _lib.ptr(pl), _lib.ptr(pr)), "soslam_ba_set_projection")

    def set_problem(self, n_cam, n_pt, obs_cam, obs_pt, obs_uv, cam_fixed=None):
        oc = np.ascontiguousarray(obs_cam, np.uint32)
        op = np.ascontiguousarray(obs_pt, np.uint32)
        uv = np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 4)
        fx = None if cam_fixed is None else np.ascontiguousarray(cam_fixed, np.uint8)
        if len(oc) != len(op) or len(uv) != len(oc) or (fx is not None and len(fx) != n_cam):
            raise ValueError("inconsistent observation arrays")
        _lib.check(self._L.soslam_ba_set_problem(self._h, n_cam, n_pt, len(oc), _lib.ptr(oc), _lib.ptr(op), _lib.ptr(uv),
                                                 _lib.ptr(fx)), "soslam_ba_set_problem")
        self.n_cam, self.n_pt, self.n_obs = n_cam, n_pt, len(oc)
        self.n_free = n_cam - (int(fx.astype(bool).sum()) if fx is not None else 0)

    def set_state(self, poses, points):
        c = np.ascontiguousarray(poses, np.float64).reshape(self.n_cam, 6)
        p = np.ascontiguousarray(points, np.float64).reshape(self.n_pt, 3)
        _lib.check(self._L.soslam_ba_set_state(self._h, _lib.ptr(c), _lib.ptr(p)), "soslam_ba_set_state")

    def load(self, prob, poses=None, points=None):
        """Upload a synth.BaProblem (projection, graph, initial state)."""
        self.set_projection(prob.proj_l, prob.proj_r)
        self.set_problem(prob.n_cam, prob.n_pt, prob.obs_cam, prob.obs_pt, prob.obs_uv, prob.cam_fixed)
        self.set_state(prob.poses_cw() if poses is None else poses, prob.points_f64() if points is None else points)
        return self

    def get_state(self):
        c, p = np.zeros((self.n_cam, 6)), np.zeros((self.n_pt, 3))
        _lib.check(self._L.soslam_ba_get_state(self._h, _lib.ptr(c), _lib.ptr(p)), "soslam_ba_get_state")
        return c, p

    def solve(self) -> BaSummary:
        s = BaSummary()
        _lib.check(self._L.soslam_ba_solve(self._h, C.byref(s)), "soslam_ba_solve")
        return s

    def iterate(self, n: int) -> BaSummary:
        s = BaSummary()
        _lib.check(self._L.soslam_ba_iterate(self._h, n, C.byref(s)), "soslam_ba_iterate")
        return s

    def iteration_log(self):
        n = C.c_int32()
        _lib.check(self._L.soslam_ba_get_iteration_log(self._h, None, 0, C.byref(n)), "soslam_ba_get_iteration_log")
        buf = (BaIteration * max(1, n.value))()
        _lib.check(self._L.soslam_ba_get_iteration_log(self._h, C.cast(buf, C.c_void_p), n.value, C.byref(n)),
                   "soslam_ba_get_iteration_log")
        return list(buf)[: n.value]

    # ---- multi-GPU plumbing ---------------------------------------------------------------------------
    def set_covisibility(self, pairs):
        """Job-wide camera pairs [n,2] so every rank builds the same block pattern (call before set_problem)."""
        pr = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        a, b = np.ascontiguousarray(pr[:, 0]), np.ascontiguousarray(pr[:, 1])
        _lib.check(self._L.soslam_ba_set_covisibility(self._h, len(a), _lib.ptr(a), _lib.ptr(b)), "soslam_ba_set_covisibility")

    def reduce_buffer_count(self) -> int:
        n = C.c_uint64()
        _lib.check(self._L.soslam_ba_reduce_buffer_count(self._h, C.byref(n)), "soslam_ba_reduce_buffer_count")
        return n.value

    def set_reduce_buffer(self, device_ptr: int, count: int):
        _lib.check(self._L.soslam_ba_set_reduce_buffer(self._h, C.c_void_p(device_ptr), count), "soslam_ba_set_reduce_buffer")

    def set_allreduce(self, fn, rank: int, world: int):
        """fn(device_ptr:int, count:int, op:int, stream:int) -> int (0 = ok)."""
        def tramp(user, buf, count, op, stream):
            try:
                return int(fn(buf or 0, count, op, stream or 0) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                print(f"[soslam] all-reduce callback raised: {e!r}", flush=True)
                return 1
        cb = _lib.ALLREDUCE_FN(tramp)
        self._keep.append(cb)
        _lib.check(self._L.soslam_ba_set_allreduce(self._h, cb, None, rank, world), "soslam_ba_set_allreduce")

    def set_host_allreduce(self, fn, rank: int, world: int):
        """fn(host_array: np.ndarray[f64], op: int) -> int (0 = ok): in-place all-reduce of a host range the library
        staged (soslam_ba_set_host_allreduce)."""
        def tramp(user, buf, count, op):
            try:
                a = np.ctypeslib.as_array(buf, shape=(count,))
                return int(fn(a, op) or 0)
            except Exception as e:  # never let an exception cross the C boundary
                print(f"[soslam] host all-reduce callback raised: {e!r}", flush=True)
                return 1
        cb = _lib.HOST_ALLREDUCE_FN(tramp)
        self._keep.append(cb)
        _lib.check(self._L.soslam_ba_set_host_allreduce(self._h, cb, None, rank, world), "soslam_ba_set_host_allreduce")

    def init_rccl(self, unique_id: bytes, rank: int, world: int):
        """The library's own RCCL leg (soslam_ba_init_rccl): every rank calls it with rank 0's unique id."""
        if len(unique_id) != _lib.RCCL_UNIQUE_ID_BYTES:
            raise ValueError("an RCCL unique id has 128 bytes")
        buf = C.create_string_buffer(bytes(unique_id), _lib.RCCL_UNIQUE_ID_BYTES)
        _lib.check(self._L.soslam_ba_init_rccl(self._h, C.cast(buf, C.c_void_p), rank, world), "soslam_ba_init_rccl")

    def agree_status(self, local_status: int) -> int:
        """MAX of the ranks' status words through the attached collective (soslam_ba_agree_status): a rank whose set-up failed
        calls this instead of skipping the solve, so that the other ranks do not wait for it in their first all-reduce."""
        out = C.c_int(local_status)
        _lib.check(self._L.soslam_ba_agree_status(self._h, int(local_status), C.byref(out)), "soslam_ba_agree_status")
        return out.value

    def get_state_global(self, n_pt_global: int, shard_begin: int):
        """Poses and the points of all ranks (soslam_ba_get_state_global)."""
        c, p = np.zeros((self.n_cam, 6)), np.zeros((n_pt_global, 3))
        _lib.check(self._L.soslam_ba_get_state_global(self._h, _lib.ptr(c), n_pt_global, shard_begin, _lib.ptr(p)),
                   "soslam_ba_get_state_global")
        return c, p

    # ---- stage-level access ---------------------------------------------------------------------------
    def time_kernel(self, kernel: int, reps: int = 20) -> float:
        ms = C.c_float()
        _lib.check(self._L.soslam_ba_time_kernel(self._h, kernel, reps, C.byref(ms)), "soslam_ba_time_kernel")
        return ms.value

    def debug_step(self, radius: float):
        _lib.check(self._L.soslam_ba_debug_step(self._h, radius), "soslam_ba_debug_step")

    def debug_read(self, what: int) -> np.ndarray:
        F = self.n_free
        shape = {
            _lib.DBG_RESIDUALS: (self.n_obs, 4), _lib.DBG_JAC_CAM: (self.n_obs, 4, 6), _lib.DBG_JAC_POINT: (self.n_obs, 4, 3),
            _lib.DBG_COST: (1,), _lib.DBG_S_DENSE: (6 * F, 6 * F), _lib.DBG_RHS: (6 * F,), _lib.DBG_STEP_CAM: (self.n_cam, 6),
            _lib.DBG_STEP_POINT: (self.n_pt, 3), _lib.DBG_STEP_SCALARS: (6,), _lib.DBG_COMPACT_ROWS: (self.n_obs, 9),
        }[what]
        out = np.zeros(shape)
        _lib.check(self._L.soslam_ba_debug_read(self._h, what, _lib.ptr(out), out.nbytes), "soslam_ba_debug_read")
        return out


def rccl_unique_id() -> bytes:
    """ncclGetUniqueId through the library (rank 0 draws it, the other ranks receive it by a side channel)."""
    buf = C.create_string_buffer(_lib.RCCL_UNIQUE_ID_BYTES)
    _lib.check(_lib.lib().soslam_rccl_get_unique_id(C.cast(buf, C.c_void_p)), "soslam_rccl_get_unique_id")
    return buf.raw


def optimize(prob, options: BaOptions | None = None):
    """soslam_ba_optimize on a synth.BaProblem: returns (poses_cw, points, summary)."""
    L = _lib.lib()
    o = options or default_options()
    c = prob.poses_cw()
    p = prob.points_f64()
    pl = np.ascontiguousarray(prob.proj_l, np.float64)
    pr = np.ascontiguousarray(prob.proj_r, np.float64)
    oc = np.ascontiguousarray(prob.obs_cam, np.uint32)
    op = np.ascontiguousarray(prob.obs_pt, np.uint32)
    uv = np.ascontiguousarray(prob.obs_uv, np.float32)
    fx = np.ascontiguousarray(prob.cam_fixed, np.uint8)
    s = BaSummary()
    _lib.check(L.soslam_ba_optimize(C.byref(o), _lib.ptr(pl), _lib.ptr(pr), prob.n_cam, _lib.ptr(c), prob.n_pt, _lib.ptr(p),
                                    prob.n_obs, _lib.ptr(oc), _lib.ptr(op), _lib.ptr(uv), _lib.ptr(fx), C.byref(s)),
               "soslam_ba_optimize")
    return c, p, s
