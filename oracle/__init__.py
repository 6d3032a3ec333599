"""CPU oracle bindings (ctypes over oracle/_build/liboracle.so).

TEST INFRASTRUCTURE, NOT PRODUCT.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this package; nothing under
``stereo_orb_slam_amd/`` does.  PARITY UNPINNED - see ``oracle/ba_oracle.h``.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

u32p = np.ctypeslib.ndpointer(np.uint32, flags="C_CONTIGUOUS")
u8p = np.ctypeslib.ndpointer(np.uint8, flags="C_CONTIGUOUS")
f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")


class BaOptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("check_termination", C.c_int32),
        ("huber_delta", C.c_double), ("lower_bound", C.c_double), ("upper_bound", C.c_double),
        ("initial_radius", C.c_double), ("max_radius", C.c_double), ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double), ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double),
        ("parameter_tolerance", C.c_double), ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("jacobi_scaling", C.c_int32), ("num_threads", C.c_int32),
    ]


class BaIteration(C.Structure):
    _fields_ = [
        ("cost", C.c_double), ("candidate_cost", C.c_double), ("model_cost_change", C.c_double),
        ("relative_decrease", C.c_double), ("radius", C.c_double), ("step_norm", C.c_double),
        ("gradient_max_norm", C.c_double), ("accepted", C.c_int32), ("valid", C.c_int32),
    ]


class BaSummary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32),
        ("accepted", C.c_int32), ("termination", C.c_int32), ("line_search_steps", C.c_int32),
        ("solve_seconds", C.c_double), ("setup_seconds", C.c_double),
    ]


class PgOptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("max_trials", C.c_int32),
        ("huber_delta", C.c_double), ("initial_lambda_scale", C.c_double),
        ("num_threads", C.c_int32), ("reserved", C.c_int32),
    ]


class PgIteration(C.Structure):
    _fields_ = [("chi2", C.c_double), ("lam", C.c_double), ("trials", C.c_int32), ("accepted", C.c_int32)]


class PgSummary(C.Structure):
    _fields_ = [
        ("initial_chi2", C.c_double), ("final_chi2", C.c_double), ("iterations", C.c_int32),
        ("termination", C.c_int32), ("solve_seconds", C.c_double), ("setup_seconds", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile the oracle with gcc (building the checker is not using it)."""
    if force or not os.path.exists(_LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []))
    return _LIB_PATH


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        build()
    L = C.CDLL(_LIB_PATH)
    L.oracle_ba_options_default.argtypes = [C.POINTER(BaOptions)]
    L.oracle_ba_residual.argtypes = [f64p] * 6
    L.oracle_ba_residual_jacobian.argtypes = [f64p] * 8
    L.oracle_huber.argtypes = [C.c_double, C.c_double, f64p]
    common = [C.c_uint32, u32p, u32p, f32p, f64p, f64p, f64p, f64p]
    L.oracle_ba_linearize.argtypes = common + [C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_ba_linearize.restype = C.c_double
    L.oracle_ba_cost.argtypes = common + [C.c_double]
    L.oracle_ba_cost.restype = C.c_double
    prob = [C.c_uint32, C.c_uint32, C.c_uint32, u32p, u32p, f32p, f64p, f64p, f64p, f64p, C.c_void_p, C.POINTER(BaOptions)]
    L.oracle_ba_step.argtypes = prob + [C.c_double] + [C.c_void_p] * 5
    L.oracle_ba_step.restype = C.c_int
    L.oracle_ba_step_sharded.argtypes = [C.c_uint32] + prob + [C.c_double, C.c_void_p, C.c_void_p]
    L.oracle_ba_step_sharded.restype = C.c_int
    L.oracle_ba_shard_system.argtypes = [C.c_uint32, C.c_uint32] + prob + [C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    L.oracle_ba_shard_system.restype = C.c_int
    L.oracle_ba_solve.argtypes = prob + [C.POINTER(BaSummary), C.c_void_p]
    L.oracle_ba_solve.restype = C.c_int
    if hasattr(L, "oracle_pg_solve"):
        L.oracle_pg_options_default.argtypes = [C.POINTER(PgOptions)]
        L.oracle_pg_edge.argtypes = [f64p, f64p, f64p, f64p, f64p, f64p]
        pg = [C.c_uint32, C.c_uint32, f64p, u8p, u32p, u32p, f64p, f64p, C.POINTER(PgOptions)]
        L.oracle_pg_solve.argtypes = pg + [C.POINTER(PgSummary), C.c_void_p]
        L.oracle_pg_solve.restype = C.c_int
        L.oracle_pg_chi2.argtypes = [C.c_uint32, C.c_uint32, f64p, u32p, u32p, f64p, f64p, C.c_double, C.c_void_p]
        L.oracle_pg_chi2.restype = C.c_double
        L.oracle_pg_linearize.argtypes = pg[:-1] + [C.c_double, f64p, f64p]
        L.oracle_pg_linearize.restype = C.c_double
    _lib = L
    return L


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def default_options(**kw) -> BaOptions:
    o = BaOptions()
    lib().oracle_ba_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


def residual(cam, pt, uv, pl, pr):
    r = np.zeros(4)
    lib().oracle_ba_residual(*(np.ascontiguousarray(a, np.float64) for a in (cam, pt, uv, pl, pr)), r)
    return r


def residual_jacobian(cam, pt, uv, pl, pr):
    r, jc, jp = np.zeros(4), np.zeros(24), np.zeros(12)
    lib().oracle_ba_residual_jacobian(*(np.ascontiguousarray(a, np.float64) for a in (cam, pt, uv, pl, pr)), r, jc, jp)
    return r, jc.reshape(4, 6), jp.reshape(4, 3)


def huber(s, delta=1.0):
    rho = np.zeros(3)
    lib().oracle_huber(float(s), float(delta), rho)
    return rho


def _prep(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr):
    return (np.ascontiguousarray(obs_cam, np.uint32), np.ascontiguousarray(obs_pt, np.uint32),
            np.ascontiguousarray(obs_uv, np.float32).reshape(-1, 4),
            np.ascontiguousarray(cams, np.float64).reshape(-1, 6), np.ascontiguousarray(pts, np.float64).reshape(-1, 3),
            np.ascontiguousarray(pl, np.float64).reshape(12), np.ascontiguousarray(pr, np.float64).reshape(12))


def linearize(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, fixed=None, delta=1.0):
    oc, op, uv, cams, pts, pl, pr = _prep(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr)
    n = len(oc)
    r, jc, jp = np.zeros((n, 4)), np.zeros((n, 24)), np.zeros((n, 12))
    fx = None if fixed is None else np.ascontiguousarray(fixed, np.uint8)
    cost = lib().oracle_ba_linearize(n, oc, op, uv, cams, pts, pl, pr, _ptr(fx), delta, _ptr(r), _ptr(jc), _ptr(jp))
    return cost, r, jc.reshape(n, 4, 6), jp.reshape(n, 4, 3)


def cost(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, delta=1.0):
    oc, op, uv, cams, pts, pl, pr = _prep(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr)
    return lib().oracle_ba_cost(len(oc), oc, op, uv, cams, pts, pl, pr, delta)


def step(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, fixed, radius, opts=None, n_rank=0):
    """One LM step; returns dict(S, rhs, dc, dp, cost, model_cost_change, candidate_cost, step_norm)."""
    oc, op, uv, cams, pts, pl, pr = _prep(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr)
    opts = opts or default_options()
    fx = np.ascontiguousarray(fixed, np.uint8)
    nc, npt = len(cams), len(pts)
    nf = int(nc - fx.sum())
    S, rhs = np.zeros((6 * nf, 6 * nf)), np.zeros(6 * nf)
    if n_rank:
        rc = lib().oracle_ba_step_sharded(n_rank, nc, npt, len(oc), oc, op, uv, cams, pts, pl, pr, _ptr(fx),
                                          C.byref(opts), radius, _ptr(S), _ptr(rhs))
        if rc:
            raise RuntimeError(f"oracle_ba_step_sharded failed: {rc}")
        return dict(S=S, rhs=rhs)
    dc, dp, sc = np.zeros((nc, 6)), np.zeros((npt, 3)), np.zeros(4)
    rc = lib().oracle_ba_step(nc, npt, len(oc), oc, op, uv, cams, pts, pl, pr, _ptr(fx), C.byref(opts), radius,
                              _ptr(S), _ptr(rhs), _ptr(dc), _ptr(dp), _ptr(sc))
    if rc:
        raise RuntimeError(f"oracle_ba_step failed: {rc}")
    return dict(S=S, rhs=rhs, dc=dc, dp=dp, cost=sc[0], model_cost_change=sc[1], candidate_cost=sc[2], step_norm=sc[3])


def shard_system(rank, n_rank, obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, fixed, radius, opts=None):
    """One rank's pre-reduction payload of a point-sharded job: dict(S, rhs, diag)."""
    oc, op, uv, cams, pts, pl, pr = _prep(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr)
    opts = opts or default_options()
    fx = np.ascontiguousarray(fixed, np.uint8)
    nf = int(len(cams) - fx.sum())
    S, rhs, diag = np.zeros((6 * nf, 6 * nf)), np.zeros(6 * nf), np.zeros(6 * nf)
    rc = lib().oracle_ba_shard_system(rank, n_rank, len(cams), len(pts), len(oc), oc, op, uv, cams, pts, pl, pr, _ptr(fx),
                                      C.byref(opts), radius, _ptr(S), _ptr(rhs), _ptr(diag))
    if rc:
        raise RuntimeError(f"oracle_ba_shard_system failed: {rc}")
    return dict(S=S, rhs=rhs, diag=diag)


def solve(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr, fixed, opts=None):
    """Full LM solve; returns (cams, pts, summary, iteration_log)."""
    oc, op, uv, cams, pts, pl, pr = _prep(obs_cam, obs_pt, obs_uv, cams, pts, pl, pr)
    cams, pts = cams.copy(), pts.copy()
    opts = opts or default_options()
    fx = np.ascontiguousarray(fixed, np.uint8)
    summ = BaSummary()
    log = (BaIteration * (opts.max_iterations + 1))()
    rc = lib().oracle_ba_solve(len(cams), len(pts), len(oc), oc, op, uv, cams, pts, pl, pr, _ptr(fx), C.byref(opts),
                               C.byref(summ), C.cast(log, C.c_void_p))
    if rc:
        raise RuntimeError(f"oracle_ba_solve failed: {rc}")
    return cams, pts, summ, list(log)[: summ.iterations + 1]
