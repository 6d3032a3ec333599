"""Thin Python handle over the pose-graph C ABI (include/soslam_pg.h) for tests and benchmarks."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


class PgOptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("max_trials", C.c_int32), ("huber_delta", C.c_double), ("tau", C.c_double),
        ("pcg_tolerance", C.c_double), ("pcg_max_iterations", C.c_int32), ("verbose", C.c_int32), ("device", C.c_int32),
        ("preconditioner", C.c_int32), ("stream", C.c_void_p),
    ]


class PgIteration(C.Structure):
    _fields_ = [("chi2", C.c_double), ("lam", C.c_double), ("trials", C.c_int32), ("accepted", C.c_int32),
                ("linear_iterations", C.c_int32), ("reserved", C.c_int32)]


class PgSummary(C.Structure):
    _fields_ = [
        ("initial_chi2", C.c_double), ("final_chi2", C.c_double), ("iterations", C.c_int32), ("termination", C.c_int32),
        ("linear_iterations", C.c_int32), ("reserved", C.c_int32), ("solve_seconds", C.c_double), ("setup_seconds", C.c_double),
        ("linearize_ms", C.c_double), ("linear_solve_ms", C.c_double),
    ]


PG_SYMBOLS = ["soslam_pg_options_default", "soslam_pg_create", "soslam_pg_destroy", "soslam_pg_set_graph", "soslam_pg_append",
              "soslam_pg_graph_size", "soslam_pg_optimize",
              "soslam_pg_get_estimates", "soslam_pg_get_iteration_log", "soslam_pg_solve", "soslam_pg_debug_linearize",
              "soslam_pg_time_linearize"]
_bound = False


def _L():
    global _bound
    L = _lib.lib()
    if not _bound:
        vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
        L.soslam_pg_options_default.argtypes = [C.POINTER(PgOptions)]
        L.soslam_pg_options_default.restype = None
        L.soslam_pg_create.argtypes = [C.POINTER(PgOptions), C.POINTER(vp)]
        L.soslam_pg_destroy.argtypes = [vp]
        L.soslam_pg_destroy.restype = None
        L.soslam_pg_set_graph.argtypes = [vp, u32, vp, vp, u32, vp, vp, vp, vp]
        L.soslam_pg_append.argtypes = [vp, u32, vp, vp, u32, vp, vp, vp, vp]
        L.soslam_pg_graph_size.argtypes = [vp, C.POINTER(u32), C.POINTER(u32)]
        L.soslam_pg_optimize.argtypes = [vp, C.POINTER(PgSummary)]
        L.soslam_pg_get_estimates.argtypes = [vp, vp]
        L.soslam_pg_get_iteration_log.argtypes = [vp, vp, i32, C.POINTER(i32)]
        L.soslam_pg_solve.argtypes = [C.POINTER(PgOptions), u32, vp, vp, u32, vp, vp, vp, vp, C.POINTER(PgSummary)]
        L.soslam_pg_debug_linearize.argtypes = [vp, vp, vp, vp, C.POINTER(C.c_double), vp, vp]
        L.soslam_pg_time_linearize.argtypes = [vp, i32, C.POINTER(C.c_float)]
        _bound = True
    return L


def default_options(**kw) -> PgOptions:
    o = PgOptions()
    _L().soslam_pg_options_default(C.byref(o))
    for k, v in kw.items():
        setattr(o, k, v)
    return o


class PoseGraph:
    def __init__(self, options: PgOptions | None = None):
        self._L = _L()
        self._h = C.c_void_p()
        self.options = options or default_options()
        _lib.check(self._L.soslam_pg_create(C.byref(self.options), C.byref(self._h)), "soslam_pg_create")
        self.n_vertex = self.n_edge = self.n_free = 0

    def close(self):
        if self._h:
            self._L.soslam_pg_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_graph(self, est, fixed, e_from, e_to, meas, info):
        est = np.ascontiguousarray(est, np.float64).reshape(-1, 7)
        fx = np.ascontiguousarray(fixed, np.uint8)
        ef, et = np.ascontiguousarray(e_from, np.uint32), np.ascontiguousarray(e_to, np.uint32)
        ms = np.ascontiguousarray(meas, np.float64).reshape(-1, 7)
        inf = np.ascontiguousarray(info, np.float64).reshape(36)
        _lib.check(self._L.soslam_pg_set_graph(self._h, len(est), _lib.ptr(est), _lib.ptr(fx), len(ef), _lib.ptr(ef), _lib.ptr(et),
                                               _lib.ptr(ms), _lib.ptr(inf)), "soslam_pg_set_graph")
        self.n_vertex, self.n_edge, self.n_free = len(est), len(ef), len(est) - int(fx.astype(bool).sum())

    def append(self, est_add, fixed_add, e_from, e_to, meas, info=None):
        """Grow the handle's graph (soslam_pg_append): new vertices and edges; existing vertices keep the estimates the last
        optimize() left on the device."""
        est = np.ascontiguousarray(est_add, np.float64).reshape(-1, 7)
        fx = np.ascontiguousarray(fixed_add, np.uint8)
        ef, et = np.ascontiguousarray(e_from, np.uint32), np.ascontiguousarray(e_to, np.uint32)
        ms = np.ascontiguousarray(meas, np.float64).reshape(-1, 7)
        inf = None if info is None else np.ascontiguousarray(info, np.float64).reshape(36)
        _lib.check(self._L.soslam_pg_append(self._h, len(est), _lib.ptr(est) if len(est) else None, _lib.ptr(fx) if len(fx) else None,
                                            len(ef), _lib.ptr(ef) if len(ef) else None, _lib.ptr(et) if len(et) else None,
                                            _lib.ptr(ms) if len(ms) else None, _lib.ptr(inf)), "soslam_pg_append")
        nv, ne = C.c_uint32(), C.c_uint32()
        _lib.check(self._L.soslam_pg_graph_size(self._h, C.byref(nv), C.byref(ne)), "soslam_pg_graph_size")
        self.n_free = (self.n_free if self.n_vertex else 0) + len(est) - int(fx.astype(bool).sum())
        self.n_vertex, self.n_edge = nv.value, ne.value

    def load(self, g):
        self.set_graph(g.est, g.fixed, g.e_from, g.e_to, g.meas, g.info)
        return self

    def optimize(self) -> PgSummary:
        s = PgSummary()
        _lib.check(self._L.soslam_pg_optimize(self._h, C.byref(s)), "soslam_pg_optimize")
        return s

    def estimates(self):
        est = np.zeros((self.n_vertex, 7))
        _lib.check(self._L.soslam_pg_get_estimates(self._h, _lib.ptr(est)), "soslam_pg_get_estimates")
        return est

    def iteration_log(self):
        n = C.c_int32()
        _lib.check(self._L.soslam_pg_get_iteration_log(self._h, None, 0, C.byref(n)), "soslam_pg_get_iteration_log")
        buf = (PgIteration * max(1, n.value))()
        _lib.check(self._L.soslam_pg_get_iteration_log(self._h, C.cast(buf, C.c_void_p), n.value, C.byref(n)), "soslam_pg_get_iteration_log")
        return list(buf)[: n.value]

    def debug_linearize(self, dense=True):
        e, ji, jj = np.zeros((self.n_edge, 6)), np.zeros((self.n_edge, 6, 6)), np.zeros((self.n_edge, 6, 6))
        chi = C.c_double()
        n6 = 6 * self.n_free
        H = np.zeros((n6, n6)) if dense else None
        b = np.zeros(n6)
        _lib.check(self._L.soslam_pg_debug_linearize(self._h, _lib.ptr(e), _lib.ptr(ji), _lib.ptr(jj), C.byref(chi), _lib.ptr(H),
                                                     _lib.ptr(b)), "soslam_pg_debug_linearize")
        return dict(e=e, ji=ji, jj=jj, chi2=chi.value, H=H, b=b)

    def time_linearize(self, reps=20) -> float:
        ms = C.c_float()
        _lib.check(self._L.soslam_pg_time_linearize(self._h, reps, C.byref(ms)), "soslam_pg_time_linearize")
        return ms.value
