"""ctypes binding of libsoslam_ba.so (the C ABI declared in include/soslam_ba.h and soslam_synth.h).

Python is plumbing here: it loads the in-tree shared library built by ``__graft_entry__.build()`` and
passes plain pointers.  There is no Python or CPU fallback for any solver entry point - if the library
is missing or no gfx950 device is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SOSLAM_LIB") or os.path.join(_HERE, "libsoslam_ba.so")   # SOSLAM_LIB: development builds
CSRC = os.path.join(_HERE, "csrc")

NUM_STAGES = 8
STAGE_NAMES = ["linearize", "point_reduce", "schur", "allreduce", "solve", "backsub", "cost", "sync"]

(OK, ERR_INVALID_ARGUMENT, ERR_HIP, ERR_NO_DEVICE, ERR_NON_FINITE, ERR_LINEAR_SOLVER, ERR_COMM, ERR_STATE) = range(8)
SOLVER_AUTO, SOLVER_DENSE_CHOLESKY, SOLVER_PCG, SOLVER_BAND_CHOLESKY = 0, 1, 2, 3
TERM_NAMES = ["max_iterations", "parameter_tolerance", "function_tolerance", "gradient_tolerance", "min_radius",
              "invalid_steps", "time"]
KERNEL_LINEARIZE, KERNEL_COST, KERNEL_POINT_REDUCE, KERNEL_SCHUR, KERNEL_BACKSUB = range(5)
(DBG_RESIDUALS, DBG_JAC_CAM, DBG_JAC_POINT, DBG_COST, DBG_S_DENSE, DBG_RHS, DBG_STEP_CAM, DBG_STEP_POINT,
 DBG_STEP_SCALARS, DBG_COMPACT_ROWS) = range(10)
REDUCE_SUM, REDUCE_MAX = 0, 1
RCCL_UNIQUE_ID_BYTES = 128


class BaOptions(C.Structure):
    _fields_ = [
        ("max_iterations", C.c_int32), ("check_termination", C.c_int32), ("linear_solver", C.c_int32),
        ("pcg_max_iterations", C.c_int32), ("pcg_tolerance", C.c_double), ("huber_delta", C.c_double),
        ("lower_bound", C.c_double), ("upper_bound", C.c_double), ("initial_radius", C.c_double),
        ("max_radius", C.c_double), ("min_radius", C.c_double), ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double), ("max_lm_diagonal", C.c_double), ("parameter_tolerance", C.c_double),
        ("function_tolerance", C.c_double), ("gradient_tolerance", C.c_double),
        ("max_solver_time_seconds", C.c_double), ("jacobi_scaling", C.c_int32), ("verbose", C.c_int32),
        ("device", C.c_int32), ("profile_stages", C.c_int32), ("stream", C.c_void_p),
    ]


class BaIteration(C.Structure):
    _fields_ = [
        ("cost", C.c_double), ("candidate_cost", C.c_double), ("model_cost_change", C.c_double),
        ("relative_decrease", C.c_double), ("radius", C.c_double), ("step_norm", C.c_double),
        ("gradient_max_norm", C.c_double), ("accepted", C.c_int32), ("valid", C.c_int32),
        ("linear_iterations", C.c_int32), ("reserved", C.c_int32),
    ]


class BaSummary(C.Structure):
    _fields_ = [
        ("initial_cost", C.c_double), ("final_cost", C.c_double), ("iterations", C.c_int32), ("accepted", C.c_int32),
        ("termination", C.c_int32), ("line_search_steps", C.c_int32), ("linear_solver", C.c_int32),
        ("linear_iterations", C.c_int32), ("solve_seconds", C.c_double), ("setup_seconds", C.c_double),
        ("stage_ms", C.c_double * NUM_STAGES), ("stage_calls", C.c_int32 * NUM_STAGES),
    ]


class SynthBaParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_cam", C.c_uint32), ("n_pt", C.c_uint32), ("track_mode", C.c_uint32),
        ("track_len", C.c_uint32), ("spacing", C.c_double), ("curvature", C.c_double), ("pixel_sigma", C.c_double),
        ("outlier_frac", C.c_double), ("outlier_px", C.c_double), ("pose_rot_sigma", C.c_double),
        ("pose_trans_sigma", C.c_double), ("depth_noise", C.c_double),
    ]


class SynthPgParams(C.Structure):
    _fields_ = [
        ("seed", C.c_uint64), ("n_node", C.c_uint32), ("n_loop_max", C.c_uint32), ("min_gap", C.c_uint32),
        ("row_len", C.c_uint32), ("step", C.c_double), ("radius", C.c_double), ("meas_trans_sigma", C.c_double),
        ("meas_rot_sigma", C.c_double), ("init_trans_sigma", C.c_double), ("init_rot_sigma", C.c_double),
    ]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int32, C.c_void_p)
HOST_ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_uint64, C.c_int32)

# every symbol include/*.h declares; tests/test_cabi.py checks the library exports each one
BA_SYMBOLS = [
    "soslam_version", "soslam_status_string", "soslam_last_error", "soslam_ba_options_default", "soslam_ba_create",
    "soslam_ba_destroy", "soslam_ba_set_options", "soslam_ba_set_projection", "soslam_ba_set_problem", "soslam_ba_set_state",
    "soslam_ba_get_state", "soslam_ba_solve", "soslam_ba_iterate", "soslam_ba_get_iteration_log",
    "soslam_ba_optimize", "soslam_ba_set_covisibility", "soslam_ba_set_allreduce", "soslam_ba_reduce_buffer_count", "soslam_ba_set_reduce_buffer",
    "soslam_ba_set_host_allreduce", "soslam_rccl_get_unique_id", "soslam_ba_init_rccl", "soslam_ba_agree_status", "soslam_ba_get_state_global",
    "soslam_ba_shard_range", "soslam_ba_time_kernel", "soslam_ba_debug_step", "soslam_ba_debug_read",
    "soslam_pose_from_global_matrix", "soslam_global_matrix_from_pose",
]
SYNTH_SYMBOLS = [
    "soslam_synth_u64", "soslam_synth_uniform", "soslam_synth_normal", "soslam_synth_ba_config",
    "soslam_synth_ba_count", "soslam_synth_ba_generate", "soslam_synth_pg_config", "soslam_synth_pg_count",
    "soslam_synth_pg_generate",
]

PG_SYMBOLS = [
    "soslam_pg_options_default", "soslam_pg_create", "soslam_pg_destroy", "soslam_pg_set_graph", "soslam_pg_append",
    "soslam_pg_graph_size", "soslam_pg_optimize",
    "soslam_pg_get_estimates", "soslam_pg_get_iteration_log", "soslam_pg_solve", "soslam_pg_debug_linearize",
    "soslam_pg_time_linearize",
]

_lib = None


class SoslamError(RuntimeError):
    def __init__(self, status: int, where: str):
        L = lib()
        msg = L.soslam_last_error().decode() or L.soslam_status_string(status).decode()
        super().__init__(f"{where}: status {status} ({msg})")
        self.status = status


def build(force: bool = False) -> str:
    """Compile the HIP library for gfx950 with hipcc (cross-compiles without a GPU)."""
    cmd = ["make", "-C", CSRC, "-j4"] + (["-B"] if force else [])
    subprocess.check_call(cmd)
    return LIB_PATH


def _pin_hip_runtime():
    """One HIP runtime per process.  The PyTorch wheel carries its own libamdhip64.so (SONAME libamdhip64.so.7) and
    asks for it by FILE name, so a process that loads libsoslam_ba.so first (system runtime, by SONAME) and torch
    afterwards ends up with two runtimes, and the second one to initialise sees no device.  Loading torch's copy
    first makes both requests resolve to the same object, whatever the import order.  Without torch installed
    nothing is preloaded and the system runtime is used, as in a C++ host."""
    import importlib.util
    spec = importlib.util.find_spec("torch")
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        C.CDLL(cand, mode=C.RTLD_GLOBAL)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run __graft_entry__.build() (there is no fallback implementation)")
    _pin_hip_runtime()
    L = C.CDLL(LIB_PATH)
    vp, i32, u32, u64, dbl = C.c_void_p, C.c_int32, C.c_uint32, C.c_uint64, C.c_double
    L.soslam_version.restype = C.c_char_p
    L.soslam_status_string.restype = C.c_char_p
    L.soslam_status_string.argtypes = [C.c_int]
    L.soslam_last_error.restype = C.c_char_p
    L.soslam_ba_options_default.argtypes = [C.POINTER(BaOptions)]
    L.soslam_ba_options_default.restype = None
    L.soslam_ba_create.argtypes = [C.POINTER(BaOptions), C.POINTER(vp)]
    L.soslam_ba_set_options.argtypes = [vp, C.POINTER(BaOptions)]
    L.soslam_ba_destroy.argtypes = [vp]
    L.soslam_ba_destroy.restype = None
    L.soslam_ba_set_projection.argtypes = [vp, vp, vp]
    L.soslam_ba_set_problem.argtypes = [vp, u32, u32, u32, vp, vp, vp, vp]
    L.soslam_ba_set_state.argtypes = [vp, vp, vp]
    L.soslam_ba_get_state.argtypes = [vp, vp, vp]
    L.soslam_ba_solve.argtypes = [vp, C.POINTER(BaSummary)]
    L.soslam_ba_iterate.argtypes = [vp, i32, C.POINTER(BaSummary)]
    L.soslam_ba_get_iteration_log.argtypes = [vp, vp, i32, C.POINTER(i32)]
    L.soslam_ba_optimize.argtypes = [C.POINTER(BaOptions), vp, vp, u32, vp, u32, vp, u32, vp, vp, vp, vp, C.POINTER(BaSummary)]
    L.soslam_ba_set_covisibility.argtypes = [vp, u64, vp, vp]
    L.soslam_ba_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp, i32, i32]
    L.soslam_ba_reduce_buffer_count.argtypes = [vp, C.POINTER(u64)]
    L.soslam_ba_set_reduce_buffer.argtypes = [vp, vp, u64]
    L.soslam_ba_set_host_allreduce.argtypes = [vp, HOST_ALLREDUCE_FN, vp, i32, i32]
    L.soslam_rccl_get_unique_id.argtypes = [vp]
    L.soslam_ba_init_rccl.argtypes = [vp, vp, i32, i32]
    L.soslam_ba_get_state_global.argtypes = [vp, vp, u32, u32, vp]
    L.soslam_ba_shard_range.argtypes = [u32, i32, i32, C.POINTER(u32), C.POINTER(u32)]
    L.soslam_ba_shard_range.restype = None
    L.soslam_ba_time_kernel.argtypes = [vp, i32, i32, C.POINTER(C.c_float)]
    L.soslam_ba_debug_step.argtypes = [vp, dbl]
    L.soslam_ba_debug_read.argtypes = [vp, i32, vp, u64]
    L.soslam_pose_from_global_matrix.argtypes = [vp, vp]
    L.soslam_pose_from_global_matrix.restype = None
    L.soslam_global_matrix_from_pose.argtypes = [vp, vp]
    L.soslam_global_matrix_from_pose.restype = None
    # synthetic workloads
    L.soslam_synth_u64.argtypes = [u64, u64, u64]
    L.soslam_synth_u64.restype = u64
    L.soslam_synth_uniform.argtypes = [u64, u64, u64]
    L.soslam_synth_uniform.restype = dbl
    L.soslam_synth_normal.argtypes = [u64, u64, u64]
    L.soslam_synth_normal.restype = dbl
    L.soslam_synth_ba_config.argtypes = [C.c_int, C.POINTER(SynthBaParams)]
    L.soslam_synth_ba_count.argtypes = [C.POINTER(SynthBaParams), C.POINTER(u32)]
    L.soslam_synth_ba_generate.argtypes = [C.POINTER(SynthBaParams)] + [vp] * 9
    L.soslam_synth_pg_config.argtypes = [C.c_int, C.POINTER(SynthPgParams)]
    L.soslam_synth_pg_count.argtypes = [C.POINTER(SynthPgParams), C.POINTER(u32)]
    L.soslam_synth_pg_generate.argtypes = [C.POINTER(SynthPgParams)] + [vp] * 5
    _lib = L
    return L


def ptr(a):
    """Pointer to a contiguous numpy array (or None)."""
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def check(status: int, where: str) -> None:
    if status != OK:
        raise SoslamError(status, where)
