"""The C-ABI library loads and exports every symbol include/*.h declares (no compute calls without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared(header):
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(soslam_[a-z0-9_]+)\s*\(", txt)) - {"soslam_allreduce_fn", "soslam_host_allreduce_fn"})


@pytest.mark.parametrize("header", [h for h in sorted(os.listdir(os.path.join(ROOT, "include"))) if h.endswith(".h")])
def test_every_declared_symbol_is_exported(soslam, header):
    L = soslam.lib()
    names = _declared(header)
    assert names, header
    missing = [n for n in names if not hasattr(L, n)]
    assert not missing, f"{header}: not exported: {missing}"


def test_binding_lists_cover_the_headers(soslam):
    declared = set(_declared("soslam_ba.h")) | set(_declared("soslam_synth.h")) | set(_declared("soslam_pg.h"))
    assert declared == set(soslam.BA_SYMBOLS + soslam.SYNTH_SYMBOLS + soslam.PG_SYMBOLS)


def test_default_options_are_the_reference_configuration(soslam):
    # /root/reference/src/params.h:34-47, /root/reference/src/bundle_adjuster.cpp:14-36, SURVEY.md Appendix A.3
    o = soslam.BaOptions()
    soslam.lib().soslam_ba_options_default(C.byref(o))
    assert o.max_iterations == 50
    assert (o.lower_bound, o.upper_bound) == (-10000.0, 10000.0)
    assert o.huber_delta == 1.0
    assert o.initial_radius == 1e4 and o.min_relative_decrease == 1e-3
    assert o.function_tolerance == 1e-16 and o.gradient_tolerance == 1e-16 and o.parameter_tolerance == 1e-8
    assert o.min_lm_diagonal == 1e-6 and o.max_lm_diagonal == 1e32
    assert o.jacobi_scaling == 1 and o.max_solver_time_seconds == 0.0


def test_no_cpu_fallback(soslam):
    """Without a GPU the product path fails loudly instead of computing on the host."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    st = soslam.lib().soslam_ba_create(None, C.byref(h))
    assert st == soslam.ERR_NO_DEVICE
    assert not h.value
    assert b"no CPU fallback" in soslam.lib().soslam_last_error()


def test_shard_ranges_partition_the_points(soslam):
    L = soslam.lib()
    for n_pt, world in ((100000, 8), (7, 3), (5, 8)):
        prev = 0
        for r in range(world):
            b, e = C.c_uint32(), C.c_uint32()
            L.soslam_ba_shard_range(n_pt, r, world, C.byref(b), C.byref(e))
            assert b.value == prev and e.value >= b.value
            prev = e.value
        assert prev == n_pt


def test_pose_matrix_round_trip(soslam):
    # MatrixToPose / PoseToMatrix in float32 (/root/reference/src/math_utils.h:12-41)
    L = soslam.lib()
    rng = np.random.default_rng(0)
    for _ in range(20):
        pose = np.concatenate([rng.normal(0, 0.4, 3), rng.normal(0, 3, 3)])
        T = np.zeros(16, np.float32)
        L.soslam_global_matrix_from_pose(pose.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p))
        back = np.zeros(6)
        L.soslam_pose_from_global_matrix(T.ctypes.data_as(C.c_void_p), back.ctypes.data_as(C.c_void_p))
        np.testing.assert_allclose(back, pose, atol=2e-5)
        R = T.reshape(4, 4)[:3, :3].astype(np.float64)
        np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-5)
    # zero rotation vector: identity rotation, no NaN
    T = np.zeros(16, np.float32)
    pose = np.array([0, 0, 0, 1.0, 2.0, 3.0])
    L.soslam_global_matrix_from_pose(pose.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p))
    assert np.isfinite(T).all()
    np.testing.assert_allclose(T.reshape(4, 4)[:3, :3], np.eye(3), atol=0)
    np.testing.assert_allclose(T.reshape(4, 4)[:3, 3], [-1, -2, -3], atol=1e-6)


def test_no_kernel_is_over_its_register_budget():
    """The build's own check (csrc/Makefile runs it after the link): no gfx950 kernel of the library spills vector registers or
    uses scratch memory, except the two that are listed and tested that way.  VERDICT round 2, task 5: the windowed Schur kernel
    sits two registers under its budget, and the variant that went over it was only noticed on the device."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_kernel_resources.py"),
                        os.path.join(root, "stereo_orb_slam_amd", "libsoslam_ba.so")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
