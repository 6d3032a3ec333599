"""Host shim (C++ mirror of the reference's class API) - CPU-only checks: container side effects, float32
conversions and the Dump text format, in C++ (host_selftest) and in the Python twin."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereo_orb_slam_amd", "host")


@pytest.fixture(scope="module")
def host_build(soslam):
    subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return os.path.join(HOST, "build")


def test_cpp_selftest(host_build, tmp_path):
    out = subprocess.run([os.path.join(host_build, "host_selftest"), str(tmp_path)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.strip().endswith("OK"), out.stdout + out.stderr


def test_dump_round_trip_python(soslam, tmp_path):
    from stereo_orb_slam_amd import dump_io, synth
    p = synth.generate_ba(None, n_cam=5, n_pt=60, track_mode=1, track_len=3, spacing=0.8)
    dump_io.write_dump(str(tmp_path), p)
    q = dump_io.read_dump(str(tmp_path), p.proj_l, p.proj_r)
    for k in ("poses_wc", "points", "obs_cam", "obs_pt", "obs_uv"):
        np.testing.assert_array_equal(getattr(p, k), getattr(q, k))
    # the format itself: counts on the first line, 16 / 3 / 7 columns
    lines = open(tmp_path / "constraints.txt").read().split("\n")
    assert int(lines[0]) == p.n_obs and len(lines[1].split()) == 7
    assert len(open(tmp_path / "poses.txt").read().split("\n")[1].split()) == 16


def test_golden_dump_fixture(soslam, golden_dir):
    """A committed 4-frame dump pins the on-disk format (frame-major constraints, camera->world poses)."""
    from stereo_orb_slam_amd import dump_io
    q = dump_io.read_dump(os.path.join(golden_dir, "dump_small"))
    assert (q.n_cam, q.n_pt, q.n_obs) == (4, 30, 69)
    assert np.all(np.diff(q.obs_cam.astype(int)) >= 0)
    np.testing.assert_allclose(q.poses_wc[:, 3, :], np.tile([0, 0, 0, 1], (4, 1)))
    R = q.poses_wc[:, :3, :3].astype(np.float64)
    np.testing.assert_allclose(R @ R.transpose(0, 2, 1), np.tile(np.eye(3), (4, 1, 1)), atol=1e-5)


def test_malformed_dump_is_rejected(soslam, tmp_path):
    from stereo_orb_slam_amd import dump_io, synth
    p = synth.generate_ba(None, n_cam=3, n_pt=10, track_mode=0, track_len=2, spacing=0.8)
    dump_io.write_dump(str(tmp_path), p)
    with open(tmp_path / "points.txt", "w") as f:
        f.write("10\n1 2 3\n")
    with pytest.raises(ValueError):
        dump_io.read_dump(str(tmp_path), p.proj_l, p.proj_r)


def test_oracle_backed_demo_runs_the_schedule(soslam, oracle_lib, tmp_path):
    """The checker of the schedule parity test (tests/test_host_shim_gpu.py) on its own: the unchanged shim + demo sources
    over the oracle (oracle/cabi_over_oracle.c) play the reference's schedule on a small map and reduce its cost."""
    from stereo_orb_slam_amd import dump_io, synth
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "demo"], stdout=subprocess.DEVNULL)
    p = synth.generate_ba(None, n_cam=8, n_pt=400, track_mode=1, track_len=4, spacing=0.9)
    dump_io.write_dump(str(tmp_path / "in"), p)
    os.makedirs(tmp_path / "out")
    out = subprocess.run([os.path.join(ROOT, "oracle", "_build", "ba_demo_oracle"), str(tmp_path / "in"), str(tmp_path / "out"),
                          "--schedule", "4", "--iters", "6", "--quiet"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    q = dump_io.read_dump(str(tmp_path / "out"), p.proj_l, p.proj_r)
    c0 = oracle_lib.cost(p.obs_cam, p.obs_pt, p.obs_uv, p.poses_cw(), p.points_f64(), p.proj_l, p.proj_r)
    c1 = oracle_lib.cost(p.obs_cam, p.obs_pt, p.obs_uv, q.poses_cw(), q.points_f64(), p.proj_l, p.proj_r)
    assert c1 < 0.5 * c0
