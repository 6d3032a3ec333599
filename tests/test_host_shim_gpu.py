"""BundleAdjuster::Optimize of the C++ host shim, end to end on the GPU: Dump in -> Optimize -> Dump out,
checked against the CPU oracle on the same float32 inputs."""
import os
import re
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "stereo_orb_slam_amd", "host")


@pytest.fixture(scope="module")
def demo(soslam):
    exe = os.path.join(HOST, "build", "ba_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    return exe


def _run(demo, src, dst, *args):
    os.makedirs(dst, exist_ok=True)
    out = subprocess.run([demo, str(src), str(dst), *map(str, args)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    return out.stdout


def test_optimize_window_matches_oracle(demo, oracle_lib, tmp_path):
    from stereo_orb_slam_amd import dump_io, synth
    p = synth.generate_ba(1)
    dump_io.write_dump(str(tmp_path / "in"), p)
    text = _run(demo, tmp_path / "in", tmp_path / "out", "--iters", 15, "--quiet")
    m = re.search(r"RESULT status (\d+) initial (\S+) final (\S+) iterations (\d+)", text)
    assert m and int(m.group(1)) == 0
    q = dump_io.read_dump(str(tmp_path / "out"), p.proj_l, p.proj_r)

    o = oracle_lib.default_options(max_iterations=15)
    ocams, opts_, osum, _ = oracle_lib.solve(p.obs_cam, p.obs_pt, p.obs_uv, p.poses_cw(), p.points_f64(), p.proj_l, p.proj_r,
                                             p.cam_fixed, o)
    assert float(m.group(2)) == pytest.approx(osum.initial_cost, rel=1e-5)
    assert float(m.group(3)) == pytest.approx(osum.final_cost, rel=1e-5)      # north_star: 1e-5 on the final cost
    # written-back map state: poses (camera->world float32) and points (float32) against the oracle's doubles
    np.testing.assert_allclose(q.poses_cw()[1:], ocams[1:], atol=1e-4)        # north_star: 1e-4 on pose parameters
    np.testing.assert_allclose(q.points, opts_, rtol=1e-4, atol=1e-3)
    # the first pose of the window is constant: only float32 re-orthonormalisation may touch it
    np.testing.assert_allclose(q.poses_wc[0], p.poses_wc[0], atol=2e-6)
    # cost of the written-back state, re-evaluated by the oracle, is the optimum up to float32 rounding
    c = oracle_lib.cost(p.obs_cam, p.obs_pt, p.obs_uv, q.poses_cw(), q.points_f64(), p.proj_l, p.proj_r)
    assert c == pytest.approx(osum.final_cost, rel=1e-3)


def test_sharded_shim_path_one_rank(demo, tmp_path):
    """BundleAdjuster::EnableSharding - the shim's multi-GPU path (RCCL unique id, communicator on the handle, point shard
    with the job-wide block pattern, all-reduces inside the LM iteration, global read-back of all ranks' points) - with
    the one rank a one-GPU box allows.  A collective being attached only changes WHERE the sums are taken: the written-back
    map must be the plain path's, byte for byte."""
    from stereo_orb_slam_amd import dump_io, synth
    p = synth.generate_ba(1)
    dump_io.write_dump(str(tmp_path / "in"), p)
    _run(demo, tmp_path / "in", tmp_path / "plain", "--iters", 10, "--quiet")
    _run(demo, tmp_path / "in", tmp_path / "sharded", "--iters", 10, "--quiet", "--shard-one-rank")
    for name in ("poses.txt", "points.txt"):
        assert open(tmp_path / "plain" / name, "rb").read() == open(tmp_path / "sharded" / name, "rb").read(), name


def test_half_open_range_and_untouched_frames(demo, tmp_path):
    """Optimize(2, 7) adjusts frames 2..6 only; frame 2 is the constant one; points never seen in the window
    keep their position (/root/reference/src/bundle_adjuster.cpp:62,113)."""
    from stereo_orb_slam_amd import dump_io, synth
    p = synth.generate_ba(1)
    dump_io.write_dump(str(tmp_path / "in"), p)
    _run(demo, tmp_path / "in", tmp_path / "out", "--start", 2, "--end", 7, "--iters", 5, "--quiet")
    q = dump_io.read_dump(str(tmp_path / "out"), p.proj_l, p.proj_r)
    moved = np.abs(q.poses_wc - p.poses_wc).reshape(p.n_cam, -1).max(1)
    assert (moved[[0, 1, 7, 8, 9]] < 2e-6).all() and moved[2] < 2e-6
    assert (moved[3:7] > 1e-5).all()
    in_window = np.isin(p.obs_cam, np.arange(2, 7))
    seen = np.zeros(p.n_pt, bool)
    seen[p.obs_pt[in_window]] = True
    np.testing.assert_array_equal(q.points[~seen], p.points[~seen])
    assert (np.abs(q.points[seen] - p.points[seen]).max(1) > 0).mean() > 0.9


def test_slam_schedule_matches_the_oracle(demo, oracle_lib, tmp_path):
    """/root/reference/src/slam.cpp:121-129: after every frame a structure-only pass on that frame (its pose constant), every
    `interval` frames a window of 2 x interval frames - 24 frames, interval 4: 24 per-frame calls and 6 windows, every call
    writing float32 poses and points back into the map the next one reads.  The SAME shim and demo sources run once over
    the device library and once over the oracle (oracle/cabi_over_oracle.c -> ba_demo_oracle); the final maps must agree
    at the north_star's bars: 1e-5 relative on the final cost, 1e-4 on the pose parameters."""
    from stereo_orb_slam_amd import dump_io, synth
    ref_demo = os.path.join(ROOT, "oracle", "_build", "ba_demo_oracle")
    if not os.path.exists(ref_demo):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "demo"], stdout=subprocess.DEVNULL)
    p = synth.generate_ba(None, n_cam=24, n_pt=1800, track_mode=1, track_len=5, spacing=0.8)
    dump_io.write_dump(str(tmp_path / "in"), p)
    _run(demo, tmp_path / "in", tmp_path / "gpu", "--schedule", 4, "--iters", 8, "--quiet")
    _run(ref_demo, tmp_path / "in", tmp_path / "ref", "--schedule", 4, "--iters", 8, "--quiet")
    q = dump_io.read_dump(str(tmp_path / "gpu"), p.proj_l, p.proj_r)
    r = dump_io.read_dump(str(tmp_path / "ref"), p.proj_l, p.proj_r)
    args = (p.obs_cam, p.obs_pt, p.obs_uv)
    c0 = oracle_lib.cost(*args, p.poses_cw(), p.points_f64(), p.proj_l, p.proj_r)
    cq = oracle_lib.cost(*args, q.poses_cw(), q.points_f64(), p.proj_l, p.proj_r)
    cr = oracle_lib.cost(*args, r.poses_cw(), r.points_f64(), p.proj_l, p.proj_r)
    assert cr < 0.7 * c0                                              # the schedule did its work
    assert cq == pytest.approx(cr, rel=1e-5)
    np.testing.assert_allclose(q.poses_cw(), r.poses_cw(), atol=1e-4)
    np.testing.assert_allclose(q.points, r.points, rtol=1e-4, atol=1e-3)


def _tq_to_mat(v):
    from scipy.spatial.transform import Rotation
    T = np.tile(np.eye(4), (len(v), 1, 1))
    T[:, :3, :3] = Rotation.from_quat(v[:, 3:]).as_matrix()
    T[:, :3, 3] = v[:, :3]
    return T


def _mat_to_tq(T):
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(T[:, :3, :3]).as_quat()
    return np.concatenate([T[:, :3, 3], q], 1)


def test_pose_graph_optimizer_matches_oracle(oracle_lib, tmp_path):
    """PoseGraphOptimizer::Optimize() of the host shim (loop measurements supplied as data, trailing global BA
    skipped) against the oracle on the same float32 graph."""
    import ctypes as C
    from stereo_orb_slam_amd import dump_io, synth
    exe = os.path.join(HOST, "build", "pg_demo")
    if not os.path.exists(exe):
        subprocess.check_call(["make", "-C", HOST], stdout=subprocess.DEVNULL)
    g = synth.generate_pg(6)
    n = len(g.est)
    T = _tq_to_mat(g.est).astype(np.float32)
    prob = synth.BaProblem(T, np.zeros((0, 3), np.float32), np.zeros(0, np.uint32), np.zeros(0, np.uint32),
                           np.zeros((0, 4), np.float32), np.zeros(12), np.zeros(12))
    dump_io.write_dump(str(tmp_path / "in"), prob)
    loops = [(int(g.e_to[k]), int(g.e_from[k]), _tq_to_mat(g.meas[k:k + 1])[0].astype(np.float32)) for k in range(n - 1, len(g.e_from))]
    with open(tmp_path / "loops.txt", "w") as f:
        for a, b, M in loops:
            f.write(f"{a} {b} " + " ".join("%.9g" % x for x in M.reshape(-1)) + "\n")
    os.makedirs(tmp_path / "out", exist_ok=True)
    out = subprocess.run([exe, str(tmp_path / "in"), str(tmp_path / "out"), "--loops", str(tmp_path / "loops.txt"), "--skip-ba",
                          "--quiet", "--graph", str(tmp_path / "graph.txt")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-1500:]
    m = re.search(r"RESULT status 0 initial (\S+) final (\S+) iterations (\d+)", out.stdout)
    assert m
    q = dump_io.read_dump(str(tmp_path / "out"), np.zeros(12), np.zeros(12))

    # the same graph for the oracle: vertices and odometry measurements from the float32 matrices the shim saw
    T64 = T.astype(np.float64)
    est0 = _mat_to_tq(T64)
    rel = np.linalg.inv(T64[:-1]) @ T64[1:]
    ef = np.concatenate([np.arange(n - 1), [b for _, b, _ in loops]]).astype(np.uint32)
    et = np.concatenate([np.arange(1, n), [a for a, _, _ in loops]]).astype(np.uint32)
    meas = np.concatenate([_mat_to_tq(rel), _mat_to_tq(np.array([M for _, _, M in loops], np.float64))])
    L = oracle_lib.lib()
    o = oracle_lib.PgOptions()
    L.oracle_pg_options_default(C.byref(o))
    s = oracle_lib.PgSummary()
    est = np.ascontiguousarray(est0)
    fixed = np.zeros(n, np.uint8)
    fixed[0] = 1
    assert L.oracle_pg_solve(n, len(ef), est, fixed, ef, et, np.ascontiguousarray(meas), np.ascontiguousarray(g.info), C.byref(o),
                             C.byref(s), None) == 0
    assert float(m.group(1)) == pytest.approx(s.initial_chi2, rel=1e-3, abs=1e-9)
    assert float(m.group(2)) == pytest.approx(s.final_chi2, rel=1e-3, abs=1e-9)
    got = _mat_to_tq(q.poses_wc.astype(np.float64))
    np.testing.assert_allclose(got[:, :3], est[:, :3], atol=2e-3)
    sign = np.sign((got[:, 3:] * est[:, 3:]).sum(1))[:, None]
    np.testing.assert_allclose(got[:, 3:] * sign, est[:, 3:], atol=1e-4)
    # the saved graph file has the reference's SavePoseGraph layout: header, 7 numbers per vertex, 9 per edge
    lines = open(tmp_path / "graph.txt").read().strip().split("\n")
    nv, ne = map(int, lines[0].split())
    assert (nv, ne) == (n, len(ef)) and len(lines[1].split()) == 7 and len(lines[1 + nv].split()) == 9
