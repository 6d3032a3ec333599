"""The CPU oracle against the committed golden vectors (oracle/gen_golden.py): independent torch-autograd /
numpy / scipy restatements.  The reference itself holds no fixture for this path (parity unpinned,
oracle/ba_oracle.h), so these are what pins the oracle."""
import os

import numpy as np
import pytest


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name), allow_pickle=False)


def test_residual_and_jacobian_match_autograd(oracle_lib, golden_dir):
    g = _load(golden_dir, "ba_residual_jacobian.npz")
    for i in range(len(g["cam"])):
        r, jc, jp = oracle_lib.residual_jacobian(g["cam"][i], g["pt"][i], g["uv"][i], g["proj_l"][i], g["proj_r"][i])
        r0 = oracle_lib.residual(g["cam"][i], g["pt"][i], g["uv"][i], g["proj_l"][i], g["proj_r"][i])
        np.testing.assert_allclose(r, g["r"][i], rtol=1e-12, atol=1e-9)
        np.testing.assert_array_equal(r, r0)
        # abs 1e-10 rel 1e-12 in f64, loosened near theta ~ sqrt(eps) where the
        # (1 - cos) cancellation of the Rodrigues form costs ~eps/theta in both implementations
        th = np.linalg.norm(g["cam"][i][:3])
        slack = 1.0 if (th == 0 or th > 1e-3) else 1e-16 / th * 1e4
        scale = max(1.0, np.abs(g["jc"][i]).max())
        np.testing.assert_allclose(jc, g["jc"][i], rtol=1e-9, atol=1e-10 * scale + slack * scale)
        np.testing.assert_allclose(jp, g["jp"][i], rtol=1e-9, atol=1e-10 * scale + slack * scale)


def test_zero_angle_branch(oracle_lib):
    # at w = 0 autodiff differentiates y = x + w x x: d/dw = -[x]x, d/dx = I
    pl = np.array([1.0, 0, 0, 0, 0, 1, 0, 0, 0, 0, 0, 1])  # u = x, v = y (d = 1)
    cam = np.zeros(6)
    x = np.array([0.3, -0.7, 2.0])
    r, jc, jp = oracle_lib.residual_jacobian(cam, x, np.zeros(4), pl, pl)
    np.testing.assert_allclose(jp[:2], np.eye(3)[:2], atol=0)
    np.testing.assert_allclose(jc[:2, :3], np.array([[0, x[2], -x[1]], [-x[2], 0, x[0]]]), atol=0)
    np.testing.assert_allclose(jc[:2, 3:], np.eye(3)[:2], atol=0)


def test_huber_branches(oracle_lib):
    np.testing.assert_allclose(oracle_lib.huber(0.25), [0.25, 1.0, 0.0])
    np.testing.assert_allclose(oracle_lib.huber(1.0), [1.0, 1.0, 0.0])          # s == delta^2 is the quadratic branch
    rho = oracle_lib.huber(9.0)
    np.testing.assert_allclose(rho, [2 * 3.0 - 1.0, 1.0 / 3.0, -(1.0 / 3.0) / 18.0])
    rho = oracle_lib.huber(16.0, 2.0)
    np.testing.assert_allclose(rho, [2 * 2 * 4.0 - 4.0, 0.5, -0.5 / 32.0])


def test_one_step_matches_dense_normal_equations(oracle_lib, golden_dir):
    g = _load(golden_dir, "ba_step_dense.npz")
    fixed = np.zeros(len(g["cams"]), np.uint8)
    fixed[0] = 1
    out = oracle_lib.step(g["obs_cam"], g["obs_pt"], g["obs_uv"], g["cams"], g["pts"], g["proj_l"], g["proj_r"], fixed,
                          float(g["radius"]))
    assert out["cost"] == pytest.approx(float(g["cost"]), rel=1e-12)
    np.testing.assert_allclose(out["S"], g["S"], rtol=1e-7, atol=1e-6 * np.abs(g["S"]).max())
    np.testing.assert_allclose(out["rhs"], g["rhs"], rtol=1e-7, atol=1e-7 * np.abs(g["rhs"]).max())
    np.testing.assert_allclose(out["dc"], g["dc"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(out["dp"], g["dp"], rtol=1e-6, atol=1e-8)
    assert out["model_cost_change"] == pytest.approx(float(g["model_cost_change"]), rel=1e-8)
    assert out["candidate_cost"] == pytest.approx(float(g["candidate_cost"]), rel=1e-8)


def test_sharded_reduced_system_equals_unsharded(oracle_lib, golden_dir):
    # per-shard Schur contributions summed over ranks == the unsharded system
    g = _load(golden_dir, "ba_step_dense.npz")
    fixed = np.zeros(len(g["cams"]), np.uint8)
    fixed[0] = 1
    args = (g["obs_cam"], g["obs_pt"], g["obs_uv"], g["cams"], g["pts"], g["proj_l"], g["proj_r"], fixed, 1e4)
    ref = oracle_lib.step(*args)
    for n_rank in (2, 3, 8):
        sh = oracle_lib.step(*args, n_rank=n_rank)
        np.testing.assert_allclose(sh["S"], ref["S"], rtol=1e-10, atol=1e-9 * np.abs(ref["S"]).max())
        np.testing.assert_allclose(sh["rhs"], ref["rhs"], rtol=1e-10, atol=1e-10 * np.abs(ref["rhs"]).max())


def test_solve_reaches_scipy_minimum(oracle_lib, golden_dir):
    g = _load(golden_dir, "ba_minimum_scipy.npz")
    fixed = np.zeros(len(g["cams0"]), np.uint8)
    fixed[0] = 1
    opts = oracle_lib.default_options(max_iterations=200)
    cams, pts, summ, log = oracle_lib.solve(g["obs_cam"], g["obs_pt"], g["obs_uv"], g["cams0"], g["pts0"], g["proj_l"],
                                            g["proj_r"], fixed, opts)
    assert summ.termination in (1, 2), "expected parameter/function tolerance, not the iteration cap"
    # north_star bars: 1e-5 relative on the final residual (cost), 1e-4 on pose parameters
    assert summ.final_cost == pytest.approx(float(g["cost"]), rel=1e-5)
    np.testing.assert_allclose(cams, g["cams"], atol=1e-4)
    np.testing.assert_array_equal(cams[0], g["cams0"][0])  # gauge: first pose untouched
    costs = [e.cost for e in log]
    assert all(b <= a + 1e-12 for a, b in zip(costs, costs[1:])), "monotonic steps (use_nonmonotonic_steps = false)"


def test_structure_only_when_single_fixed_camera(oracle_lib):
    # Optimize(n-1, n): the only pose is constant, every point is an independent 3-variable problem
    # (/root/reference/src/slam.cpp:123)
    rng = np.random.default_rng(3)
    from oracle.gen_golden import KITTI_L, KITTI_R
    cam = np.array([[0.01, -0.02, 0.005, 0.1, 0.0, -0.2]])
    pts_true = np.stack([rng.uniform(-5, 5, 30), rng.uniform(-1, 1, 30), rng.uniform(8, 30, 30)], -1)
    uv = np.zeros((30, 4), np.float32)
    for i in range(30):
        uv[i] = oracle_lib.residual(cam[0], pts_true[i], np.zeros(4), KITTI_L, KITTI_R)
    oc, op = np.zeros(30, np.uint32), np.arange(30, dtype=np.uint32)
    pts0 = pts_true * 1.03
    cams, pts, summ, _ = oracle_lib.solve(oc, op, uv, cam, pts0, KITTI_L, KITTI_R, np.ones(1, np.uint8))
    np.testing.assert_array_equal(cams, cam)
    np.testing.assert_allclose(pts, pts_true, rtol=1e-5)
    assert summ.final_cost < 1e-8


def test_invalid_steps_halve_the_radius(oracle_lib):
    """Ceres' HandleInvalidStep (LevenbergMarquardtStrategy::StepIsInvalid): radius *= 0.5 per invalid step, the
    rejected-step factor untouched; five in a row end the solve.  An unobserved free camera with a zero diagonal floor
    makes every factorisation fail."""
    from stereo_orb_slam_amd import synth
    p = synth.generate_ba(1)
    poses = np.concatenate([p.poses_wc, p.poses_wc[-1:]])
    fixed = np.concatenate([p.cam_fixed, np.zeros(1, np.uint8)])
    q = synth.BaProblem(poses, p.points, p.obs_cam, p.obs_pt, p.obs_uv, p.proj_l, p.proj_r, cam_fixed=fixed)
    args = (q.obs_cam, q.obs_pt, q.obs_uv, q.poses_cw(), q.points_f64(), q.proj_l, q.proj_r, q.cam_fixed)
    _, _, _, log = oracle_lib.solve(*args, oracle_lib.default_options(max_iterations=4, check_termination=0, min_lm_diagonal=0.0))
    assert [e.valid for e in log[1:]] == [0, 0, 0, 0]
    assert [e.radius for e in log[1:]] == [1e4, 5e3, 2.5e3, 1.25e3]
    _, _, s, _ = oracle_lib.solve(*args, oracle_lib.default_options(max_iterations=20, min_lm_diagonal=0.0))
    assert s.iterations == 5 and s.termination == 5   # ORACLE_TERM_INVALID_STEPS


# ---- the controllers against an independent restatement of the whole loop (oracle/gen_controller_golden.py) -----------------------

def _check_ba_trajectory(g, tag, log, line_search_steps, final_cost, cams):
    """log: the iteration records of the oracle or of the device path (entry 0 = the initial point)."""
    n = len(g[tag + "_radius"])
    its = log[1:n + 1]
    assert [int(e.accepted) for e in its] == g[tag + "_accepted"].tolist()
    assert [int(e.valid) for e in its] == g[tag + "_valid"].tolist()
    np.testing.assert_allclose([e.radius for e in its], g[tag + "_radius"], rtol=1e-6)
    np.testing.assert_allclose([e.model_cost_change for e in its], g[tag + "_model"], rtol=1e-6)
    np.testing.assert_allclose([e.candidate_cost for e in its], g[tag + "_cand"], rtol=1e-7)
    np.testing.assert_allclose([e.relative_decrease for e in its], g[tag + "_rho"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose([e.step_norm for e in its], g[tag + "_step_norm"], rtol=1e-6)
    assert line_search_steps == int(g[tag + "_ls_iters"].sum())
    assert final_cost == pytest.approx(float(g[tag + "_final_cost"]), rel=1e-8)
    np.testing.assert_allclose(cams, g[tag + "_cams"], atol=1e-7)


def _ba_options(oracle_lib, g, tag):
    kw = dict(max_iterations=len(g[tag + "_radius"]), check_termination=0, lower_bound=float(g[tag + "_lo"]), upper_bound=float(g[tag + "_hi"]))
    if tag + "_radius0" in g.files:
        kw["initial_radius"] = float(g[tag + "_radius0"])
    return kw


@pytest.mark.parametrize("tag", ["reject", "bounds", "points"])
def test_trust_region_trajectory_matches_the_independent_restatement(oracle_lib, golden_dir, tag):
    """Radius sequence through accepted and rejected steps ("reject": 14 iterations, five rejections, factors 2, 4, growth through
    1 / max(1/3, 1 - (2 rho - 1)^3)) and the bounded problem's Armijo search with cubic interpolation ("bounds": three
    iterations whose full projected step fails the sufficient-decrease test and is contracted to 0.21..; "points": the same with
    every camera constant - the per-frame call, slam.cpp:123 -, two searches that succeed and two that give up after 14 contractions) - every iteration's radius, model
    cost change, candidate cost, step quality and step norm against the numpy / autograd restatement."""
    g = _load(golden_dir, "ba_lm_trajectory.npz")
    fixed = np.zeros(len(g[tag + "_cams0"]), np.uint8)
    fixed[0] = 1
    if tag == "points":
        fixed[:] = 1
    o = oracle_lib.default_options(**_ba_options(oracle_lib, g, tag))
    cams, pts, summ, log = oracle_lib.solve(g["obs_cam"], g["obs_pt"], g["obs_uv"], g[tag + "_cams0"], g[tag + "_pts0"], g["proj_l"],
                                            g["proj_r"], fixed, o)
    _check_ba_trajectory(g, tag, log, summ.line_search_steps, summ.final_cost, cams)
    if tag == "reject":
        assert (g["reject_accepted"] == 0).sum() >= 4
    elif tag == "bounds":
        assert (g[tag + "_alpha"] < 1.0).sum() >= 3 and g[tag + "_ls_iters"].sum() >= 3
    if tag == "points":
        # both outcomes of the search: contractions that succeed, and searches that give up after 14 (the step is then rejected)
        assert (g["points_alpha"] < 1.0).sum() >= 2 and (g["points_ls_iters"] >= 13).sum() >= 2 and (g["points_accepted"] == 0).sum() >= 2
        np.testing.assert_allclose(pts, g["points_pts"], atol=1e-7)


def test_levenberg_schedule_matches_the_independent_restatement(oracle_lib, golden_dir):
    """g2o's lambda / nu schedule on a 12-vertex graph from a poor start (one rejected trial on the way): chi2, lambda and the
    number of trials of every iteration against the numpy / autograd restatement."""
    import ctypes as C
    g = _load(golden_dir, "pg_lm_trajectory.npz")
    L = oracle_lib.lib()
    o = oracle_lib.PgOptions()
    L.oracle_pg_options_default(C.byref(o))
    n_it = len(g["chi2"])
    o.max_iterations = n_it
    est = g["est0"].copy()
    fixed = np.zeros(len(est), np.uint8)
    fixed[0] = 1
    s = oracle_lib.PgSummary()
    log = (oracle_lib.PgIteration * n_it)()
    rc = L.oracle_pg_solve(len(est), len(g["e_from"]), est, fixed, np.ascontiguousarray(g["e_from"]), np.ascontiguousarray(g["e_to"]),
                           np.ascontiguousarray(g["meas"]), np.ascontiguousarray(g["info"]), C.byref(o), C.byref(s), C.cast(log, C.c_void_p))
    assert rc == 0 and s.iterations == n_it
    assert [e.trials for e in log] == g["trials"].tolist() and max(g["trials"]) >= 2
    np.testing.assert_allclose([e.chi2 for e in log], g["chi2"], rtol=1e-6)
    np.testing.assert_allclose([e.lam for e in log], g["lam"], rtol=1e-5)
    np.testing.assert_allclose(est[:, :3], g["est"][:, :3], atol=1e-6)
