#!/usr/bin/env python3
"""Golden TRAJECTORIES of the two controllers (run once in the build container; commit the outputs).

tests/golden/*.npz of gen_golden.py pin the per-observation blocks, one step and the minimum.  What they do not pin is the
CONTROLLER - Ceres' trust-region radius sequence through accepted and rejected steps, its Armijo / cubic line search on a
problem with active bounds, g2o's lambda / nu schedule - which oracle/ba_oracle.c and csrc/ba_solver.hip (oracle/pg_oracle.c
and csrc/pg_solver.hip) both restate, by the same author.  This script restates the two loops a THIRD time with other tools:
dense numpy algebra on torch-autograd Jacobians, numpy.linalg.solve for the interpolation conditions, numpy.roots (the
eigenvalues of the companion matrix, as Ceres does) for the minimiser of the interpolating polynomial.  The per-iteration
records are committed and compared with BOTH the oracle and the device path.  No reference fixture exists, so
parity stays formally unpinned; a misreading shared by oracle and device would no longer pass unnoticed unless this
restatement shares it too.

  ba_lm_trajectory.npz   two runs of the small problem of gen_golden.small_problem:
                           "reject": a strongly perturbed start, so that full steps overshoot and are rejected
                                     (radius / 2, / 4 .. and back up through 1 / max(1/3, 1 - (2 rho - 1)^3));
                           "bounds": the box [-B, B]^3 cut tight around the points, so that projected steps fail the
                                     sufficient-decrease test and the Armijo search contracts them by cubic interpolation.
  pg_lm_trajectory.npz   ten Levenberg iterations of a 12-vertex pose graph with loop edges from a poor start
                         (lambda_0 = tau max diag H; accepted: lambda *= max(1/3, min(2/3, 1 - (2 rho - 1)^3)), nu = 2;
                          rejected: lambda *= nu, nu *= 2).
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G   # noqa: E402  (the residual restatement and the small problem; nothing of the oracle)

OUT = G.OUT


# ---- Ceres: TrustRegionMinimizer + LevenbergMarquardtStrategy + ArmijoLineSearch (CUBIC) -------------------------------------

def interpolating_polynomial(samples):
    """Coefficients (highest power first) of the polynomial through every value and gradient of the samples."""
    rows, rhs = [], []
    n = sum((1 if s["v_ok"] else 0) + (1 if s["g_ok"] else 0) for s in samples)
    deg = n - 1
    for s in samples:
        if s["v_ok"]:
            rows.append([s["x"] ** (deg - j) for j in range(deg + 1)]); rhs.append(s["v"])
        if s["g_ok"]:
            rows.append([(deg - j) * s["x"] ** (deg - j - 1) if j < deg else 0.0 for j in range(deg + 1)]); rhs.append(s["g"])
    return np.linalg.solve(np.array(rows), np.array(rhs))


def minimize_polynomial(c, lo, hi):
    best_x, best_v = (lo + hi) / 2, np.polyval(c, (lo + hi) / 2)
    for x in (lo, hi):
        v = np.polyval(c, x)
        if v < best_v:
            best_x, best_v = x, v
    d = np.polyder(c)
    while len(d) > 1 and d[0] == 0.0:
        d = d[1:]
    if len(d) > 1:
        for root in np.roots(d):
            x = float(np.real(root))
            if x < lo or x > hi:
                continue
            v = np.polyval(c, x)
            if v < best_v:
                best_x, best_v = x, v
    return best_x


def armijo(evaluate, f0, g0, f1, dmax):
    """Step size by ArmijoLineSearch::DoSearch from 1 (whose cost f1 is known); 1.0 when satisfied at once or when the search fails."""
    suff, max_contr, min_contr, min_step, max_it = 1e-4, 1e-3, 0.6, 1e-9, 20
    if f1 <= f0 + suff * g0:
        return 1.0, 0
    initial = dict(x=0.0, v=f0, g=g0, v_ok=True, g_ok=True)
    previous = dict(x=0.0, v=0.0, g=0.0, v_ok=False, g_ok=False)
    current = evaluate(1.0)
    it = 0
    while (not current["v_ok"]) or current["v"] > f0 + suff * g0 * current["x"]:
        it += 1
        if it >= max_it:
            return 1.0, it
        lo, hi = max_contr * current["x"], min_contr * current["x"]
        if not current["v_ok"]:
            step = min(max(current["x"] * 0.5, lo), hi)
        else:
            smp = [initial, current] + ([previous] if previous["v_ok"] else [])
            step = minimize_polynomial(interpolating_polynomial(smp), lo, hi)
        if step * dmax < min_step:
            return 1.0, it
        previous = current
        current = evaluate(step)
    return current["x"], it


def ba_trajectory(problem, iters, lo, hi, radius0=1e4, structure_only=False):
    """structure_only: every camera constant (BundleAdjuster::Optimize(n-1, n), slam.cpp:123) - the unknowns are the points."""
    oc, op, uv, cams, pts = problem
    cams, pts = cams.copy(), np.clip(pts, lo, hi)
    n_cam, n_pt = len(cams), len(pts)
    nc6 = 0 if structure_only else 6 * n_cam - 6
    free = np.arange(6 * n_cam if structure_only else 6, 6 * n_cam + 3 * n_pt)
    radius, dec = radius0, 2.0
    r, J, cost = G.robust_blocks(oc, op, uv, cams, pts)
    Jf = J[:, free]
    scale = 1.0 / (1.0 + np.sqrt((Jf * Jf).sum(0)))       # jacobi scaling: once, from the first Jacobian
    rec = []
    for _ in range(iters):
        Js = Jf * scale
        H = Js.T @ Js
        D = np.clip(np.diag(H), 1e-6, 1e32) / radius
        step_s = np.linalg.solve(H + np.diag(D), -Js.T @ r)
        delta = step_s * scale
        model = -(Js @ step_s) @ (r + 0.5 * (Js @ step_s))
        e = dict(cost=cost, radius=radius, model=model, valid=1, accepted=0, cand=0.0, rho=0.0, alpha=1.0, ls_iters=0, step_norm=0.0)
        if not (model > 0.0):
            e["valid"] = 0
            radius *= 0.5
            rec.append(e)
            continue
        dc = np.zeros((n_cam, 6))
        if not structure_only:
            dc.reshape(-1)[6:] = delta[:nc6]
        dp = delta[nc6:].reshape(n_pt, 3)
        g = Jf.T @ r
        gdot = float(g @ delta)

        def candidate(a):
            return cams + a * dc, np.clip(pts + a * dp, lo, hi)

        def evaluate(a):
            cc, pp = candidate(a)
            rr, JJ, val = G.robust_blocks(oc, op, uv, cc, pp)
            grad = float((JJ[:, free].T @ rr) @ delta)
            return dict(x=a, v=val, g=grad, v_ok=bool(np.isfinite(val)), g_ok=bool(np.isfinite(grad)))

        cc, pp = candidate(1.0)
        _, _, cand = G.robust_blocks(oc, op, uv, cc, pp)
        alpha, ls_it = armijo(evaluate, cost, gdot, cand, float(np.abs(delta).max()))
        if alpha != 1.0:
            dc, dp = alpha * dc, alpha * dp
            delta = alpha * delta
            cc, pp = cams + dc, np.clip(pts + dp, lo, hi)
            _, _, cand = G.robust_blocks(oc, op, uv, cc, pp)
        rho = (cost - cand) / model
        e.update(cand=cand, rho=rho, alpha=alpha, ls_iters=ls_it,
                 step_norm=float(np.sqrt(((cc - cams)[1:] ** 2).sum() + ((pp - pts) ** 2).sum())))
        if rho > 1e-3:
            cams, pts = cc, pp
            r, J, cost = G.robust_blocks(oc, op, uv, cams, pts)
            Jf = J[:, free]
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rho - 1.0) ** 3))
            dec = 2.0
            e["accepted"] = 1
        else:
            radius /= dec
            dec *= 2.0
        rec.append(e)
    keys = rec[0].keys()
    return {k: np.array([e[k] for e in rec]) for k in keys}, cams, pts, cost


def gen_ba(rng):
    base = G.small_problem(rng, n_cam=6, n_pt=60)
    oc, op, uv, cams0, pts0 = base
    out = dict(obs_cam=oc, obs_pt=op, obs_uv=uv, proj_l=G.KITTI_L, proj_r=G.KITTI_R)
    # (1) rejected steps: a poor start
    cams_bad = cams0 + rng.normal(0, 1, cams0.shape) * np.array([0.25] * 3 + [3.0] * 3)
    cams_bad[0] = cams0[0]
    pts_bad = pts0 * (1 + rng.normal(0, 0.3, (len(pts0), 1)))
    traj, cams, pts, cost = ba_trajectory((oc, op, uv, cams_bad, pts_bad), 14, -1e4, 1e4, radius0=1e7)
    print("reject run: accepted", traj["accepted"].tolist(), "radius", np.array2string(traj["radius"], precision=3), "final cost", cost)
    out.update({"reject_" + k: v for k, v in traj.items()})
    out.update(reject_cams0=cams_bad, reject_pts0=pts_bad, reject_cams=cams, reject_pts=pts, reject_final_cost=cost, reject_lo=-1e4, reject_hi=1e4, reject_radius0=1e7)
    # (2) active bounds: the box cut tight (some coordinates start on the boundary after the projection of the start)
    B = float(os.environ.get('GEN_B', '36.0'))
    traj, cams, pts, cost = ba_trajectory((oc, op, uv, cams0, pts0 * 1.04), 10, -B, B)
    print("bounds run: accepted", traj["accepted"].tolist(), "alpha", np.array2string(traj["alpha"], precision=4), "ls iters", traj["ls_iters"].tolist(),
          "final cost", cost)
    out.update({"bounds_" + k: v for k, v in traj.items()})
    out.update(bounds_cams0=cams0, bounds_pts0=np.clip(pts0 * 1.04, -B, B), bounds_cams=cams, bounds_pts=pts, bounds_final_cost=cost, bounds_lo=-B, bounds_hi=B)
    # (3) structure only (every camera constant: the per-frame call), bounds active: the device runs this one resident - controller
    # and line search in the kernel.  The draw has both outcomes of the search: two iterations whose search gives up after 14
    # contractions (the step stays, is judged and rejected) and two whose search succeeds at once (step size 0.2)
    # (a generator of its own: the pose-graph fixture below keeps the stream it was made with)
    pts_far = pts0 * (1 + np.random.default_rng(20241008).normal(0, 0.25, (len(pts0), 1)))
    traj, cams, pts, cost = ba_trajectory((oc, op, uv, cams0, pts_far), 12, -B, B, structure_only=True)
    print("points run: accepted", traj["accepted"].tolist(), "alpha", np.array2string(traj["alpha"], precision=4), "ls iters", traj["ls_iters"].tolist(),
          "final cost", cost)
    out.update({"points_" + k: v for k, v in traj.items()})
    out.update(points_cams0=cams0, points_pts0=np.clip(pts_far, -B, B), points_cams=cams, points_pts=pts, points_final_cost=cost, points_lo=-B, points_hi=B)
    np.savez(os.path.join(OUT, "ba_lm_trajectory.npz"), **out)


# ---- g2o: OptimizationAlgorithmLevenberg --------------------------------------------------------------------------------------

def pg_system(est, edges, meas, info, fixed0=True):
    n = len(est)
    H = np.zeros((6 * n, 6 * n)); b = np.zeros(6 * n)
    chi = 0.0
    for (i, j), z in zip(edges, meas):
        xi, xj, zz = torch.tensor(est[i]), torch.tensor(est[j]), torch.tensor(z)
        f = lambda di, dj: G.pg_edge_error(xi, xj, zz, di, dj)
        e = f(torch.zeros(6), torch.zeros(6)).numpy()
        Ji, Jj = torch.autograd.functional.jacobian(f, (torch.zeros(6), torch.zeros(6)))
        Ji, Jj = Ji.numpy(), Jj.numpy()
        c2 = float(e @ info @ e)
        rho, rho1 = (c2, 1.0) if c2 <= 1.0 else (2.0 * np.sqrt(c2) - 1.0, 1.0 / np.sqrt(c2))
        chi += rho
        W = rho1 * info
        for (a, Ja) in ((i, Ji), (j, Jj)):
            b[6 * a:6 * a + 6] -= Ja.T @ W @ e
            for (c, Jc) in ((i, Ji), (j, Jj)):
                H[6 * a:6 * a + 6, 6 * c:6 * c + 6] += Ja.T @ W @ Jc
    return H, b, chi


def pg_chi2(est, edges, meas, info):
    chi = 0.0
    for (i, j), z in zip(edges, meas):
        e = G.pg_edge_error(torch.tensor(est[i]), torch.tensor(est[j]), torch.tensor(z), torch.zeros(6), torch.zeros(6)).numpy()
        c2 = float(e @ info @ e)
        chi += c2 if c2 <= 1.0 else 2.0 * np.sqrt(c2) - 1.0
    return chi


def pg_apply(est, x):
    out = est.copy()
    for v in range(1, len(est)):   # vertex 0 fixed
        d = torch.tensor(x[6 * v:6 * v + 6])
        R, t = G.quat_to_rot(torch.tensor(est[v, 3:])), torch.tensor(est[v, :3])
        # g2o internal::fromCompactQuaternion: an increment whose vector part is longer than 1 is no unit quaternion - the
        # rotation increment is then the IDENTITY (isometry3d_mappings.cpp); gen_golden.from_vector_mqt, written for small
        # increments, clamps instead.  The first iterations of this fixture's poor start run into exactly that rule.
        if float((d[3:] * d[3:]).sum()) > 1.0:
            dR, dt = torch.eye(3), d[:3]
        else:
            dR, dt = G.from_vector_mqt(d)
        Rn, tn = R @ dR, R @ dt + t
        q = G.rot_to_quat_vec(Rn).numpy()
        w = np.sqrt(max(0.0, 1.0 - q @ q))
        out[v] = np.concatenate([tn.numpy(), q, [w]])
    return out


def gen_pg(rng):
    n = 12
    est_true = np.zeros((n, 7))
    for v in range(n):
        ang = 2 * np.pi * v / n
        q = np.array([0.0, 0.0, np.sin(ang / 2), np.cos(ang / 2)])
        est_true[v] = np.concatenate([[4 * np.cos(ang), 4 * np.sin(ang), 0.1 * v], q])
    edges = [(v, v + 1) for v in range(n - 1)] + [(11, 0), (0, 2), (3, 5), (7, 10), (11, 1)]
    info = np.diag([0.01, 0.01, 0.01, 1.0, 1.0, 1.0])
    meas = []
    for (i, j) in edges:
        Ri, ti = G.quat_to_rot(torch.tensor(est_true[i, 3:])), torch.tensor(est_true[i, :3])
        Rj, tj = G.quat_to_rot(torch.tensor(est_true[j, 3:])), torch.tensor(est_true[j, :3])
        Rz, tz = Ri.T @ Rj, Ri.T @ (tj - ti)
        q = G.rot_to_quat_vec(Rz).numpy() + rng.normal(0, 0.004, 3)
        meas.append(np.concatenate([tz.numpy() + rng.normal(0, 0.03, 3), q, [np.sqrt(max(0.0, 1 - q @ q))]]))
    meas = np.array(meas)
    # a poor start: drift accumulated along the chain (large enough that some Levenberg trials are rejected)
    est0 = est_true.copy()
    for v in range(1, n):
        est0[v, :3] += rng.normal(0, 3.0, 3) * np.sqrt(v)
        dq = rng.normal(0, 0.55, 3)
        R = G.quat_to_rot(torch.tensor(est0[v, 3:])) @ G.from_vector_mqt(torch.tensor(np.concatenate([np.zeros(3), dq])))[0]
        q = G.rot_to_quat_vec(R).numpy()
        est0[v, 3:] = np.concatenate([q, [np.sqrt(max(0.0, 1 - q @ q))]])
    est = est0.copy()
    lam, nu = 0.0, 2.0
    rec = []
    for it in range(int(os.environ.get('GEN_PG_ITERS', '7'))):
        H, b, chi = pg_system(est, edges, meas, info)
        Hf, bf = H[6:, 6:], b[6:]
        if it == 0:
            lam = 1e-5 * np.abs(np.diag(Hf)).max()
        trials, rho, accepted = 0, 0.0, 0
        current = chi
        while True:
            x = np.linalg.solve(Hf + lam * np.eye(len(bf)), bf)
            cand = pg_apply(est, np.concatenate([np.zeros(6), x]))
            temp = pg_chi2(cand, edges, meas, info)
            scale = float(x @ (lam * x + bf)) + 1e-3
            rho = (current - temp) / scale
            if rho > 0 and np.isfinite(temp):
                alpha = min(1.0 - (2.0 * rho - 1.0) ** 3, 2.0 / 3.0)
                lam *= max(1.0 / 3.0, alpha)
                nu = 2.0
                current = temp
                est = cand
                accepted = 1
            else:
                lam *= nu
                nu *= 2.0
            trials += 1
            if not (rho < 0 and trials < 10):
                break
        rec.append(dict(chi2=current, lam=lam, trials=trials, accepted=accepted))
    print("pg run: chi2", [round(r["chi2"], 6) for r in rec], "trials", [r["trials"] for r in rec])
    np.savez(os.path.join(OUT, "pg_lm_trajectory.npz"), est0=est0, e_from=np.array([a for a, _ in edges], np.uint32),
             e_to=np.array([b for _, b in edges], np.uint32), meas=meas, info=info.reshape(36), est=est,
             chi2=np.array([r["chi2"] for r in rec]), lam=np.array([r["lam"] for r in rec]), trials=np.array([r["trials"] for r in rec]),
             accepted=np.array([r["accepted"] for r in rec]))


def main():
    rng = np.random.default_rng(20241005)
    gen_ba(rng)
    gen_pg(rng)


if __name__ == "__main__":
    main()
