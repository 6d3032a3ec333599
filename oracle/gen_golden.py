#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ (run once in the build container; commit the outputs).

The reference holds no test, fixture or golden vector for its BA / pose-graph path and cannot be built
here, so these vectors come from INDEPENDENT restatements written with different
tools than the C oracle and the HIP kernels:

  ba_residual_jacobian.npz   residual formula of /root/reference/src/reprojection_error.h:12-41 restated in
                             torch float64, derivative blocks by torch autograd of the branch taken - what
                             AutoDiffCostFunction<ReprojectionError,4,6,3> (:58) returns up to rounding.
  ba_step_dense.npz          one damped Gauss-Newton step on a small problem solved as ONE dense system
                             (numpy.linalg.solve on J^T J + D, J from autograd): pins Schur elimination,
                             Jacobi scaling, damping and gauge handling.
  ba_minimum_scipy.npz       converged minimum of the robustified problem by scipy.optimize.least_squares
                             (trf) on residual blocks pre-scaled by sqrt(rho(s)/s): pins the fixed point.
  pg_edge_jacobian.npz       SE(3) edge error of g2o's EdgeSE3 restated in torch,
                             Jacobians w.r.t. the 6-dof increments by autograd.

Nothing here is needed at test time on the GPU box; tests read only the .npz files.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

torch.set_default_dtype(torch.float64)
EPS = float(np.finfo(np.float64).eps)


def rotate(w: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """ceres::AngleAxisRotatePoint semantics."""
    th2 = (w * w).sum()
    if float(th2.detach()) > EPS:
        th = torch.sqrt(th2)
        k = w / th
        return x * torch.cos(th) + torch.linalg.cross(k, x) * torch.sin(th) + k * (k @ x) * (1.0 - torch.cos(th))
    return x + torch.linalg.cross(w, x)


def residual(cam: torch.Tensor, pt: torch.Tensor, uv: torch.Tensor, pl: torch.Tensor, pr: torch.Tensor) -> torch.Tensor:
    p = rotate(cam[:3], pt) + cam[3:]
    ph = torch.cat([p, torch.ones(1)])
    out = []
    for P in (pl.reshape(3, 4), pr.reshape(3, 4)):
        h = P @ ph
        s = 1.0 / h[2]
        out += [h[0] * s, h[1] * s]
    return torch.stack(out) - uv


def huber(s: float, delta: float = 1.0):
    b = delta * delta
    if s > b:
        r = np.sqrt(s)
        return 2 * delta * r - b, delta / r
    return s, 1.0


KITTI_L = np.array([718.856, 0, 607.1928, 0, 0, 718.856, 185.2157, 0, 0, 0, 1, 0], np.float32).astype(np.float64)
KITTI_R = KITTI_L.copy()
KITTI_R[3] = np.float64(np.float32(-386.1448))


def gen_residual_jacobian(rng: np.random.Generator):
    cams, pts, uvs, pls, prs, rs, jcs, jps = [], [], [], [], [], [], [], []
    cases = []
    for i in range(48):
        if i < 24:
            w = rng.normal(0, [0.02, 0.3, 1.2][i % 3], 3)
        elif i < 30:
            w = np.zeros(3)                      # exactly zero: first-order branch, derivative -[x]x
        elif i < 36:
            w = rng.normal(0, 1, 3)
            w *= 10.0 ** (-9 + (i - 30)) / np.linalg.norm(w)   # 1e-9 .. 1e-4: straddles sqrt(eps) = 1.5e-8
        else:
            w = rng.normal(0, 1, 3)
            w *= (np.pi - 10.0 ** (-(i - 35))) / np.linalg.norm(w)  # approaching pi
        t = rng.normal(0, 2.0, 3)
        x = np.array([rng.uniform(-15, 15), rng.uniform(-3, 3), rng.uniform(5, 60)])
        pl, pr = KITTI_L.copy(), KITTI_R.copy()
        if i % 4 == 3:  # general 3x4 matrices, not just the rectified rig
            pl[:] = pl + rng.normal(0, 1, 12) * np.array([5, 5, 5, 20, 5, 5, 5, 20, 1e-3, 1e-3, 1e-3, 0.05])
            pr[:] = pr + rng.normal(0, 1, 12) * np.array([5, 5, 5, 20, 5, 5, 5, 20, 1e-3, 1e-3, 1e-3, 0.05])
        cam = np.concatenate([w, t])
        uv = rng.uniform(0, 1200, 4)
        cases.append((cam, x, uv, pl, pr))
    for cam, x, uv, pl, pr in cases:
        ct, xt = torch.tensor(cam, requires_grad=True), torch.tensor(x, requires_grad=True)
        f = lambda c, p: residual(c, p, torch.tensor(uv), torch.tensor(pl), torch.tensor(pr))
        r = f(ct, xt).detach().numpy()
        jc, jp = torch.autograd.functional.jacobian(f, (ct, xt))
        cams.append(cam); pts.append(x); uvs.append(uv); pls.append(pl); prs.append(pr)
        rs.append(r); jcs.append(jc.numpy()); jps.append(jp.numpy())
    np.savez(os.path.join(OUT, "ba_residual_jacobian.npz"), cam=np.array(cams), pt=np.array(pts), uv=np.array(uvs),
             proj_l=np.array(pls), proj_r=np.array(prs), r=np.array(rs), jc=np.array(jcs), jp=np.array(jps))
    print("ba_residual_jacobian.npz:", len(cases), "cases")


def small_problem(rng: np.random.Generator, n_cam=4, n_pt=40):
    """A small stereo BA problem built here with numpy only (no synth.c, no oracle)."""
    cams_true = np.zeros((n_cam, 6))
    for c in range(n_cam):
        cams_true[c, :3] = rng.normal(0, 0.03, 3)
        cams_true[c, 3:] = [-0.1 * c + rng.normal(0, 0.02), rng.normal(0, 0.02), -0.8 * c]
    pts_true = np.stack([rng.uniform(-8, 8, n_pt), rng.uniform(-2, 2, n_pt), rng.uniform(8, 40, n_pt)], -1)
    oc, op, uv = [], [], []
    for c in range(n_cam):
        for p in range(n_pt):
            if rng.uniform() < 0.75:
                r = residual(torch.tensor(cams_true[c]), torch.tensor(pts_true[p]), torch.zeros(4), torch.tensor(KITTI_L),
                             torch.tensor(KITTI_R)).numpy()
                m = r + rng.normal(0, 0.5, 4)
                if rng.uniform() < 0.08:
                    m += rng.uniform(-20, 20, 4)
                oc.append(c); op.append(p); uv.append(m)
    # every point needs at least one observation
    seen = set(op)
    for p in range(n_pt):
        if p not in seen:
            r = residual(torch.tensor(cams_true[0]), torch.tensor(pts_true[p]), torch.zeros(4), torch.tensor(KITTI_L),
                         torch.tensor(KITTI_R)).numpy()
            oc.append(0); op.append(p); uv.append(r)
    order = np.lexsort((op, oc))
    oc, op = np.array(oc, np.uint32)[order], np.array(op, np.uint32)[order]
    uv = np.array(uv)[order].astype(np.float32)
    cams0 = cams_true + rng.normal(0, 1, cams_true.shape) * np.array([0.004] * 3 + [0.04] * 3)
    pts0 = pts_true * (1 + rng.normal(0, 0.015, (n_pt, 1)))
    return oc, op, uv, cams0, pts0


def robust_blocks(oc, op, uv, cams, pts):
    """Corrected residuals and autograd Jacobians (dense), plus the cost."""
    n_cam, n_pt, n = len(cams), len(pts), len(oc)
    J = np.zeros((4 * n, 6 * n_cam + 3 * n_pt))
    r = np.zeros(4 * n)
    cost = 0.0
    pl, pr = torch.tensor(KITTI_L), torch.tensor(KITTI_R)
    for k in range(n):
        c, p = int(oc[k]), int(op[k])
        f = lambda a, b: residual(a, b, torch.tensor(uv[k].astype(np.float64)), pl, pr)
        ct, xt = torch.tensor(cams[c]), torch.tensor(pts[p])
        rk = f(ct, xt).numpy()
        jc, jp = torch.autograd.functional.jacobian(f, (ct, xt))
        rho, rho1 = huber(float(rk @ rk))
        w = np.sqrt(rho1)
        cost += 0.5 * rho
        r[4 * k:4 * k + 4] = w * rk
        J[4 * k:4 * k + 4, 6 * c:6 * c + 6] = w * jc.numpy()
        J[4 * k:4 * k + 4, 6 * n_cam + 3 * p:6 * n_cam + 3 * p + 3] = w * jp.numpy()
    return r, J, cost


def gen_step_dense(rng: np.random.Generator):
    oc, op, uv, cams0, pts0 = small_problem(rng)
    n_cam, n_pt = len(cams0), len(pts0)
    r, J, cost = robust_blocks(oc, op, uv, cams0, pts0)
    free = np.arange(6, 6 * n_cam + 3 * n_pt)  # camera 0 constant (/root/reference/src/bundle_adjuster.cpp:113)
    Jf = J[:, free]
    radius = 1e4
    s = 1.0 / (1.0 + np.sqrt((Jf * Jf).sum(0)))          # Ceres jacobi scaling
    Js = Jf * s
    H = Js.T @ Js
    D = np.clip(np.diag(H), 1e-6, 1e32) / radius
    step_s = np.linalg.solve(H + np.diag(D), -Js.T @ r)
    delta = step_s * s
    model = -(Js @ step_s) @ (r + 0.5 * (Js @ step_s))
    dc = np.zeros((n_cam, 6)); dc.reshape(-1)[6:] = delta[:6 * n_cam - 6]
    dp = delta[6 * n_cam - 6:].reshape(n_pt, 3)
    # reduced camera system in unscaled variables, for comparing S and rhs directly
    Hu = Jf.T @ Jf + np.diag(D / (s * s))
    nc6 = 6 * n_cam - 6
    Bm, Wm, Cm = Hu[:nc6, :nc6], Hu[:nc6, nc6:], Hu[nc6:, nc6:]
    g = Jf.T @ r
    S = Bm - Wm @ np.linalg.solve(Cm, Wm.T)
    rhs = -g[:nc6] + Wm @ np.linalg.solve(Cm, g[nc6:])
    cand_r, _, cand_cost = robust_blocks(oc, op, uv, cams0 + dc, pts0 + dp)
    np.savez(os.path.join(OUT, "ba_step_dense.npz"), obs_cam=oc, obs_pt=op, obs_uv=uv, cams=cams0, pts=pts0,
             proj_l=KITTI_L, proj_r=KITTI_R, radius=radius, cost=cost, dc=dc, dp=dp, model_cost_change=model,
             candidate_cost=cand_cost, S=S, rhs=rhs)
    print("ba_step_dense.npz: n_obs", len(oc), "cost", cost, "model", model, "cand", cand_cost)
    return oc, op, uv, cams0, pts0


def gen_minimum_scipy(problem):
    from scipy.optimize import least_squares
    oc, op, uv, cams0, pts0 = problem
    n_cam, n_pt = len(cams0), len(pts0)
    pl, pr = torch.tensor(KITTI_L), torch.tensor(KITTI_R)
    uvt = torch.tensor(uv.astype(np.float64))
    oct_, opt_ = torch.tensor(oc.astype(np.int64)), torch.tensor(op.astype(np.int64))

    def batched(cams, pts):
        # vectorised restatement (Rodrigues branch only: all angles here are far from zero)
        w, t = cams[oct_, :3], cams[oct_, 3:]
        x = pts[opt_]
        th = torch.linalg.norm(w, dim=1, keepdim=True)
        k = w / th
        y = x * torch.cos(th) + torch.linalg.cross(k, x) * torch.sin(th) + k * (k * x).sum(1, keepdim=True) * (1 - torch.cos(th)) + t
        yh = torch.cat([y, torch.ones(len(y), 1)], 1)
        hl, hr = yh @ pl.reshape(3, 4).T, yh @ pr.reshape(3, 4).T
        return torch.stack([hl[:, 0] / hl[:, 2], hl[:, 1] / hl[:, 2], hr[:, 0] / hr[:, 2], hr[:, 1] / hr[:, 2]], 1) - uvt

    def fun(z):
        cams = torch.cat([torch.tensor(cams0[:1]), torch.tensor(z[:6 * n_cam - 6]).reshape(-1, 6)])
        pts = torch.tensor(z[6 * n_cam - 6:]).reshape(-1, 3)
        r = batched(cams, pts)
        s = (r * r).sum(1)
        rho = torch.where(s > 1.0, 2 * torch.sqrt(s) - 1.0, s)
        scale = torch.sqrt(rho / torch.clamp(s, min=1e-300))   # |scale * r|^2 = rho(s)
        return (r * scale[:, None]).reshape(-1).numpy()

    z0 = np.concatenate([cams0[1:].reshape(-1), pts0.reshape(-1)])
    sol = least_squares(fun, z0, method="trf", xtol=1e-15, ftol=1e-15, gtol=1e-12, max_nfev=2000)
    cams = np.concatenate([cams0[:1], sol.x[:6 * n_cam - 6].reshape(-1, 6)])
    pts = sol.x[6 * n_cam - 6:].reshape(-1, 3)
    np.savez(os.path.join(OUT, "ba_minimum_scipy.npz"), obs_cam=oc, obs_pt=op, obs_uv=uv, cams0=cams0, pts0=pts0,
             proj_l=KITTI_L, proj_r=KITTI_R, cams=cams, pts=pts, cost=sol.cost, status=sol.status, nfev=sol.nfev,
             optimality=sol.optimality)
    print("ba_minimum_scipy.npz: cost", sol.cost, "status", sol.status, "nfev", sol.nfev, "optimality", sol.optimality)


# ---- pose graph (g2o EdgeSE3 semantics) -------------------------------------------

def quat_to_rot(q: torch.Tensor) -> torch.Tensor:
    x, y, z, w = q[0], q[1], q[2], q[3]
    return torch.stack([
        torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)]),
        torch.stack([2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)]),
        torch.stack([2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)])])


def from_vector_mqt(v: torch.Tensor):
    """g2o internal::fromVectorMQT: [t, q.xyz] with q.w = sqrt(1 - |q.xyz|^2)."""
    qv = v[3:]
    w2 = 1.0 - (qv * qv).sum()
    w = torch.sqrt(torch.clamp(w2, min=0.0))
    return quat_to_rot(torch.cat([qv, w.reshape(1)])), v[:3]


def rot_to_quat_vec(R: torch.Tensor) -> torch.Tensor:
    """vector part of the unit quaternion with w >= 0 (g2o toVectorMQT), via the trace branch or Shepperd."""
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if float(tr) > 0:
        s = torch.sqrt(tr + 1.0) * 2
        q = torch.stack([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax([float(R[0, 0]), float(R[1, 1]), float(R[2, 2])]))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = torch.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = [None] * 3
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        q = torch.stack(v + [(R[k, j] - R[j, k]) / s])
    q = q / torch.linalg.norm(q)
    if float(q[3]) < 0:
        q = -q
    return q[:3]


def pg_edge_error(xi, xj, z, di, dj):
    """e = toVectorMQT(Z^-1 (Xi * Exp(di))^-1 (Xj * Exp(dj))), estimates as [t, q]."""
    Ri, ti = quat_to_rot(xi[3:] / torch.linalg.norm(xi[3:])), xi[:3]
    Rj, tj = quat_to_rot(xj[3:] / torch.linalg.norm(xj[3:])), xj[:3]
    Rz, tz = quat_to_rot(z[3:] / torch.linalg.norm(z[3:])), z[:3]
    dRi, dti = from_vector_mqt(di)
    dRj, dtj = from_vector_mqt(dj)
    Ri2, ti2 = Ri @ dRi, Ri @ dti + ti
    Rj2, tj2 = Rj @ dRj, Rj @ dtj + tj
    Rij, tij = Ri2.T @ Rj2, Ri2.T @ (tj2 - ti2)
    Re, te = Rz.T @ Rij, Rz.T @ (tij - tz)
    return torch.cat([te, rot_to_quat_vec(Re)])


def rand_pose(rng):
    q = rng.normal(0, 1, 4)
    q /= np.linalg.norm(q)
    return np.concatenate([rng.normal(0, 5, 3), q])


def gen_pg_edge(rng: np.random.Generator):
    xis, xjs, zs, es, jis, jjs = [], [], [], [], [], []
    for i in range(24):
        xi, xj = rand_pose(rng), rand_pose(rng)
        if i < 16:
            # measurement near the true relative pose (the usual regime)
            Ri, Rj = quat_to_rot(torch.tensor(xi[3:])).numpy(), quat_to_rot(torch.tensor(xj[3:])).numpy()
            Rz, tz = Ri.T @ Rj, Ri.T @ (xj[:3] - xi[:3])
            w = rng.normal(0, 0.05, 3)
            from scipy.spatial.transform import Rotation
            Rz = Rz @ Rotation.from_rotvec(w).as_matrix()
            qz = Rotation.from_matrix(Rz).as_quat()
            z = np.concatenate([tz + rng.normal(0, 0.1, 3), qz])
        else:
            z = rand_pose(rng)  # arbitrary, large error
        f = lambda a, b: pg_edge_error(torch.tensor(xi), torch.tensor(xj), torch.tensor(z), a, b)
        e = f(torch.zeros(6), torch.zeros(6)).numpy()
        ji, jj = torch.autograd.functional.jacobian(f, (torch.zeros(6), torch.zeros(6)))
        xis.append(xi); xjs.append(xj); zs.append(z); es.append(e); jis.append(ji.numpy()); jjs.append(jj.numpy())
    np.savez(os.path.join(OUT, "pg_edge_jacobian.npz"), xi=np.array(xis), xj=np.array(xjs), z=np.array(zs),
             e=np.array(es), ji=np.array(jis), jj=np.array(jjs))
    print("pg_edge_jacobian.npz:", len(es), "cases")


def gen_pg_minimum_scipy(rng: np.random.Generator):
    """Small pose graph (chain + loop edges) minimised by scipy over [t, rotation-vector] per vertex; the robust
    objective sum rho(e^T Omega e) is parametrisation-independent, so its minimum pins the g2o-style solver."""
    from scipy.optimize import least_squares
    from scipy.spatial.transform import Rotation
    n = 24
    truth = np.zeros((n, 7))
    for i in range(n):
        ang = 2 * np.pi * i / n
        truth[i, :3] = [6 * np.cos(ang), 6 * np.sin(ang), 0.3 * np.sin(2 * ang)]
        truth[i, 3:] = Rotation.from_euler("zyx", [ang + np.pi / 2, 0.05 * np.sin(ang), 0.03 * np.cos(ang)]).as_quat()
    edges = [(i, i + 1) for i in range(n - 1)] + [(n - 1, 0), (12, 0), (18, 6), (20, 3)]   # (from, to)
    meas = []
    for a, b in edges:
        Ra, Rb = Rotation.from_quat(truth[a, 3:]), Rotation.from_quat(truth[b, 3:])
        Rz = Ra.inv() * Rb * Rotation.from_rotvec(rng.normal(0, 0.01, 3))
        tz = Ra.inv().apply(truth[b, :3] - truth[a, :3]) + rng.normal(0, 0.05, 3)
        q = Rz.as_quat()
        meas.append(np.concatenate([tz, q if q[3] >= 0 else -q]))
    meas = np.array(meas)
    meas[-1, :3] += [20.0, -15.0, 5.0]                   # one gross outlier (chi2 > 1): the Huber branch
    est0 = truth.copy()
    for i in range(1, n):
        est0[i, :3] += rng.normal(0, 0.15, 3)
        est0[i, 3:] = (Rotation.from_quat(truth[i, 3:]) * Rotation.from_rotvec(rng.normal(0, 0.04, 3))).as_quat()
    info = np.diag([0.01, 0.01, 0.01, 1.0, 1.0, 1.0])
    Lw = np.sqrt(info)

    def edge_err(xa, xb, z):
        return pg_edge_error(torch.tensor(xa), torch.tensor(xb), torch.tensor(z), torch.zeros(6), torch.zeros(6)).numpy()

    def unpack(v):
        est = np.zeros((n, 7))
        est[0] = est0[0]
        for i in range(1, n):
            est[i, :3] = v[6 * (i - 1):6 * (i - 1) + 3]
            q = Rotation.from_rotvec(v[6 * (i - 1) + 3:6 * i]).as_quat()
            est[i, 3:] = q
        return est

    def fun(v):
        est = unpack(v)
        out = []
        for k, (a, b) in enumerate(edges):
            e = edge_err(est[a], est[b], meas[k])
            chi = float(e @ info @ e)
            rho = chi if chi <= 1.0 else 2 * np.sqrt(chi) - 1.0
            out.append(Lw @ e * np.sqrt(rho / max(chi, 1e-300)))
        return np.concatenate(out)

    v0 = np.concatenate([np.concatenate([est0[i, :3], Rotation.from_quat(est0[i, 3:]).as_rotvec()]) for i in range(1, n)])
    sol = least_squares(fun, v0, method="trf", xtol=1e-14, ftol=1e-14, gtol=1e-12, max_nfev=400)
    est = unpack(sol.x)
    np.savez(os.path.join(OUT, "pg_minimum_scipy.npz"), est0=est0, e_from=np.array([a for a, _ in edges], np.uint32),
             e_to=np.array([b for _, b in edges], np.uint32), meas=meas, info=info.reshape(36), est=est, chi2=2 * sol.cost,
             chi2_0=float((fun(v0) ** 2).sum()))
    print("pg_minimum_scipy.npz: chi2", float((fun(v0) ** 2).sum()), "->", 2 * sol.cost, "status", sol.status, "nfev", sol.nfev)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20241004)
    gen_residual_jacobian(rng)
    prob = gen_step_dense(rng)
    gen_minimum_scipy(prob)
    gen_pg_edge(rng)
    gen_pg_minimum_scipy(rng)


if __name__ == "__main__":
    main()
