"""Reader / writer of the reference's Dump text format (VisualOdometer::Dump,
/root/reference/src/visual_odometer.cpp:446-505): poses.txt (camera->world 4x4 per frame), points.txt,
constraints.txt (frame_id point_id u_l v_l u_r v_r sigma).  The C++ twin is stereo_orb_slam_amd/host/dump_io.h."""
from __future__ import annotations

import os

import numpy as np

from .synth import BaProblem


def write_dump(folder: str, prob: BaProblem, digits: int = 9) -> None:
    """digits=9 round-trips float32; the reference itself streams at the default 6 significant digits."""
    os.makedirs(folder, exist_ok=True)
    fmt = f"%.{digits}g"
    with open(os.path.join(folder, "poses.txt"), "w") as f:
        f.write(f"{prob.n_cam}\n")
        np.savetxt(f, np.asarray(prob.poses_wc, np.float32).reshape(prob.n_cam, 16), fmt=fmt)
    with open(os.path.join(folder, "points.txt"), "w") as f:
        f.write(f"{prob.n_pt}\n")
        np.savetxt(f, np.asarray(prob.points, np.float32), fmt=fmt)
    with open(os.path.join(folder, "constraints.txt"), "w") as f:
        f.write(f"{prob.n_obs}\n")
        uv = np.asarray(prob.obs_uv, np.float32)
        for k in range(prob.n_obs):
            f.write(f"{int(prob.obs_cam[k])} {int(prob.obs_pt[k])} " + " ".join(fmt % v for v in uv[k]) + " 1\n")


def read_dump(folder: str, proj_l=None, proj_r=None) -> BaProblem:
    def body(name, cols):
        with open(os.path.join(folder, name)) as f:
            n = int(f.readline().split()[0])
            a = np.loadtxt(f, dtype=np.float64, ndmin=2) if n else np.zeros((0, cols))
        if a.shape != (n, cols):
            raise ValueError(f"{name}: header says {n} rows of {cols}, found {a.shape}")
        return a

    poses = body("poses.txt", 16).astype(np.float32).reshape(-1, 4, 4)
    pts = body("points.txt", 3).astype(np.float32)
    con = body("constraints.txt", 7)
    oc, op = con[:, 0].astype(np.uint32), con[:, 1].astype(np.uint32)
    if len(con) and (oc.max() >= len(poses) or op.max() >= len(pts)):
        raise ValueError("constraints.txt refers to a frame or point that does not exist")
    if proj_l is None:
        from . import synth
        kitti = synth.generate_ba(None, n_cam=2, n_pt=2, track_mode=0, track_len=1)
        proj_l, proj_r = kitti.proj_l, kitti.proj_r
    return BaProblem(poses, pts, oc, op, con[:, 2:6].astype(np.float32), np.asarray(proj_l, np.float64),
                     np.asarray(proj_r, np.float64), name=os.path.basename(os.path.normpath(folder)))
