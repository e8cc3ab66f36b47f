"""Synthetic analytic scenes for parity tests and bench.py (scene "S").

The reference ships no data, so inputs are generated: two analytic planes
(ground y=1.5, wall z=12 tilted 20 deg) plus a box, exact per-frame ray-cast depth, exact
rigid flow pi(R X + t) - p, end-point noise drawn from the reference's own log-logistic
residual model (gpu-kernels/residual_model.h:6-31), an independently moving patch and 2 %
gross outliers.  Ground-truth poses / depth are returned next to the flows.

Pose convention (voldor/py_export.cpp:56-76): pose f = (rvec|t) maps frame-f coordinates to
frame f+1, X_{f+1} = R_f X_f + t_f; flow f lives on frame f's pixel grid.
"""
from __future__ import annotations

import numpy as np


def rodrigues(rvec: np.ndarray) -> np.ndarray:
    rvec = np.asarray(rvec, dtype=np.float64)
    th = np.linalg.norm(rvec)
    if th < 1e-15:
        return np.eye(3)
    k = rvec / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.cos(th) * np.eye(3) + (1 - np.cos(th)) * np.outer(k, k) + np.sin(th) * Kx


def _raycast(orig: np.ndarray, dirs: np.ndarray) -> np.ndarray:
    """Distance along `dirs` (world frame, un-normalised, shape [...,3]) from `orig` to scene S."""
    big = np.full(dirs.shape[:-1], np.inf)

    def plane(n, d):  # n.X = d
        n = np.asarray(n, dtype=np.float64)
        den = dirs @ n
        with np.errstate(divide="ignore", invalid="ignore"):
            s = (d - orig @ n) / den
        return np.where((den != 0) & (s > 1e-6), s, np.inf)

    out = np.minimum(big, plane([0, 1, 0], 1.5))  # ground
    a = np.deg2rad(20.0)
    nw = np.array([np.sin(a), 0.0, np.cos(a)])
    out = np.minimum(out, plane(nw, nw @ np.array([0, 0, 12.0])))  # tilted wall
    lo = np.array([-2.5, 0.3, 5.0])
    hi = np.array([-1.0, 1.5, 7.0])  # box
    for ax in range(3):
        for bound in (lo[ax], hi[ax]):
            n = np.zeros(3)
            n[ax] = 1.0
            s = plane(n, bound)
            with np.errstate(invalid="ignore"):
                P = orig + dirs * np.where(np.isfinite(s), s, 0.0)[..., None]
            ok = np.isfinite(s)
            for o in range(3):
                if o != ax:
                    ok &= (P[..., o] >= lo[o] - 1e-9) & (P[..., o] <= hi[o] + 1e-9)
            out = np.minimum(out, np.where(ok, s, np.inf))
    return out


def make_scene(w=640, h=480, n_flows=5, fx=320.0, fy=320.0, cx=320.0, cy=240.0, seed=233,
               noise=True, moving_patch=True, outlier_frac=0.02, basefocal=0.0,
               disparity_noise=0.3):
    """Returns dict: flows [N,h,w,2] f32, poses_gt [N,6] f32, depth_gt [h,w] f32 (frame 0),
    K (fx,fy,cx,cy), optionally disparity [h,w] f32 (if basefocal>0)."""
    rng = np.random.default_rng(seed)
    Kinv = np.array([[1 / fx, 0, -cx / fx], [0, 1 / fy, -cy / fy], [0, 0, 1.0]])
    Km = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1.0]])
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    pix = np.stack([xs, ys, np.ones_like(xs)], -1)
    rays = pix @ Kinv.T  # frame-local rays with z=1

    poses = np.zeros((n_flows, 6))
    Rw = np.eye(3)  # world(frame0) -> frame f
    tw = np.zeros(3)
    flows = np.zeros((n_flows, h, w, 2), dtype=np.float32)
    depth0 = None
    for f in range(n_flows):
        t = np.array([0.02, 0.01, 0.25]) + rng.normal(0, 0.01, 3)
        rv = rng.normal(0, 0.004, 3)
        poses[f, :3] = rv
        poses[f, 3:] = t
        R = rodrigues(rv)
        # ray-cast in frame f: X_f = Rw X_w + tw -> origin_w = -Rw^T tw, dir_w = Rw^T ray
        orig = -Rw.T @ tw
        dirs = rays @ Rw  # (Rw^T ray)^T = ray^T Rw
        z = _raycast(orig, dirs)  # since ray has z=1 in frame f, distance parameter == depth
        z = np.where(np.isfinite(z), z, 30.0)
        if f == 0:
            depth0 = z.copy()
        X = rays * z[..., None]
        Xn = X @ R.T + t
        p2 = Xn @ Km.T
        flow = p2[..., :2] / p2[..., 2:3] - pix[..., :2]
        if noise:
            mag = np.linalg.norm(flow, axis=-1)
            g = np.clip(0.5 * mag, 2.0, 100.0)
            c = 1.0 - 0.0022 * g
            s = 0.01 * np.exp(0.09 * g)
            U = rng.uniform(1e-6, 1 - 1e-6, size=mag.shape)
            u = s * (U / (1 - U)) ** (1.0 / c)  # log-logistic in (0.5*EPE)^2
            epe = np.minimum(2.0 * np.sqrt(u), 50.0)
            ang = rng.uniform(0, 2 * np.pi, size=mag.shape)
            flow = flow + np.stack([epe * np.cos(ang), epe * np.sin(ang)], -1)
        if moving_patch:
            ph, pw = max(2, int(h * 90 / 480)), max(2, int(w * 120 / 640))
            y0, x0 = int(h * 0.3), int(w * 0.6)
            flow[y0:y0 + ph, x0:x0 + pw] = np.array([6.0, -3.0])
        if outlier_frac > 0:
            m = rng.uniform(size=(h, w)) < outlier_frac
            flow[m] = rng.uniform(-30, 30, size=(int(m.sum()), 2))
        flows[f] = flow.astype(np.float32)
        # advance world->frame transform
        Rw = R @ Rw
        tw = R @ tw + t
    out = dict(flows=flows, poses_gt=poses.astype(np.float32), depth_gt=depth0.astype(np.float32),
               K=(float(fx), float(fy), float(cx), float(cy)))
    if basefocal > 0:
        disp = basefocal / depth0 + (rng.normal(0, disparity_noise, depth0.shape) if noise else 0.0)
        out["disparity"] = np.maximum(disp, 1e-3).astype(np.float32)
    return out


def pose_errors(poses, poses_gt):
    """Per-pose rotation geodesic [rad] and relative translation error."""
    poses = np.asarray(poses, dtype=np.float64)
    poses_gt = np.asarray(poses_gt, dtype=np.float64)
    n = min(len(poses), len(poses_gt))
    rot, tr = [], []
    for i in range(n):
        Ra, Rb = rodrigues(poses[i, :3]), rodrigues(poses_gt[i, :3])
        c = np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)
        rot.append(float(np.arccos(c)))
        tr.append(float(np.linalg.norm(poses[i, 3:] - poses_gt[i, 3:]) / max(np.linalg.norm(poses_gt[i, 3:]), 1e-12)))
    return np.array(rot), np.array(tr)
