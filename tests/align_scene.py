"""Key-frame fixtures for the frame-alignment tests: scene S (voldor_amd/synth.py) ray-cast from arbitrary cam->world poses,
with a texture that is a function of the 3-D world point (photometric consistency across views)."""
import numpy as np

from voldor_amd import synth


def keyframes(w=160, h=120, n=3, seed=0):
    rng = np.random.default_rng(seed)
    fx = fy = w / 2.0
    cx, cy = w / 2.0, h / 2.0
    K = np.array([[fx, 0, cx], [0, fy, cy], [0, 0, 1]], np.float32)
    Kinv = np.linalg.inv(K.astype(np.float64))
    xs, ys = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64))
    rays = np.stack([xs, ys, np.ones_like(xs)], -1) @ Kinv.T
    params, depths, images = [], [], []
    for f in range(n):
        rv = rng.normal(0, 0.01, 3) * (f > 0)
        t = np.array([0.05, -0.02, 0.3]) * f + rng.normal(0, 0.01, 3) * (f > 0)
        R = synth.rodrigues(rv)  # cam -> world
        d = synth._raycast(t, rays @ R.T)
        d = np.where(np.isfinite(d), d, 50.0)
        Xw = t + (rays * d[..., None]) @ R.T
        img = 0.5 + 0.25 * np.sin(1.3 * Xw[..., 0] + 0.4 * Xw[..., 2]) + 0.2 * np.cos(2.1 * Xw[..., 1] + 0.3 * Xw[..., 2])
        params.append(np.concatenate([rv, t, [0.0, 0.0, 0.0]]).astype(np.float32))
        depths.append(d.astype(np.float32))
        images.append(img.astype(np.float32))
    weights = rng.uniform(0.5, 1.0, (n, h, w)).astype(np.float32)
    return dict(K=K, params=np.stack(params), depths=np.stack(depths), images=np.stack(images), weights=weights, vbf=0.5 * fx, crw=4.0)
