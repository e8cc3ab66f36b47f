"""oracle/orc_slam.py -- numpy restatement of the step that follows every VO call in the SLAM driver:
eval_covisibility, slam_py/slam_utils.py:18-53 (called at voldor_slam.py:496-504).  TEST INFRASTRUCTURE ONLY: imported by
tests/ (and nothing under voldor_amd/).  Parity pinned against the reference function itself where it can run: it needs
only numpy once its unused `import cv2` is stubbed (tests/golden/gen_golden.py writes ref_covis.npz that way).
"""
from __future__ import annotations

import numpy as np


def eval_covisibility(depth, Tc1c2, K, mask=None, stride=4, return_counts=False):
    """slam_utils.py:18-53, same operation order and dtypes (float32 inputs stay float32 through the matmuls)."""
    h, w = depth.shape
    Iy, Ix = np.mgrid[0:h:stride, 0:w:stride]                                   # :30
    coords_2d = np.stack([Ix, Iy, np.ones_like(Ix)], axis=2).astype(np.float32).reshape(-1, 3)  # :31-32
    coords_3d_shared = (np.linalg.inv(K) @ coords_2d.T).T                       # :33
    coords_3d = np.copy(coords_3d_shared) * depth[::stride, ::stride].reshape(-1, 1)            # :37
    if mask is not None:
        coords_3d = coords_3d[mask[::stride, ::stride].reshape(-1)]             # :38-39
    coords_3d = (Tc1c2[:3, :3] @ coords_3d.T).T                                 # :40
    coords_3d = coords_3d + Tc1c2[:3, 3]                                        # :41
    proj = (K @ coords_3d.T).T                                                  # :43
    proj = proj[proj[:, 2] > 0]                                                 # :44
    proj = proj[:, :2] / proj[:, 2:3]                                           # :45
    vis = (proj[:, 0] > 0) & (proj[:, 0] < w) & (proj[:, 1] > 0) & (proj[:, 1] < h)           # :48
    n_vis = int(np.sum(vis))
    visibility = n_vis / ((w // stride) * (h // stride))                        # :49
    nbx, nby = w // (2 * stride), h // (2 * stride)
    cov, _, _ = np.histogram2d(proj[:, 0], proj[:, 1], bins=(nbx, nby), range=((0, w), (0, h)))  # :51
    n_cov = int(np.sum(cov > 0))
    coverage = n_cov / (nbx * nby)                                              # :52
    score = 2 * (visibility * coverage) / max(visibility + coverage, 1)        # :53
    return (score, n_vis, n_cov) if return_counts else score
