"""Mirror of the reference's slam_py/slam_utils.py entry points that sit right after the VO call
(voldor_slam.py:496-504), over the C-ABI.  eval_covisibility keeps the reference signature; depth / mask may be numpy
arrays or torch CUDA(HIP) tensors (the maps a window leaves in HBM)."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def eval_covisibility(depth, Tc1c2, K, mask=None, stride=4, return_counts=False):
    """slam_utils.py:18-53 -> covisibility score in [0, 1]."""
    def ptr(a, ctype, np_dtype):
        if a is None:
            return None, None
        if hasattr(a, "data_ptr"):  # torch tensor already in HBM
            assert a.is_contiguous()
            return C.cast(a.data_ptr(), C.POINTER(ctype)), a
        keep = np.ascontiguousarray(a, dtype=np_dtype)
        return keep.ctypes.data_as(C.POINTER(ctype)), keep

    h, w = depth.shape
    dp, _d = ptr(depth, C.c_float, np.float32)
    if mask is not None and hasattr(mask, "data_ptr"):
        import torch
        mask = mask.to(torch.uint8)
    mp, _m = ptr(mask, C.c_ubyte, np.uint8)
    if hasattr(depth, "data_ptr") or hasattr(mask, "data_ptr"):  # the library's stream does not order with torch's (pyvoldor.voldor_device)
        import torch
        torch.cuda.current_stream().synchronize()
    T = capi.f32(np.asarray(Tc1c2, np.float32).reshape(4, 4))
    Kf = capi.f32(np.asarray(K, np.float32).reshape(3, 3))
    score = C.c_float(0)
    counts = (C.c_int * 2)()
    rc = capi.lib().vk_eval_covisibility(dp, mp, capi.fp(T), capi.fp(Kf), C.c_int(w), C.c_int(h), C.c_int(stride), C.byref(score), counts)
    capi.check(rc, "vk_eval_covisibility")
    return (score.value, counts[0], counts[1]) if return_counts else score.value


# ---- pose conversions either side of the VO call (slam_py/slam_utils.py:55-93; voldor_slam.py:440 builds the depth-prior poses
# with T44_to_T6, :492 turns the returned poses into Tc1c2 with T6_to_T44, :520 re-orthonormalises the running pose).  The
# reference goes through cv2.Rodrigues; cv2 is not a dependency here, so its algorithm is restated in numpy (calib3d
# cvRodrigues2: double arithmetic, matrix input orthonormalised by SVD first, theta from acos of the clamped trace).
def _rodrigues_vec_to_mat(r):
    r = np.asarray(r, np.float64).reshape(3)
    theta = float(np.sqrt(r @ r))
    if theta < np.finfo(np.float64).eps:
        return np.eye(3)
    k = r / theta
    c, s = np.cos(theta), np.sin(theta)
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return c * np.eye(3) + (1 - c) * np.outer(k, k) + s * Kx


def _rodrigues_mat_to_vec(R):
    u, _, vt = np.linalg.svd(np.asarray(R, np.float64).reshape(3, 3))
    R = u @ vt
    r = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    s = np.sqrt((r @ r) * 0.25)
    c = min(max((np.trace(R) - 1) * 0.5, -1.0), 1.0)
    theta = np.arccos(c)
    if s < 1e-5:
        if c > 0:
            return np.zeros(3)
        t = np.sqrt(np.maximum((np.diag(R) + 1) * 0.5, 0.0))  # theta = pi: axis from the diagonal, signs from the off-diagonals
        t[1] *= -1.0 if R[0, 1] < 0 else 1.0
        t[2] *= -1.0 if R[0, 2] < 0 else 1.0
        if abs(t[0]) < abs(t[1]) and abs(t[0]) < abs(t[2]) and (R[1, 2] > 0) != (t[1] * t[2] > 0):
            t[2] = -t[2]
        return t * (theta / np.sqrt(t @ t))
    return r * (theta / (2 * s))


def polish_T44(pose):
    """In place: replace the rotation block by its nearest rotation (slam_utils.py:55-57)."""
    u, _, vt = np.linalg.svd(pose[:3, :3])
    pose[:3, :3] = u @ vt


def T44_to_T6(poses):
    """[4,4] -> [6] or [N,4,4] -> [N,6] (rvec | t), dtype kept (slam_utils.py:59-75)."""
    poses = np.asarray(poses)
    if poses.ndim == 2:
        ret = np.zeros((6,), poses.dtype)
        ret[:3] = _rodrigues_mat_to_vec(poses[:3, :3])
        ret[3:] = poses[:3, 3]
        return ret
    if poses.ndim == 3:
        return np.stack([T44_to_T6(p) for p in poses]) if len(poses) else np.zeros((0, 6), poses.dtype)
    raise ValueError("Invalid Input")


def T6_to_T44(poses):
    """[6] -> [4,4] or [N,6] -> [N,4,4], dtype kept (slam_utils.py:77-93)."""
    poses = np.asarray(poses)
    if poses.ndim == 1:
        ret = np.zeros((4, 4), poses.dtype)
        ret[:3, :3] = _rodrigues_vec_to_mat(poses[:3])
        ret[:3, 3] = poses[3:6]
        ret[3, 3] = 1
        return ret
    if poses.ndim == 2:
        return np.stack([T6_to_T44(p) for p in poses]) if len(poses) else np.zeros((0, 4, 4), poses.dtype)
    raise ValueError("Invalid Input")
