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
    T = capi.f32(np.asarray(Tc1c2, np.float32).reshape(4, 4))
    Kf = capi.f32(np.asarray(K, np.float32).reshape(3, 3))
    score = C.c_float(0)
    counts = (C.c_int * 2)()
    rc = capi.lib().vk_eval_covisibility(dp, mp, capi.fp(T), capi.fp(Kf), C.c_int(w), C.c_int(h), C.c_int(stride), C.byref(score), counts)
    capi.check(rc, "vk_eval_covisibility")
    return (score.value, counts[0], counts[1]) if return_counts else score.value
