"""Drop-in for the reference's Cython module `pyvoldor_vo` (slam_py/install/pyvoldor_vo.pyx:14-70).

`voldor(...)` has the identical keyword signature, argument meaning, dtype/shape handling and
return dict, so `slam_py/voldor_slam.py:447-457` can `import voldor_amd.pyvoldor as pyvoldor`
unchanged.  It calls py_voldor_wrapper of libvoldor_hip.so through the C-ABI
(include/voldor_hip.h: vk_py_voldor_wrapper); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _arr(a, ndim, name):
    if a is None:
        return None
    a = np.ascontiguousarray(a)
    if a.dtype != np.float32:
        # the .pyx declares np.ndarray[float, ndim=k]: a wrong dtype is a ValueError there too
        raise ValueError(f"Buffer dtype mismatch, expected 'float' but got '{a.dtype}' for {name}")
    if a.ndim != ndim:
        raise ValueError(f"Buffer has wrong number of dimensions (expected {ndim}, got {a.ndim}) for {name}")
    return a


def voldor(flows, fx, fy, cx, cy, basefocal=0, disparity=None, disparity_pconf=None, depth_priors=None,
           depth_prior_poses=None, depth_prior_pconfs=None, config=""):
    if flows is None:
        raise TypeError("Argument 'flows' must not be None")
    flows = _arr(flows, 4, "flows")
    disparity = _arr(disparity, 2, "disparity")
    disparity_pconf = _arr(disparity_pconf, 2, "disparity_pconf")
    depth_priors = _arr(depth_priors, 3, "depth_priors")
    depth_prior_poses = _arr(depth_prior_poses, 2, "depth_prior_poses")
    depth_prior_pconfs = _arr(depth_prior_pconfs, 3, "depth_prior_pconfs")
    N, h, w = flows.shape[0], flows.shape[1], flows.shape[2]
    N_dp = 0 if depth_priors is None else depth_priors.shape[0]

    poses = np.zeros((N, 6), dtype=np.float32)
    poses_covar = np.zeros((N, 6, 6), dtype=np.float32)
    depth = np.zeros((h, w), dtype=np.float32)
    depth_conf = np.zeros((h, w), dtype=np.float32)
    n_registered = C.c_int(0)
    rc = capi.lib().vk_py_voldor_wrapper(
        capi.fp(flows), capi.fp(disparity), capi.fp(disparity_pconf), capi.fp(depth_priors), capi.fp(depth_prior_poses),
        capi.fp(depth_prior_pconfs), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(basefocal),
        C.c_int(N), C.c_int(N_dp), C.c_int(w), C.c_int(h), str(config).encode(), C.byref(n_registered),
        capi.fp(poses), capi.fp(poses_covar), capi.fp(depth), capi.fp(depth_conf))
    capi.check(rc, "py_voldor_wrapper")
    n = n_registered.value
    return {"n_registered": n, "poses": poses[:n], "poses_covar": poses_covar[:n], "depth": depth, "depth_conf": depth_conf}


def voldor_device(flows, fx, fy, cx, cy, basefocal=0, disparity=None, disparity_pconf=None, depth_priors=None,
                  depth_prior_poses=None, depth_prior_pconfs=None, config="", depth_out=None, depth_conf_out=None):
    """Same call for torch CUDA(HIP) tensors already resident in HBM (bench.py). Image-sized
    arguments are torch float32 tensors on the current device; poses come back as numpy."""
    import torch

    def dp(t):
        if t is None:
            return None
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return C.cast(t.data_ptr(), C.POINTER(C.c_float))

    N, h, w = flows.shape[0], flows.shape[1], flows.shape[2]
    N_dp = 0 if depth_priors is None else depth_priors.shape[0]
    poses = np.zeros((N, 6), dtype=np.float32)
    poses_covar = np.zeros((N, 6, 6), dtype=np.float32)
    dpp = None if depth_prior_poses is None else capi.f32(depth_prior_poses)
    n_registered = C.c_int(0)
    rc = capi.lib().vk_voldor_device(
        dp(flows), dp(disparity), dp(disparity_pconf), dp(depth_priors), capi.fp(dpp), dp(depth_prior_pconfs),
        C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(basefocal), C.c_int(N), C.c_int(N_dp),
        C.c_int(w), C.c_int(h), str(config).encode(), C.byref(n_registered), capi.fp(poses), capi.fp(poses_covar),
        dp(depth_out), dp(depth_conf_out))
    capi.check(rc, "vk_voldor_device")
    n = n_registered.value
    return {"n_registered": n, "poses": poses[:n], "poses_covar": poses_covar[:n], "depth": depth_out, "depth_conf": depth_conf_out}


def voldor_device_batch(flows_list, fx, fy, cx, cy, basefocal=0, disparity_list=None, config="", depth_out=None, depth_conf_out=None):
    """Several independent windows (same geometry / config) in flight together on the current device
    (vk_voldor_device_batch): flows_list[b] is a torch float32 CUDA(HIP) tensor [N,h,w,2]; depth_out / depth_conf_out are
    optional lists of [h,w] device tensors.  Returns one result dict per window, as voldor_device does."""
    import torch

    B = len(flows_list)
    N, h, w = flows_list[0].shape[0], flows_list[0].shape[1], flows_list[0].shape[2]
    PF = C.POINTER(C.c_float)

    def arr(ts):
        if ts is None:
            return None
        for t in ts:
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return (PF * B)(*[C.cast(t.data_ptr(), PF) for t in ts])

    for t in flows_list:
        assert tuple(t.shape) == (N, h, w, 2)
    poses = np.zeros((B, N, 6), dtype=np.float32)
    covar = np.zeros((B, N, 6, 6), dtype=np.float32)
    nreg = (C.c_int * B)()
    rc = capi.lib().vk_voldor_device_batch(
        C.c_int(B), arr(flows_list), arr(disparity_list), None, None, None, None, C.c_float(fx), C.c_float(fy), C.c_float(cx),
        C.c_float(cy), C.c_float(basefocal), C.c_int(N), C.c_int(0), C.c_int(w), C.c_int(h), str(config).encode(), nreg,
        capi.fp(poses), capi.fp(covar), arr(depth_out), arr(depth_conf_out))
    capi.check(rc, "vk_voldor_device_batch")
    return [{"n_registered": nreg[b], "poses": poses[b, :nreg[b]], "poses_covar": covar[b, :nreg[b]],
             "depth": None if depth_out is None else depth_out[b], "depth_conf": None if depth_conf_out is None else depth_conf_out[b]}
            for b in range(B)]
