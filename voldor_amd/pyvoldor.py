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


def _check_shapes(flows, disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs):
    """The C side reads w*h*N (resp. N_dp) floats from every buffer: a mismatch must be a Python error, not an overrun."""
    if len(flows) != 4 or flows[3] != 2:
        raise ValueError(f"flows must be [N, h, w, 2], got {tuple(flows)}")
    N, h, w = flows[0], flows[1], flows[2]
    for name, shp in (("disparity", disparity), ("disparity_pconf", disparity_pconf)):
        if shp is not None and tuple(shp) != (h, w):
            raise ValueError(f"{name} must be [h, w] = {(h, w)}, got {tuple(shp)}")
    if disparity_pconf is not None and disparity is None:
        raise ValueError("disparity_pconf without disparity")
    if depth_priors is not None:
        if len(depth_priors) != 3 or tuple(depth_priors[1:]) != (h, w):
            raise ValueError(f"depth_priors must be [N_dp, h, w] with (h, w) = {(h, w)}, got {tuple(depth_priors)}")
        n_dp = depth_priors[0]
        if depth_prior_poses is None or tuple(depth_prior_poses) != (n_dp, 6):
            raise ValueError(f"depth_prior_poses must be [N_dp, 6] = {(n_dp, 6)}, got {None if depth_prior_poses is None else tuple(depth_prior_poses)}")
        if depth_prior_pconfs is not None and tuple(depth_prior_pconfs) != (n_dp, h, w):
            raise ValueError(f"depth_prior_pconfs must be [N_dp, h, w] = {(n_dp, h, w)}, got {tuple(depth_prior_pconfs)}")
    elif depth_prior_poses is not None or depth_prior_pconfs is not None:
        raise ValueError("depth_prior_poses / depth_prior_pconfs without depth_priors")


def last_camera_stats(n):
    """Per-camera statistics of the last window of this thread (voldor/utils.h:41-45): pose_sample_count, pose_density,
    pose_rigidness_density, last_used_ms_iters, last_used_gu_iters."""
    cnt = np.zeros(n, np.int32); dens = np.zeros(n, np.float32); rdens = np.zeros(n, np.float32); ms = np.zeros(n, np.int32); gu = np.zeros(n, np.int32)
    I = C.POINTER(C.c_int)
    capi.check(capi.lib().vk_last_camera_stats(cnt.ctypes.data_as(I), capi.fp(dens), capi.fp(rdens), ms.ctypes.data_as(I), gu.ctypes.data_as(I), int(n)),
               "vk_last_camera_stats")
    return {"pose_sample_count": cnt, "pose_density": dens, "pose_rigidness_density": rdens, "ms_iters": ms, "gu_iters": gu}


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
    _check_shapes(flows.shape, None if disparity is None else disparity.shape, None if disparity_pconf is None else disparity_pconf.shape,
                  None if depth_priors is None else depth_priors.shape, None if depth_prior_poses is None else depth_prior_poses.shape,
                  None if depth_prior_pconfs is None else depth_prior_pconfs.shape)

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
                  depth_prior_poses=None, depth_prior_pconfs=None, config="", depth_out=None, depth_conf_out=None, pose_block_out=None):
    """Same call for torch CUDA(HIP) tensors already resident in HBM (bench.py). Image-sized
    arguments are torch float32 tensors on the current device; poses come back as numpy.
    pose_block_out: optional float32 device tensor of voldor_amd.dist.block_len(N) elements that receives
    [n_registered | poses | covariances] on the device (the send buffer of the multi-GPU pose all-gather)."""
    import torch

    def dp(t):
        if t is None:
            return None
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return C.cast(t.data_ptr(), C.POINTER(C.c_float))

    N, h, w = flows.shape[0], flows.shape[1], flows.shape[2]
    N_dp = 0 if depth_priors is None else depth_priors.shape[0]
    shp = lambda t: None if t is None else tuple(t.shape)  # noqa: E731
    _check_shapes(shp(flows), shp(disparity), shp(disparity_pconf), shp(depth_priors), shp(depth_prior_poses), shp(depth_prior_pconfs))
    for name, t in (("depth_out", depth_out), ("depth_conf_out", depth_conf_out)):
        if t is not None and tuple(t.shape) != (h, w):
            raise ValueError(f"{name} must be [h, w] = {(h, w)}, got {tuple(t.shape)}")
    # The library works on its own (non-blocking) stream: whatever torch still has in flight on the caller's stream (the kernels
    # or H2D copies that produce these tensors) must have finished before it reads them.  Outputs are complete on return.
    torch.cuda.current_stream().synchronize()
    poses = np.zeros((N, 6), dtype=np.float32)
    poses_covar = np.zeros((N, 6, 6), dtype=np.float32)
    dpp = None if depth_prior_poses is None else capi.f32(depth_prior_poses)
    n_registered = C.c_int(0)
    if pose_block_out is not None and pose_block_out.numel() != 1 + 42 * N:
        raise ValueError(f"pose_block_out must hold 1 + 42 N = {1 + 42 * N} floats, got {pose_block_out.numel()}")
    rc = capi.lib().vk_voldor_device_block(
        dp(flows), dp(disparity), dp(disparity_pconf), dp(depth_priors), capi.fp(dpp), dp(depth_prior_pconfs),
        C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(basefocal), C.c_int(N), C.c_int(N_dp),
        C.c_int(w), C.c_int(h), str(config).encode(), C.byref(n_registered), capi.fp(poses), capi.fp(poses_covar),
        dp(depth_out), dp(depth_conf_out), dp(pose_block_out))
    capi.check(rc, "vk_voldor_device_block")
    n = n_registered.value
    return {"n_registered": n, "poses": poses[:n], "poses_covar": poses_covar[:n], "depth": depth_out, "depth_conf": depth_conf_out}


def voldor_sharded(flows, fx, fy, cx, cy, basefocal=0, disparity=None, config="", depth_out=None, depth_conf_out=None, n_flows=None, shape=None):
    """One batch step of the multi-GPU job below the C-ABI (vk_voldor_sharded, include/voldor_hip.h section D): this rank's window
    (torch device tensors as in voldor_device; flows=None when this rank has no sequence in this step -- then n_flows and
    shape=(h, w) describe the job) followed by the RCCL all-gather of the pose records, both inside libvoldor_hip.so.
    Returns (result dict of this rank's window or None, blocks[world, 1 + 42 N] float32: the record of every rank)."""
    import torch

    def dp(t):
        if t is None:
            return None
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        return C.cast(t.data_ptr(), C.POINTER(C.c_float))

    lib = capi.lib()
    world = lib.vk_dist_world()
    if world < 1:
        raise capi.VoldorHipError("voldor_sharded needs voldor_amd.dist.capi_init (vk_dist_init) first")
    if flows is not None:
        N, h, w = flows.shape[0], flows.shape[1], flows.shape[2]
        _check_shapes(tuple(flows.shape), None if disparity is None else tuple(disparity.shape), None, None, None, None)
    else:
        N, (h, w) = int(n_flows), shape
    torch.cuda.current_stream().synchronize()  # see voldor_device
    poses = np.zeros((N, 6), dtype=np.float32)
    poses_covar = np.zeros((N, 6, 6), dtype=np.float32)
    blocks = np.zeros((world, 1 + 42 * N), dtype=np.float32)
    n_registered = C.c_int(0)
    rc = lib.vk_voldor_sharded(dp(flows), dp(disparity), None, None, None, None, C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                               C.c_float(basefocal), C.c_int(N), C.c_int(0), C.c_int(w), C.c_int(h), str(config).encode(), C.byref(n_registered),
                               capi.fp(poses), capi.fp(poses_covar), dp(depth_out), dp(depth_conf_out), capi.fp(blocks))
    capi.check(rc, "vk_voldor_sharded")
    n = n_registered.value
    out = None if flows is None else {"n_registered": n, "poses": poses[:n], "poses_covar": poses_covar[:n], "depth": depth_out, "depth_conf": depth_conf_out}
    return out, blocks


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
    for lst, shape in ((disparity_list, (h, w)), (depth_out, (h, w)), (depth_conf_out, (h, w))):
        if lst is not None:
            if len(lst) != B or any(tuple(t.shape) != shape for t in lst):
                raise ValueError(f"per-window lists must hold {B} tensors of shape {shape}")
    torch.cuda.current_stream().synchronize()  # see voldor_device: the library's streams do not order with torch's
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
