"""Python mirror of the B-inner kernel library (include/gpu_kernels.h = the reference's
gpu-kernels/gpu_kernels.h:11-58): same function names, argument order and meaning, host numpy
arrays in and out, NULL (None) protocol preserved.  Thin ctypes calls into libvoldor_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .capi import f32, fp, ptr_table


def optimize_depth_gpu(flows, rigidnesses, depth_priors, depth_prior_pconfs, depth_prior_confs, depth, K, Rs, ts,
                       dp_Rs, dp_ts, abs_resize_factor, N, N_dp, w, h, basefocal, n_rand_samples, global_prop_step,
                       local_prop_width, lambda_, omega, disp_delta, delta, fb_smooth, s0_ems_prob, no_change_prob,
                       range_factor, update_rigidness_only, download=True):
    """Inputs may be None (= reuse the device copy). Returns (depth, rigidnesses[N,h,w], confs[N_dp,h,w])."""
    t_fl, k1 = ptr_table(None if flows is None else [flows[i] for i in range(N)])
    t_rg, k2 = ptr_table(None if rigidnesses is None else [rigidnesses[i] for i in range(N)])
    t_pr, k3 = ptr_table(None if depth_priors is None else [depth_priors[i] for i in range(N_dp)])
    t_pc, k4 = ptr_table(None if depth_prior_pconfs is None else [depth_prior_pconfs[i] for i in range(N_dp)])
    t_cf, k5 = ptr_table(None if depth_prior_confs is None else [depth_prior_confs[i] for i in range(N_dp)])
    t_R, k6 = ptr_table(None if Rs is None else [np.reshape(Rs[i], 9) for i in range(N)])
    t_t, k7 = ptr_table(None if ts is None else [np.reshape(ts[i], 3) for i in range(N)])
    t_dR, k8 = ptr_table(None if dp_Rs is None else [np.reshape(dp_Rs[i], 9) for i in range(N_dp)])
    t_dt, k9 = ptr_table(None if dp_ts is None else [np.reshape(dp_ts[i], 3) for i in range(N_dp)])
    d_in = None if depth is None else f32(depth)
    Kf = None if K is None else f32(K).reshape(9)
    o_depth = np.zeros((h, w), np.float32) if download else None
    o_rig = np.zeros((max(N, 1), h, w), np.float32) if download else None
    o_cf = np.zeros((max(N_dp, 1), h, w), np.float32) if download else None
    t_org, _ = ptr_table(None if not download else [o_rig[i] for i in range(N)]) if N > 0 else (None, None)
    t_ocf, _ = ptr_table(None if not download else [o_cf[i] for i in range(N_dp)]) if N_dp > 0 else (None, None)
    rc = capi.lib().vk_optimize_depth_gpu(
        t_fl, t_rg, t_org, t_pr, t_pc, t_cf, t_ocf, fp(d_in), fp(o_depth), fp(Kf), t_R, t_t, t_dR, t_dt,
        C.c_float(abs_resize_factor), N, N_dp, w, h, C.c_float(basefocal), n_rand_samples, global_prop_step, local_prop_width,
        C.c_float(lambda_), C.c_float(omega), C.c_float(disp_delta), C.c_float(delta), int(bool(fb_smooth)),
        C.c_float(s0_ems_prob), C.c_float(no_change_prob), C.c_float(range_factor), int(bool(update_rigidness_only)))
    capi.check(rc, "optimize_depth_gpu")
    if not download:
        return None, None, None
    return o_depth, o_rig[:N], o_cf[:N_dp]


def collect_p3p_instances(flows, rigidnesses, depth, K, Rs, ts, N, w, h, active_idx, rigidness_thresh=0.5,
                          rigidness_sum_thresh=1.0, sample_min_depth=0.1, sample_max_depth=1000.0, max_trace_on_flow=3):
    t_fl, k1 = ptr_table(None if flows is None else [flows[i] for i in range(N)])
    t_rg, k2 = ptr_table(None if rigidnesses is None else [rigidnesses[i] for i in range(N)])
    t_R, k3 = ptr_table(None if Rs is None else [np.reshape(Rs[i], 9) for i in range(N)])
    t_t, k4 = ptr_table(None if ts is None else [np.reshape(ts[i], 3) for i in range(N)])
    d_in = None if depth is None else f32(depth)
    Kf = None if K is None else f32(K).reshape(9)
    p2 = np.zeros((h, w, 2), np.float32)
    p3 = np.zeros((h, w, 3), np.float32)
    rc = capi.lib().vk_collect_p3p_instances(t_fl, t_rg, fp(d_in), fp(Kf), t_R, t_t, fp(p2), fp(p3), N, w, h, active_idx,
                                             C.c_float(rigidness_thresh), C.c_float(rigidness_sum_thresh),
                                             C.c_float(sample_min_depth), C.c_float(sample_max_depth), max_trace_on_flow)
    capi.check(rc, "collect_p3p_instances")
    return p2, p3


def get_compacted_points(max_points):
    p2 = np.zeros((max_points, 2), np.float32)
    p3 = np.zeros((max_points, 3), np.float32)
    n = capi.lib().vk_get_compacted_points(fp(p2), fp(p3), max_points)
    if n < 0:
        raise capi.VoldorHipError(f"vk_get_compacted_points failed {n}")
    return p2[:n], p3[:n]


def _solve(fn, p3s, p2s, K, n_poses):
    p3s, p2s = f32(p3s), f32(p2s)
    rv = np.zeros((n_poses, 3), np.float32)
    tv = np.zeros((n_poses, 3), np.float32)
    Kf = f32(K).reshape(9)
    capi.check(fn(fp(p3s), fp(p2s), fp(rv), fp(tv), fp(Kf), p3s.shape[0], n_poses), "solve_batch_p3p")
    return rv, tv


def solve_batch_p3p_lambdatwist_gpu(p3s, p2s, K, n_poses=8192):
    return _solve(capi.lib().vk_solve_batch_p3p_lambdatwist_gpu, p3s, p2s, K, n_poses)


def solve_batch_p3p_lambdatwist_f64_gpu(p3s, p2s, K, n_poses=8192):
    return _solve(capi.lib().vk_solve_batch_p3p_lambdatwist_f64_gpu, p3s, p2s, K, n_poses)


def solve_batch_p3p_ap3p_gpu(p3s, p2s, K, n_poses=8192):
    return _solve(capi.lib().vk_solve_batch_p3p_ap3p_gpu, p3s, p2s, K, n_poses)


def meanshift_gpu(space, kernel_var, io_mean, use_external_init_mean, epsilon=1e-5, max_iters=100, max_init_trials=20,
                  good_init_confidence=0.5):
    space = f32(space)
    N, dims = space.shape
    mean = f32(io_mean).copy()
    conf = C.c_float(0)
    iters = C.c_int(0)
    rc = capi.lib().vk_meanshift_gpu(fp(space), C.c_float(kernel_var), fp(mean), C.byref(conf), C.byref(iters),
                                     int(bool(use_external_init_mean)), N, dims, C.c_float(epsilon), max_iters,
                                     max_init_trials, C.c_float(good_init_confidence))
    capi.check(rc, "meanshift_gpu")
    return mean, conf.value, iters.value


def fit_robust_gaussian(space, io_mean, io_covar, trunc_sigma=3.0, covar_reg_lambda=1e-3, epsilon=1e-5, max_iters=100):
    """Returns (rc, mean, covar, density, iters); rc==0 iff the fit is reliable (as in the reference)."""
    space = f32(space)
    N, dims = space.shape
    mean = f32(io_mean).copy()
    covar = f32(io_covar).copy()
    dens = C.c_float(0)
    iters = C.c_int(0)
    rc = capi.lib().vk_fit_robust_gaussian(fp(space), fp(mean), fp(covar), C.c_float(trunc_sigma), C.c_float(covar_reg_lambda),
                                           C.byref(dens), C.byref(iters), N, dims, C.c_float(epsilon), max_iters)
    if rc not in (0, 1):
        capi.check(rc, "fit_robust_gaussian")
    return rc, mean, covar, dens.value, iters.value


def fb_smooth_gpu(maps, s0_ems_prob=0.5, no_change_prob=0.9):
    m = f32(maps).copy()
    n, h, w = m.shape
    rc = capi.lib().vk_fb_smooth(fp(m), n, w, h, C.c_float(s0_ems_prob), C.c_float(no_change_prob))
    return rc, m


def align_frame_init_gpu(images, depths, weights, K, vbf, crw):
    """gpu_kernels.h:60-66.  images [N,h,w] or None, depths / weights [N,h,w], K 3x3."""
    depths = f32(depths); weights = f32(weights)
    N, h, w = depths.shape
    images = None if images is None else f32(images)
    PF = C.POINTER(C.c_float)

    def table(a):
        return None if a is None else (PF * N)(*[a[i].ctypes.data_as(PF) for i in range(N)])

    Kf = f32(np.asarray(K, np.float32).reshape(9))
    return capi.lib().vk_align_frame_init_gpu(table(images), table(depths), table(weights), fp(Kf), C.c_float(vbf), C.c_float(crw), N, w, h), (N, h, w)


def align_frame_eval_gpu(shape, ref_fid, tar_fid, params_ref, params_tar, want_jacobian=True, apply_weights=True):
    """gpu_kernels.h:68-74 -> rc, residual [h,w], jacobian [h,w,9] (None if not wanted)."""
    _, h, w = shape
    res = np.zeros((h, w), np.float32)
    jac = np.zeros((h, w, 9), np.float32) if want_jacobian else None
    pr = None if params_ref is None else f32(params_ref)
    pt = None if params_tar is None else f32(params_tar)
    rc = capi.lib().vk_align_frame_eval_gpu(int(ref_fid), int(tar_fid), fp(pr), fp(pt), fp(res), fp(jac), int(bool(apply_weights)))
    return rc, res, jac


def gblur_gpu(src, sigma, ksize=0):
    src = f32(src)
    d, h, w = src.shape
    dst = np.zeros_like(src)
    rc = capi.lib().vk_gblur(fp(src), fp(dst), w, h, d, C.c_float(sigma), ksize)
    return rc, dst


def estimate_pose_epipolar(flow, K):
    flow = f32(flow)
    h, w, _ = flow.shape
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    rc = capi.lib().vk_estimate_pose_epipolar(fp(flow), fp(f32(K).reshape(9)), w, h, fp(R), fp(t))
    return rc == 0, R.reshape(3, 3), t


def estimate_pose_epipolar5(flow, K):
    """Host path of the two-view bootstrap with the five-point minimal solver (vk_estimate_pose_epipolar5; --bootstrap_points 5)."""
    flow = f32(flow)
    h, w, _ = flow.shape
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    rc = capi.lib().vk_estimate_pose_epipolar5(fp(flow), fp(f32(K).reshape(9)), w, h, fp(R), fp(t))
    return rc == 0, R.reshape(3, 3), t


def fivept_solve(q1, q2):
    """The five-point solver alone (vk_fivept_solve, host code): q1, q2 [5, 2] normalised correspondences -> [n, 3, 3] essential matrices."""
    q1 = np.ascontiguousarray(q1, np.float64).reshape(5, 2); q2 = np.ascontiguousarray(q2, np.float64).reshape(5, 2)
    Es = np.zeros((10, 9), np.float64)
    D = C.POINTER(C.c_double)
    n = capi.lib().vk_fivept_solve(q1.ctypes.data_as(D), q2.ctypes.data_as(D), Es.ctypes.data_as(D))
    return Es[:n].reshape(n, 3, 3)


def bootstrap_gpu(flow, K, points=8):
    """GPU bootstrap kernels of the window pipeline (pose by LMedS + closed-form depth); points = 8 | 5 (minimal solver)."""
    flow = f32(flow)
    h, w, _ = flow.shape
    R = np.zeros(9, np.float32)
    t = np.zeros(3, np.float32)
    d = np.zeros((h, w), np.float32)
    capi.check(capi.lib().vk_bootstrap_gpu_points(fp(flow), fp(f32(K).reshape(9)), w, h, int(points), fp(R), fp(t), fp(d)), "vk_bootstrap_gpu_points")
    return R.reshape(3, 3), t, d


def estimate_depth_closed_form(flow, K, R, t):
    flow = f32(flow)
    h, w, _ = flow.shape
    d = np.zeros((h, w), np.float32)
    capi.check(capi.lib().vk_estimate_depth_closed_form(fp(flow), fp(f32(K).reshape(9)), fp(f32(R).reshape(9)),
                                                        fp(f32(t).reshape(3)), w, h, fp(d)), "estimate_depth_closed_form")
    return d


def set_rand_epoch(e: int):
    capi.check(capi.lib().vk_set_rand_epoch(C.c_uint(e)), "vk_set_rand_epoch")


def get_rand_epoch() -> int:
    return capi.lib().vk_get_rand_epoch()


def set_strict_math(on: bool):
    """Process-wide default of the strict-math mode (include/voldor_hip.h: vk_set_strict_math)."""
    capi.check(capi.lib().vk_set_strict_math(1 if on else 0), "vk_set_strict_math")


def get_strict_math() -> bool:
    return bool(capi.lib().vk_get_strict_math())


def set_reference_svd(on: bool):
    """Process-wide default of rodrigues() through the reference's approximate SVD (include/voldor_hip.h: vk_set_reference_svd)."""
    capi.check(capi.lib().vk_set_reference_svd(1 if on else 0), "vk_set_reference_svd")


def get_reference_svd() -> bool:
    return bool(capi.lib().vk_get_reference_svd())


def set_reference_rng(on: bool):
    """Process-wide default of the cuRAND XORWOW streams in strict mode (include/voldor_hip.h: vk_set_reference_rng)."""
    capi.check(capi.lib().vk_set_reference_rng(1 if on else 0), "vk_set_reference_rng")


def set_reference_tex(on: bool):
    """Process-wide default of CUDA's linear texture filter in strict mode (include/voldor_hip.h: vk_set_reference_tex)."""
    capi.check(capi.lib().vk_set_reference_tex(1 if on else 0), "vk_set_reference_tex")
