"""ctypes wrapper of the CPU ORACLE (oracle/liborc.so) and of oracle/_ref.

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package voldor_amd never imports this module.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_F = C.POINTER(C.c_float)


def _fp(a):
    return None if a is None else a.ctypes.data_as(_F)


def build(force: bool = False) -> None:
    """Compile liborc.so (and oracle/_ref when /root/reference exists)."""
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_model.c", "orc_pose.c", "orc_voldor.c", "orc_align.c", "orc_lambdatwist_impl.h", "orc.h", "orc_math.h",
                                         "../voldor_amd/csrc/vk_strict_math.h", "../voldor_amd/csrc/vk_ref_svd.h", "../voldor_amd/csrc/vk_ref_cuda.h")]
    stale = force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    ref_so = os.path.join(_HERE, "_ref", "libvoldor_ref.so")
    want_ref = False
    if os.path.isdir("/root/reference/lambdatwist"):  # only where the reference checkout exists; the built .so travels
        ref_srcs = [os.path.join(_HERE, f) for f in ("ref_wrap.cpp", "ref_wrap_kernels.cpp", "ref_wrap_host.cpp", "ref_prep.pl", "Makefile")]
        for d, _, files in os.walk(os.path.join(_HERE, "ref_stubs")):
            ref_srcs += [os.path.join(d, f) for f in files]
        want_ref = force or not os.path.exists(ref_so) or any(os.path.getmtime(s) > os.path.getmtime(ref_so) for s in ref_srcs)
    if stale or want_ref:
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []) + (["liborc.so", "ref"] if want_ref else ["liborc.so"]))


class OdParams(C.Structure):
    _fields_ = [("N", C.c_int), ("N_dp", C.c_int), ("w", C.c_int), ("h", C.c_int),
                ("K", C.c_float * 9),
                ("Rs", (C.c_float * 9) * 16), ("ts", (C.c_float * 3) * 16),
                ("dp_Rs", (C.c_float * 9) * 16), ("dp_ts", (C.c_float * 3) * 16),
                ("abs_resize_factor", C.c_float), ("basefocal", C.c_float),
                ("n_rand_samples", C.c_int), ("global_prop_step", C.c_int), ("local_prop_width", C.c_int),
                ("lambda_", C.c_float), ("omega", C.c_float), ("disp_delta", C.c_float), ("delta", C.c_float),
                ("fb_smooth", C.c_int), ("s0_ems_prob", C.c_float), ("no_change_prob", C.c_float),
                ("range_factor", C.c_float), ("update_rigidness_only", C.c_int)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(os.path.join(_HERE, "liborc.so"))
        L = _lib
        L.orc_rng.restype = C.c_uint32
        L.orc_rng.argtypes = [C.c_uint32] * 3
        L.orc_u01.restype = C.c_float
        L.orc_u01.argtypes = [C.c_uint32]
        for n, na in (("orc_fun_fmag_c", 1), ("orc_fun_fmag_scale", 1), ("orc_fisk_dist_pdf", 3),
                      ("orc_fun_rigidness", 6), ("orc_fun_depth_rigidness", 5)):
            getattr(L, n).restype = C.c_float
            getattr(L, n).argtypes = [C.c_float] * na
        L.orc_fit_robust_gaussian.restype = C.c_int
        L.orc_voldor.restype = C.c_int
        L.orc_compact_p3p.restype = C.c_int
        L.orc_lambdatwist_p4p.restype = C.c_int
        L.orc_ap3p_p4p.restype = C.c_int
        L.orc_gblur.restype = C.c_int
        L.orc_get_max_threads.restype = C.c_int
        L.orc_estimate_pose_epipolar.restype = C.c_int
    return _lib


def ref():
    """oracle/_ref: the reference's own host-compilable math (None if not built)."""
    global _ref
    if _ref is None:
        p = os.path.join(_HERE, "_ref", "libvoldor_ref.so")
        if not os.path.exists(p):
            try:
                build()
            except Exception:
                pass
        if not os.path.exists(p):
            return None
        _ref = C.CDLL(p)
        for n, na in (("ref_fun_fmag_c", 1), ("ref_fun_fmag_scale", 1), ("ref_fisk_dist_pdf", 3),
                      ("ref_fun_rigidness", 6), ("ref_fun_depth_rigidness", 5)):
            getattr(_ref, n).restype = C.c_float
            getattr(_ref, n).argtypes = [C.c_float] * na
        for n in ("ref_lambdatwist_p4p_f", "ref_lambdatwist_p4p_d"):
            getattr(_ref, n).restype = C.c_int
            getattr(_ref, n).argtypes = [_F, _F, C.c_float, C.c_float, C.c_float, C.c_float, _F, _F]
    return _ref


def set_threads(n: int):
    lib().orc_set_threads(C.c_int(n))


def max_threads() -> int:
    return lib().orc_get_max_threads()


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def make_od_params(N, N_dp, w, h, K, Rs=None, ts=None, dp_Rs=None, dp_ts=None, abs_resize_factor=1.0,
                   basefocal=0.0, n_rand_samples=10, global_prop_step=8, local_prop_width=32,
                   lambda_=0.15, omega=0.15, disp_delta=-1.0, delta=0.5, fb_smooth=1, s0_ems_prob=0.5,
                   no_change_prob=0.9, range_factor=1.0, update_rigidness_only=0) -> OdParams:
    p = OdParams()
    p.N, p.N_dp, p.w, p.h = N, N_dp, w, h
    Km = f32(K).reshape(9)
    for i in range(9):
        p.K[i] = Km[i]
    for arr, dst, n in ((Rs, p.Rs, 9), (ts, p.ts, 3), (dp_Rs, p.dp_Rs, 9), (dp_ts, p.dp_ts, 3)):
        if arr is not None:
            a = f32(arr).reshape(-1, n)
            for f in range(a.shape[0]):
                for i in range(n):
                    dst[f][i] = a[f, i]
    p.abs_resize_factor, p.basefocal = abs_resize_factor, basefocal
    p.n_rand_samples, p.global_prop_step, p.local_prop_width = n_rand_samples, global_prop_step, local_prop_width
    p.lambda_, p.omega, p.disp_delta, p.delta = lambda_, omega, disp_delta, delta
    p.fb_smooth, p.s0_ems_prob, p.no_change_prob = fb_smooth, s0_ems_prob, no_change_prob
    p.range_factor, p.update_rigidness_only = range_factor, update_rigidness_only
    return p


def optimize_depth(p: OdParams, flows, rig, depth, priors=None, pconfs=None, confs=None, rand_epoch=0):
    """Returns (depth, rig, confs, cost, rand_epoch) -- copies, inputs untouched."""
    flows = f32(flows)
    rig = f32(rig).copy()
    depth = f32(depth).copy()
    cost = np.zeros_like(depth)
    priors = None if priors is None else f32(priors)
    pconfs = None if pconfs is None else f32(pconfs)
    confs = None if confs is None else f32(confs).copy()
    ep = C.c_uint32(rand_epoch)
    lib().orc_optimize_depth(C.byref(p), _fp(flows), _fp(rig), _fp(priors), _fp(pconfs), _fp(confs),
                             _fp(depth), _fp(cost), C.byref(ep))
    return depth, rig, confs, cost, ep.value


def compute_cost_map(p: OdParams, flows, rig, depth, priors=None, pconfs=None, confs=None):
    flows, rig, depth = f32(flows), f32(rig), f32(depth)
    cost = np.zeros_like(depth)
    lib().orc_compute_cost_map(C.byref(p), _fp(flows), _fp(rig), _fp(None if priors is None else f32(priors)),
                               _fp(None if pconfs is None else f32(pconfs)),
                               _fp(None if confs is None else f32(confs)), _fp(depth), _fp(cost))
    return cost


def fb_smooth(maps, s0_ems_prob=0.5, no_change_prob=0.9):
    m = f32(maps).copy()
    n, h, w = m.shape
    lib().orc_fb_smooth(_fp(m), n, w, h, C.c_float(s0_ems_prob), C.c_float(no_change_prob))
    return m


def gblur(src, sigma, ksize=0):
    s = f32(src)
    d, h, w = s.shape
    o = np.zeros_like(s)
    rc = lib().orc_gblur(_fp(s), _fp(o), w, h, d, C.c_float(sigma), ksize)
    return rc, o


def collect_p3p(flows, rig, depth, K, Rs, ts, active_idx, rigidness_thresh=0.5, rigidness_sum_thresh=1.0,
                sample_min_depth=0.1, sample_max_depth=1000.0, max_trace_on_flow=3):
    flows, rig, depth = f32(flows), f32(rig), f32(depth)
    N, h, w = rig.shape
    Rs16 = np.zeros((16, 9), np.float32)
    ts16 = np.zeros((16, 3), np.float32)
    Rs16[:N] = f32(Rs).reshape(-1, 9)[:N]
    ts16[:N] = f32(ts).reshape(-1, 3)[:N]
    p2 = np.zeros((h, w, 2), np.float32)
    p3 = np.zeros((h, w, 3), np.float32)
    Kf = f32(K).reshape(9)
    lib().orc_collect_p3p(_fp(flows), _fp(rig), _fp(depth), _fp(Kf), _fp(Rs16), _fp(ts16), _fp(p2), _fp(p3),
                          N, w, h, active_idx, C.c_float(rigidness_thresh), C.c_float(rigidness_sum_thresh),
                          C.c_float(sample_min_depth), C.c_float(sample_max_depth), max_trace_on_flow)
    return p2, p3


def compact_p3p(p2_map, p3_map):
    p2, p3 = f32(p2_map).reshape(-1, 2), f32(p3_map).reshape(-1, 3)
    o2, o3 = np.zeros_like(p2), np.zeros_like(p3)
    n = lib().orc_compact_p3p(_fp(p2), _fp(p3), p2.shape[0], _fp(o2), _fp(o3))
    return o2[:n].copy(), o3[:n].copy()


def pose_sample_indices(idx, n_pts):
    out = (C.c_int * 4)()
    lib().orc_pose_sample_indices(idx, n_pts, out)
    return list(out)


def lambdatwist_p4p(y, x, fx, fy, cx, cy, use_double=False):
    y, x = f32(y).reshape(8), f32(x).reshape(12)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    ok = lib().orc_lambdatwist_p4p(_fp(y), _fp(x), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy),
                                   int(use_double), _fp(R), _fp(t))
    return ok, R.reshape(3, 3), t


def ap3p_p4p(y, x, fx, fy, cx, cy):
    y, x = f32(y).reshape(8), f32(x).reshape(12)
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    ok = lib().orc_ap3p_p4p(_fp(y), _fp(x), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), _fp(R), _fp(t))
    return ok, R.reshape(3, 3), t


def rodrigues(R):
    R = f32(R).reshape(9)
    r = np.zeros(3, np.float32)
    lib().orc_rodrigues(_fp(R), _fp(r))
    return r


def reference_project_rotation(R):
    """rodrigues.h:82-108 alone: U V^T of the reference's approximate SVD (voldor_amd/csrc/vk_ref_svd.h)."""
    R = f32(R).reshape(9)
    Q = np.zeros(9, np.float32)
    lib().orc_reference_project_rotation(_fp(R), _fp(Q))
    return Q.reshape(3, 3)


def set_reference_svd(on):
    """orc_rodrigues through the reference's approximate SVD (the product's --reference_svd 1) instead of the exact polar factor (D8)."""
    lib().orc_set_reference_svd(1 if on else 0)


def rotmat_to_angle_axis(R):
    R = f32(R).reshape(9)
    r = np.zeros(3, np.float32)
    lib().orc_rotmat_to_angle_axis(_fp(R), _fp(r))
    return r


def rvec_to_rotmat(rvec):
    r = f32(rvec).reshape(3)
    R = np.zeros(9, np.float32)
    lib().orc_rvec_to_rotmat(_fp(r), _fp(R))
    return R.reshape(3, 3)


def solve_batch_p3p(pts3, pts2, K, n_poses=8192, use_ap3p=False, use_double=False):
    pts3, pts2 = f32(pts3), f32(pts2)
    rv = np.zeros((n_poses, 3), np.float32)
    tv = np.zeros((n_poses, 3), np.float32)
    Kf = f32(K).reshape(9)
    lib().orc_solve_batch_p3p(_fp(pts3), _fp(pts2), _fp(rv), _fp(tv), _fp(Kf), pts3.shape[0], n_poses,
                              int(use_ap3p), int(use_double))
    return rv, tv


def meanshift(space, kernel_var, init_mean, use_external_init_mean=True, epsilon=1e-5, max_iters=100,
              max_init_trials=20, good_init_confidence=0.5):
    space = f32(space)
    N, dims = space.shape
    mean = f32(init_mean).copy()
    conf = C.c_float(0)
    iters = C.c_int(0)
    lib().orc_meanshift(_fp(space), C.c_float(kernel_var), _fp(mean), C.byref(conf), C.byref(iters),
                        int(use_external_init_mean), N, dims, C.c_float(epsilon), max_iters, max_init_trials,
                        C.c_float(good_init_confidence))
    return mean, conf.value, iters.value


def fit_robust_gaussian(space, mean, covar, trunc_sigma=3.0, covar_reg_lambda=1e-3, epsilon=1e-5, max_iters=100):
    space = f32(space)
    N, dims = space.shape
    mean = f32(mean).copy()
    covar = f32(covar).copy()
    dens = C.c_float(0)
    iters = C.c_int(0)
    rc = lib().orc_fit_robust_gaussian(_fp(space), _fp(mean), _fp(covar), C.c_float(trunc_sigma),
                                       C.c_float(covar_reg_lambda), C.byref(dens), C.byref(iters), N, dims,
                                       C.c_float(epsilon), max_iters)
    return rc, mean, covar, dens.value, iters.value


def estimate_pose_epipolar(flow, K, step=4):
    flow = f32(flow)
    h, w, _ = flow.shape
    R, t = np.zeros(9, np.float32), np.zeros(3, np.float32)
    ok = lib().orc_estimate_pose_epipolar(_fp(flow), _fp(f32(K).reshape(9)), w, h, step, _fp(R), _fp(t))
    return ok, R.reshape(3, 3), t


def estimate_depth_closed_form(flow, K, R, t, min_depth=1e-2, max_depth=1e10):
    flow = f32(flow)
    h, w, _ = flow.shape
    d = np.zeros((h, w), np.float32)
    lib().orc_estimate_depth_closed_form(_fp(flow), _fp(d), _fp(f32(K).reshape(9)), _fp(f32(R).reshape(9)),
                                         _fp(f32(t).reshape(3)), w, h, C.c_float(min_depth), C.c_float(max_depth))
    return d


def voldor(flows, fx, fy, cx, cy, basefocal=0.0, disparity=None, disparity_pconf=None, depth_priors=None,
           depth_prior_poses=None, depth_prior_pconfs=None, config=""):
    """Oracle twin of pyvoldor.voldor (slam_py/install/pyvoldor_vo.pyx:14-70)."""
    flows = f32(flows)
    N, h, w, _ = flows.shape
    N_dp = 0 if depth_priors is None else depth_priors.shape[0]
    poses = np.zeros((N, 6), np.float32)
    covar = np.zeros((N, 6, 6), np.float32)
    depth = np.zeros((h, w), np.float32)
    conf = np.zeros((h, w), np.float32)
    nreg = C.c_int(0)
    a = [None if x is None else f32(x) for x in (disparity, disparity_pconf, depth_priors, depth_prior_poses, depth_prior_pconfs)]
    rc = lib().orc_voldor(_fp(flows), _fp(a[0]), _fp(a[1]), _fp(a[2]), _fp(a[3]), _fp(a[4]),
                          C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), C.c_float(basefocal),
                          N, N_dp, w, h, config.encode(), C.byref(nreg), _fp(poses), _fp(covar), _fp(depth), _fp(conf))
    if rc != 0:
        raise RuntimeError(f"orc_voldor failed rc={rc}")
    n = nreg.value
    return {"n_registered": n, "poses": poses[:n], "poses_covar": covar[:n], "depth": depth, "depth_conf": conf}



def two_view_pose(flow0, K):
    """(R float32 3x3, t float32 3) of the oracle's 8-point LMedS bootstrap BEFORE the reference's cam.t = R * t
    (geometry.cpp:330): what ref_voldor() injects in place of cv::recoverPose (deviation D5)."""
    ok, R, _ = estimate_pose_epipolar(flow0, K)
    if not ok:
        raise RuntimeError("two-view bootstrap failed")
    t = np.zeros(3, np.float32)
    lib().orc_last_two_view_translation(_fp(t))
    return np.ascontiguousarray(R, np.float32), t


def ref_voldor(flows, fx, fy, cx, cy, basefocal=0.0, disparity=None, depth_priors=None, depth_prior_poses=None,
               depth_prior_pconfs=None, config="", rand_epoch=0, two_view=None):
    """The REFERENCE's own py_voldor_wrapper (voldor/py_export.cpp) executed on the CPU through oracle/_ref
    (ref_wrap_host.cpp: voldor/*.cpp compiled in place on minicv + the reference's kernel files on the launch emulation).
    Single-threaded.  Monocular windows get a two-view pose injected where the reference calls OpenCV: the oracle's 8-point one, or `two_view` = (R 3x3, t 3,
    the unit translation BEFORE the reference's cam.t = R t) when the caller brings its own (tests/golden/gen_golden_ensemble.py --five-point).  Raises if
    oracle/_ref is not built."""
    r = ref()
    if r is None or not hasattr(r, "ref_py_voldor_wrapper"):
        raise RuntimeError("oracle/_ref with the host pipeline is not built on this box")
    flows = f32(flows)
    N, h, w, _ = flows.shape
    if disparity is None and depth_priors is None:
        R, t = two_view if two_view is not None else two_view_pose(flows[0], np.array([fx, 0, cx, 0, fy, cy, 0, 0, 1], np.float32))
        R, t = np.asarray(R, np.float32), np.asarray(t, np.float32)
        D = C.POINTER(C.c_double)
        R64, t64 = R.astype(np.float64), t.astype(np.float64)
        r.ref_set_two_view_pose(R64.ctypes.data_as(D), t64.ctypes.data_as(D))
    a = [None if x is None else f32(x) for x in (disparity, depth_priors, depth_prior_poses, depth_prior_pconfs)]
    N_dp = 0 if a[1] is None else a[1].shape[0]
    poses = np.zeros((N, 6), np.float32)
    covar = np.zeros((N, 6, 6), np.float32)
    depth = np.zeros((h, w), np.float32)
    conf = np.zeros((h, w), np.float32)
    nreg = C.c_int(0)
    rc = r.ref_py_voldor_wrapper(_fp(flows), _fp(a[0]), None, _fp(a[1]), _fp(a[2]), _fp(a[3]), C.c_float(fx), C.c_float(fy), C.c_float(cx),
                                 C.c_float(cy), C.c_float(basefocal), N, N_dp, w, h, config.encode(), C.c_uint(rand_epoch), C.byref(nreg),
                                 _fp(poses), _fp(covar), _fp(depth), _fp(conf))
    if rc != 0:
        raise RuntimeError(f"ref_py_voldor_wrapper failed rc={rc}")
    n = nreg.value
    return {"n_registered": n, "poses": poses[:n], "poses_covar": covar[:n], "depth": depth, "depth_conf": conf}

# ---- frame alignment maps (orc_align.c; gpu-kernels/align_frame.cu) ----
def rot_with_rvec(p3, rvec):
    p3 = f32(p3); rvec = f32(rvec)
    out = np.zeros(3, np.float32); jw = np.zeros((3, 3), np.float32); jp = np.zeros((3, 3), np.float32)
    lib().orc_rot_with_rvec(_fp(p3), _fp(rvec), _fp(out), _fp(jw), _fp(jp))
    return out, jw, jp


class Align:
    """orc_align_init / orc_align_eval (align_frame_init_gpu / align_frame_eval_gpu of gpu_kernels.h:60-74)."""

    def __init__(self, images, depths, weights, K, vbf, crw):
        depths = f32(depths); weights = f32(weights)
        self.shape = depths.shape
        N, h, w = depths.shape
        images = None if images is None else f32(images)
        L = lib()
        L.orc_align_init.restype = C.c_void_p
        self._h = C.c_void_p(L.orc_align_init(_fp(images), _fp(depths), _fp(weights), _fp(f32(np.asarray(K, np.float32).reshape(9))),
                                               C.c_float(vbf), C.c_float(crw), N, w, h))

    def eval(self, ref_fid, tar_fid, params_ref, params_tar, want_jacobian=True, apply_weights=True):
        _, h, w = self.shape
        res = np.zeros((h, w), np.float32)
        jac = np.zeros((h, w, 9), np.float32) if want_jacobian else None
        lib().orc_align_eval(self._h, int(ref_fid), int(tar_fid), _fp(f32(params_ref)), _fp(f32(params_tar)), _fp(res), _fp(jac), int(bool(apply_weights)))
        return res, jac

    def __del__(self):
        try:
            lib().orc_align_free(self._h)
        except Exception:
            pass
