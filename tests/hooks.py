"""ctypes handle of the TEST library voldor_amd/lib/libvoldor_hip_test.so (tests/cxx/vk_testhooks.hip): host builds of the
product's per-lane math and device-vs-host probes.  Tests only; the product never loads it."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "voldor_amd", "lib", "libvoldor_hip_test.so")
_lib = None


def lib():
    global _lib
    if _lib is None:
        from voldor_amd import build
        build.build()
        assert build.build_test_lib() is not None
        _lib = C.CDLL(PATH)
        _lib.vk_host_u01.restype = C.c_float
        _lib.vk_host_rng.restype = C.c_uint
        _lib.vkt_strict_rigidness.restype = C.c_float
        _lib.vkt_strict_rigidness.argtypes = [C.c_float] * 6
        _lib.vkt_strict_depth_rigidness.restype = C.c_float
        _lib.vkt_strict_depth_rigidness.argtypes = [C.c_float] * 5
    return _lib


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t))


def probe(op, a, b, device):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros(a.size, np.float64)
    L = lib()
    if device:
        rc = L.vkt_probe_device(int(op), _p(a), _p(b), _p(out, C.c_double), a.size)
        assert rc == 0, rc
    else:
        L.vkt_probe_host(int(op), _p(a), _p(b), _p(out, C.c_double), a.size)
    return out


def p4p(y8, x12, K4, use_double, device):
    y8 = np.ascontiguousarray(y8, np.float32); x12 = np.ascontiguousarray(x12, np.float32)
    n = y8.shape[0]
    L = lib()
    nd = L.vkt_p4p_dbg_len()
    ok = np.zeros(n, np.int32); R = np.zeros((n, 9), np.float32); t = np.zeros((n, 3), np.float32); dbg = np.zeros((n, nd), np.float64)
    fx, fy, cx, cy = [C.c_float(float(v)) for v in K4]
    fn = L.vkt_p4p_device if device else L.vkt_p4p_host
    rc = fn(_p(y8), _p(x12), n, fx, fy, cx, cy, int(use_double), _p(ok, C.c_int), _p(R), _p(t), _p(dbg, C.c_double))
    assert not device or rc == 0, rc
    return ok, R, t, dbg


def rodrigues(R9, strict, device):
    R9 = np.ascontiguousarray(R9, np.float32).reshape(-1, 9)
    rv = np.zeros((R9.shape[0], 3), np.float32)
    L = lib()
    if device:
        assert L.vkt_rodrigues_device(_p(R9), _p(rv), R9.shape[0], int(strict)) == 0
    else:
        L.vkt_rodrigues_host(_p(R9), _p(rv), R9.shape[0], int(strict))
    return rv


def cv_rvec_of_R(R9, strict, device):
    """cv::Rodrigues(matrix -> vector) of float matrices (voldor_amd/csrc/vk_ref_cv.h): host build, or the gfx950 build (strict only)."""
    R9 = np.ascontiguousarray(R9, np.float32).reshape(-1, 9)
    rv = np.zeros((R9.shape[0], 3), np.float32)
    L = lib()
    if device:
        assert strict and L.vkt_cv_rvec_of_R_device(_p(R9), _p(rv), R9.shape[0]) == 0
    else:
        L.vkt_cv_rvec_of_R_host(_p(R9), _p(rv), R9.shape[0], int(strict))
    return rv


# ---- verification entry points of the PRODUCT library that are not part of its C-ABI (voldor_amd/csrc/vk_debug.h) ----
def debug_switch(name: str, value: int) -> int:
    """vk_debug_switch: returns the previous value; raises for an unknown name / value."""
    from voldor_amd import capi
    old = capi.lib().vk_debug_switch(name.encode(), int(value))
    if old < 0:
        raise ValueError(f"vk_debug_switch({name!r}, {value}) rejected")
    return old


def set_local_serial(on): debug_switch("local_serial", int(bool(on)))
def set_cost_rand_plain(on): debug_switch("cost_rand_plain", int(bool(on)))
def set_fb_segment(steps): debug_switch("fb_segment", int(steps))
def set_global_split(on): debug_switch("global_split", int(bool(on)))
def set_refit_partition(on): debug_switch("refit_partition", int(bool(on)))
def set_split_trials(on): debug_switch("split_trials", int(bool(on)))
def set_strict_plain(on): debug_switch("strict_plain", int(bool(on)))
def set_strict_pose_coop(on): debug_switch("strict_pose_coop", int(bool(on)))
def set_strict_coop_max_polls(n): debug_switch("strict_coop_max_polls", int(n))


def debug_counter(name):
    """vk_debug_counter (vk_debug.h): read and clear"""
    from voldor_amd import capi
    import ctypes
    f = capi.lib().vk_debug_counter
    f.argtypes = [ctypes.c_char_p]; f.restype = ctypes.c_int
    v = f(name.encode())
    if v < 0:
        raise ValueError(f"vk_debug_counter({name!r}) failed")
    return v


def pose_mode_pool(rvecs, tvecs, init_pose6, use_external_init_mean=True, refit=False, kernel_var=0.2, rvec_scale=1.0, ms_epsilon=1e-5,
                   ms_max_iters=100, ms_max_init_trials=20, ms_good_init_confidence=0.5, rg_trunc_sigma=3.0, rg_covar_reg_lambda=1e-3,
                   rg_epsilon=1e-5, rg_max_iters=100, rg_pose_scaling=100.0):
    """vk_pose_mode_pool (vk_debug.h): the window pipeline's own mode kernel on a pool of hypotheses (strict-math kernel when the process-wide
    strict mode is on).  Returns dict(pose6, covar [6,6], density, sample_count, ms_iters, gu_iters, success)."""
    from voldor_amd import capi
    from voldor_amd.capi import f32, fp
    rv, tv = f32(rvecs).reshape(-1, 3), f32(tvecs).reshape(-1, 3)
    pose = f32(init_pose6).copy().reshape(6)
    cov = np.zeros((6, 6), np.float32)
    dens = C.c_float(0)
    cnt, msi, gui, ok = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    capi.check(capi.lib().vk_pose_mode_pool(fp(rv), fp(tv), rv.shape[0], int(use_external_init_mean), fp(pose), C.c_float(kernel_var), C.c_float(rvec_scale),
                                            C.c_float(ms_epsilon), int(ms_max_iters), int(ms_max_init_trials), C.c_float(ms_good_init_confidence), int(refit),
                                            C.c_float(rg_trunc_sigma), C.c_float(rg_covar_reg_lambda), C.c_float(rg_epsilon), int(rg_max_iters),
                                            C.c_float(rg_pose_scaling), fp(cov), C.byref(dens), C.byref(cnt), C.byref(msi), C.byref(gui), C.byref(ok)),
               "vk_pose_mode_pool")
    return dict(pose6=pose, covar=cov, density=dens.value, sample_count=cnt.value, ms_iters=msi.value, gu_iters=gui.value, success=ok.value)


def filter_pair(dx1, dy1, ox, oy, lam, arf):
    """(-vsm_logf(strict::rigidness), the fp32 filter's value) for the same float inputs, on the device (vk_device.hpp filt_neglog)."""
    a = [np.ascontiguousarray(v, np.float32) for v in (dx1, dy1, ox, oy)]
    n = a[0].size
    s = np.zeros(n, np.float32); f = np.zeros(n, np.float32)
    rc = lib().vkt_filter_pair_device(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), n, C.c_float(lam), C.c_float(arf), _p(s), _p(f))
    assert rc == 0, rc
    return s, f


def filter_pair_depth(d1, d2, basefocal, omega, arf):
    """(-vsm_logf(strict::depth_rigidness), the fp32 filter's value) for the same float inputs, on the device."""
    a = [np.ascontiguousarray(v, np.float32) for v in (d1, d2)]
    n = a[0].size
    s = np.zeros(n, np.float32); f = np.zeros(n, np.float32)
    rc = lib().vkt_filter_pair_depth_device(_p(a[0]), _p(a[1]), n, C.c_float(basefocal), C.c_float(omega), C.c_float(arf), _p(s), _p(f))
    assert rc == 0, rc
    return s, f
