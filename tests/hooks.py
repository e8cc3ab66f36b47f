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
