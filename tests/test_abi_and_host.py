"""CPU tier for the product library: it loads, exports every declared symbol (C facade and the mangled
C++ names the reference host code links against), fails loudly without a GPU, and its host-compiled
solver math (the same source the GPU lanes run) matches the oracle and the reference golden vectors."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")


def test_library_exports_every_declared_symbol():
    from voldor_amd import capi
    lib = capi.lib()
    hdr = open(os.path.join(ROOT, "include", "voldor_hip.h")).read()
    declared = set(re.findall(r"\b(vk_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(capi.C_SYMBOLS), declared ^ set(capi.C_SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    nm = subprocess.check_output(["nm", "-D", "--defined-only", capi.LIB_PATH]).decode()
    exported = {ln.split()[-1] for ln in nm.splitlines() if " T " in ln}
    demangled = subprocess.check_output(["c++filt"], input="\n".join(sorted(exported)).encode()).decode().splitlines()
    sigs = [d for d in demangled if "(" in d]
    for fn in ("meanshift_gpu(float*, float, float*, float*, int*, bool, int, int, float, int, int, float)",
               "fit_robust_gaussian(float*, float*, float*, float, float, float*, int*, int, int, float, int)",
               "collect_p3p_instances(float**, float**, float*, float*, float**, float**, float*, float*, int, int, int, int, float, float, float, float, int)",
               "solve_batch_p3p_ap3p_gpu(float*, float*, float*, float*, float*, int, int)",
               "solve_batch_p3p_lambdatwist_gpu(float*, float*, float*, float*, float*, int, int)",
               "align_frame_init_gpu(float**, float**, float**, float*, float, float, int, int, int)",           # gpu_kernels.h:60-66
               "align_frame_eval_gpu(int, int, float const*, float const*, float*, float*, bool)",               # gpu_kernels.h:68-74
               "py_voldor_wrapper(float const*, float const*, float const*, float const*, float const*, float const*, float, float, float, float, float, int, int, int, int, char const*, int&, float*, float*, float*, float*)"):
        assert fn in sigs, fn
    assert any(s.startswith("optimize_depth_gpu(float**, float**, float**, float**, float**, float**, float**, float*, float*, float*, float**, float**, float**, float**, float, int, int, int, int, float, int, int, int, float, float, float, float, bool, float, float, float, bool)") for s in sigs)


def test_headers_mirror_reference_boundary():
    g = open(os.path.join(ROOT, "include", "gpu_kernels.h")).read()
    for name in ("meanshift_gpu", "fit_robust_gaussian", "collect_p3p_instances", "solve_batch_p3p_ap3p_gpu",
                 "solve_batch_p3p_lambdatwist_gpu", "optimize_depth_gpu", "align_frame_init_gpu", "align_frame_eval_gpu"):
        assert re.search(r"DLL_EXPORT int " + name + r"\(", g), name
    assert "float epsilon = 1e-5f, int max_iters = 100" in g and "float good_init_confidence = 0.5f" in g  # default args
    assert "int& n_registered" in open(os.path.join(ROOT, "include", "py_export.h")).read()


def test_no_cpu_fallback_without_gpu():
    from voldor_amd import capi, pyvoldor
    if capi.lib().vk_device_count() > 0:
        pytest.skip("a GPU is present")
    flows = np.zeros((2, 8, 8, 2), np.float32)
    with pytest.raises(capi.VoldorHipError):
        pyvoldor.voldor(flows, 4.0, 4.0, 4.0, 4.0, config="--silent")


def test_pyvoldor_argument_contract():
    from voldor_amd import pyvoldor
    with pytest.raises(TypeError):
        pyvoldor.voldor(None, 1, 1, 1, 1)
    with pytest.raises(ValueError):
        pyvoldor.voldor(np.zeros((2, 4, 4, 2), np.float64), 1, 1, 1, 1)
    with pytest.raises(ValueError):
        pyvoldor.voldor(np.zeros((4, 4, 2), np.float32), 1, 1, 1, 1)
    with pytest.raises(ValueError):
        pyvoldor.voldor(np.zeros((2, 4, 4, 2), np.float32), 1, 1, 1, 1, disparity=np.zeros((1, 4, 4), np.float32))


def _host_p4p(lib, y, x, fx, fy, cx, cy, dbl):
    from voldor_amd import capi
    R = np.zeros(9, np.float32); t = np.zeros(3, np.float32)
    ok = lib.vk_host_lambdatwist_p4p(capi.fp(capi.f32(y).reshape(8)), capi.fp(capi.f32(x).reshape(12)), C.c_float(fx), C.c_float(fy),
                                     C.c_float(cx), C.c_float(cy), int(dbl), capi.fp(R), capi.fp(t))
    return ok, R, t


@pytest.mark.parametrize("dbl", [0, 1])
def test_device_solver_source_matches_reference_golden(dbl):
    """vk_p3p.hpp compiled for the host reproduces the REFERENCE's own lambdatwist bit for bit."""
    import hooks
    lib = hooks.lib()
    g = np.load(os.path.join(G, "ref_lambdatwist.npz"))
    fx, fy, cx, cy = map(float, g["K"])
    sfx = "d" if dbl else "f"
    for i in range(len(g["y"])):
        ok, R, t = _host_p4p(lib, g["y"][i], g["x"][i], fx, fy, cx, cy, dbl)
        assert ok == g["ok_" + sfx][i]
        if ok:
            np.testing.assert_array_equal(R, g["R_" + sfx][i])
            np.testing.assert_array_equal(t, g["t_" + sfx][i])


def test_host_ap3p_rodrigues_rng_match_oracle(orc):
    import hooks
    from voldor_amd import capi, synth
    lib = hooks.lib()
    lib.vk_host_u01.restype = C.c_float
    lib.vk_host_rng.restype = C.c_uint
    L = orc.lib()
    rng = np.random.default_rng(9)
    for i in range(2000):
        assert lib.vk_host_rng(233, i, i * 7 + 1) == L.orc_rng(233, i, i * 7 + 1)
    assert lib.vk_host_u01(C.c_uint(123456789)) == L.orc_u01(123456789)
    fx = fy = 350.0; cx, cy = 300.0, 220.0
    nfin = 0
    for i in range(400):
        X = rng.uniform([-3, -2, 3], [3, 2, 15], (4, 3)).astype(np.float32)
        rv, t = rng.normal(0, 0.1, 3), rng.normal(0, 0.5, 3)
        Xc = X @ synth.rodrigues(rv).T + t
        y = (np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy], -1) + rng.normal(0, 0.3, (4, 2))).astype(np.float32)
        oko, Ro, to = orc.ap3p_p4p(y, X, fx, fy, cx, cy)
        R = np.zeros(9, np.float32); tt = np.zeros(3, np.float32)
        okh = lib.vk_host_ap3p_p4p(capi.fp(y.reshape(8)), capi.fp(X.reshape(12)), C.c_float(fx), C.c_float(fy), C.c_float(cx), C.c_float(cy), capi.fp(R), capi.fp(tt))
        assert oko == okh
        if oko and np.isfinite(to).all() and np.isfinite(tt).all():
            nfin += 1
            assert np.abs(Ro.reshape(9) - R).max() < 1e-3 and np.abs(to - tt).max() < 1e-2
        Rm = (synth.rodrigues(rv) + rng.normal(0, 1e-3, (3, 3))).astype(np.float32)
        r1 = np.zeros(3, np.float32)
        lib.vk_host_rodrigues(capi.fp(Rm.reshape(9)), capi.fp(r1))
        assert np.abs(r1 - orc.rodrigues(Rm)).max() < 1e-6
    assert nfin > 300


def test_bootstrap_host_path_matches_oracle(orc, small_scene):
    from voldor_amd import kernels
    from conftest import K9
    K = K9(*small_scene["K"])
    for f in range(2):
        ok_o, Ro, to = orc.estimate_pose_epipolar(small_scene["flows"][f], K)
        ok_h, Rh, th = kernels.estimate_pose_epipolar(small_scene["flows"][f], K)
        assert ok_o and ok_h
        np.testing.assert_array_equal(Ro, Rh)
        np.testing.assert_array_equal(to, th)
