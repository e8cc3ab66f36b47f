"""The oracle against golden vectors produced by the REFERENCE's own code (tests/golden/gen_golden.py:
lambdatwist/*.h, gpu-kernels/residual_model.h, gpu-kernels/rodrigues.h compiled in place into
oracle/_ref).  Bit-exact where the oracle restates the arithmetic literally."""
import ctypes as C
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_residual_model_bit_exact(orc):
    g = np.load(os.path.join(G, "ref_residual.npz"))
    L = orc.lib()
    fl, lam, arf = g["flows"], g["lam"], g["arf"]
    rig = np.array([L.orc_fun_rigidness(*map(float, fl[i]), float(lam[i]), float(arf[i])) for i in range(len(fl))], np.float32)
    np.testing.assert_array_equal(rig.view(np.uint32), g["rigidness"].view(np.uint32))  # NaNs included
    d = g["d12"]
    drig = np.array([L.orc_fun_depth_rigidness(float(d[i, 0]), float(d[i, 1]), float(g["bf"][i]), 0.15, float(arf[i])) for i in range(len(d))], np.float32)
    np.testing.assert_array_equal(drig.view(np.uint32), g["depth_rigidness"].view(np.uint32))
    np.testing.assert_array_equal(np.array([L.orc_fun_fmag_c(float(v)) for v in g["fmag"]], np.float32), g["fmag_c"])
    np.testing.assert_array_equal(np.array([L.orc_fun_fmag_scale(float(v)) for v in g["fmag"]], np.float32), g["fmag_scale"])


@pytest.mark.parametrize("dbl", [0, 1])
def test_lambdatwist_bit_exact(orc, dbl):
    g = np.load(os.path.join(G, "ref_lambdatwist.npz"))
    fx, fy, cx, cy = map(float, g["K"])
    sfx = "d" if dbl else "f"
    n_ok = 0
    for i in range(len(g["y"])):
        ok, R, t = orc.lambdatwist_p4p(g["y"][i], g["x"][i], fx, fy, cx, cy, bool(dbl))
        assert ok == g["ok_" + sfx][i]
        if ok:
            n_ok += 1
            np.testing.assert_array_equal(R.reshape(9), g["R_" + sfx][i])
            np.testing.assert_array_equal(t, g["t_" + sfx][i])
    assert n_ok > 0.9 * len(g["y"])


def test_rodrigues_vs_reference_svd(orc):
    """The reference orthonormalises with an approximate fp32 SVD (svd3_cuda.h, 4 Jacobi sweeps), the
    oracle with an exact polar factor: agreement to the SVD's own accuracy."""
    g = np.load(os.path.join(G, "ref_rodrigues.npz"))
    err = np.array([np.abs(orc.rodrigues(g["R"][i]) - g["rvec"][i]).max() for i in range(len(g["R"]))])
    assert np.percentile(err, 99) < 2e-5 and err.max() < 2e-3, (np.percentile(err, [50, 99]), err.max())
    aa = np.array([orc.rotmat_to_angle_axis(g["R"][i]) for i in range(len(g["R"]))])
    assert np.abs(aa - g["angle_axis_no_svd"]).max() < 1e-6  # Ceres formula alone: same libm


def test_reference_svd_restatement_is_bit_identical_to_the_reference(orc):
    """voldor_amd/csrc/vk_ref_svd.h (--reference_svd 1 of the product, orc_set_reference_svd here) restates svd3_cuda.h:36-1044 and
    rodrigues.h:82-108.  Against the reference's own rodrigues() compiled in place (golden: tests/golden/gen_golden.py): the projected
    matrix U V^T in every bit on 3234 inputs -- rotations, noisy rotations, and matrices far from SO(3) that walk every branch of the
    column sort and the Givens stage --, the rotation vectors in every bit under glibc and under strict math."""
    g = np.load(os.path.join(G, "ref_rodrigues.npz"))
    R = g["R_all"]
    Q = np.array([orc.reference_project_rotation(R[i]).reshape(9) for i in range(len(R))], np.float32)
    assert np.array_equal(Q.view(np.uint32), g["proj_all"].view(np.uint32)), np.mean(np.any(Q != g["proj_all"], axis=1))
    orc.set_reference_svd(True)
    try:
        rv = np.array([orc.rodrigues(R[i]) for i in range(len(R))], np.float32)
        assert np.array_equal(rv.view(np.uint32), g["rvec_all"].view(np.uint32))
        orc.lib().orc_set_strict_math(1)
        rvs = np.array([orc.rodrigues(R[i]) for i in range(len(R))], np.float32)
        assert np.array_equal(rvs.view(np.uint32), g["rvec_all_strict"].view(np.uint32))
    finally:
        orc.lib().orc_set_strict_math(0)
        orc.set_reference_svd(False)
    # and the default (D8) stays what it was: the exact polar factor
    assert np.abs(orc.rodrigues(R[5]) - g["rvec_all"][5]).max() < 2e-3


def test_live_reference_build_if_present(orc):
    """Where /root/reference exists (authoring container) the freshly built oracle/_ref must agree too."""
    ref = orc.ref()
    if ref is None:
        pytest.skip("oracle/_ref not built (no /root/reference on this box)")
    rng = np.random.default_rng(5)
    L = orc.lib()
    for _ in range(500):
        a = [float(v) for v in rng.normal(0, 6, 4).astype(np.float32)]
        assert L.orc_fun_rigidness(*a, 0.15, 1.0) == ref.ref_fun_rigidness(*a, 0.15, 1.0)
