"""voldor_amd/csrc/vk_ref_cv.h: cv::Rodrigues(matrix -> vector) as the reference's host code applies it to a camera's FLOAT rotation matrix
(Camera::rvec(), utils.h:49-53): the start of every mean shift (geometry.cpp:184) and the poses a window returns (pose6()).  One restatement for
the OpenCV stand-in of the emulated reference, the oracle and the strict kernels (round 4).  Here: its properties on the host build (CPU), and
-- marked gpu -- the gfx950 build against the host build, bit for bit."""
import numpy as np
import pytest

import hooks


def _rotations(n, seed, max_angle):
    rng = np.random.default_rng(seed)
    ax = rng.normal(size=(n, 3)); ax /= np.linalg.norm(ax, axis=1, keepdims=True)
    ang = rng.uniform(0, max_angle, n)
    rv = (ax * ang[:, None]).astype(np.float32)
    R = np.zeros((n, 9), np.float32)
    L = hooks.lib()
    for i in range(n):
        L.vk_host_rvec_to_rotmat(hooks._p(rv[i]), hooks._p(R[i]))
    return rv, R


@pytest.mark.parametrize("strict", [0, 1])
def test_round_trip_is_the_identity_up_to_the_float_matrix(strict):
    rv, R = _rotations(4000, 1, 3.0)
    back = hooks.cv_rvec_of_R(R, strict, device=False)
    # a float matrix holds a rotation to ~6e-8 per entry: the vector comes back to that accuracy (relative to the angle's conditioning)
    err = np.abs(back.astype(np.float64) - rv).max(axis=1)
    ang = np.linalg.norm(rv, axis=1)
    assert (err < 4e-7 / np.maximum(np.sinc(ang / np.pi), 0.05)).all(), err.max()
    # and it is NOT the identity in bits: that ulp is what separated oracle and reference on a non-converging mean shift
    assert (back.view(np.uint32) != rv.view(np.uint32)).any()


def test_special_matrices():
    eye = np.eye(3, dtype=np.float32).reshape(1, 9)
    assert not hooks.cv_rvec_of_R(eye, 1, device=False).any()
    # rotation by pi about z, about an oblique axis: the theta ~ pi branch (axis from the diagonal)
    for axis in ([0, 0, 1.0], [1 / 3 ** 0.5] * 3, [0.6, 0, 0.8]):
        a = np.asarray(axis, np.float64)
        R = (2 * np.outer(a, a) - np.eye(3)).astype(np.float32).reshape(1, 9)
        for strict in (0, 1):
            r = hooks.cv_rvec_of_R(R, strict, device=False)[0].astype(np.float64)
            assert abs(np.linalg.norm(r) - np.pi) < 1e-6 and (np.allclose(r / np.pi, a, atol=1e-6) or np.allclose(r / np.pi, -a, atol=1e-6)), (axis, r)
    # a matrix that is only nearly a rotation (what a float product of rotations looks like) is orthonormalised first
    rv, R = _rotations(200, 2, 1.0)
    noisy = (R + np.random.default_rng(3).normal(0, 2e-7, R.shape)).astype(np.float32)
    d = np.abs(hooks.cv_rvec_of_R(noisy, 1, device=False) - hooks.cv_rvec_of_R(R, 1, device=False)).max()
    assert d < 2e-6, d


def test_strict_and_library_functions_agree_to_an_ulp():
    # strict mode evaluates sin / cos / acos with vk_strict_math.h (acos through atan2); the C library's agree to the last bit or two of the FLOAT result
    rv, R = _rotations(4000, 4, 3.1)
    a = hooks.cv_rvec_of_R(R, 0, device=False); b = hooks.cv_rvec_of_R(R, 1, device=False)
    ulp = np.abs(a.view(np.int32).astype(np.int64) - b.view(np.int32).astype(np.int64))
    assert ulp.max() <= 2 and (ulp == 0).mean() > 0.99, (ulp.max(), (ulp == 0).mean())


@pytest.mark.gpu
def test_device_build_gives_the_host_bits():
    rv, R = _rotations(20000, 5, 3.1)
    extra = np.concatenate([np.eye(3, dtype=np.float32).reshape(1, 9), (2 * np.outer([0, 0, 1.0], [0, 0, 1.0]) - np.eye(3)).astype(np.float32).reshape(1, 9),
                            (R[:100] + np.random.default_rng(6).normal(0, 1e-6, (100, 9))).astype(np.float32)])
    R = np.concatenate([R, extra])
    h = hooks.cv_rvec_of_R(R, 1, device=False); d = hooks.cv_rvec_of_R(R, 1, device=True)
    assert np.array_equal(h.view(np.uint32), d.view(np.uint32)), int((h.view(np.uint32) != d.view(np.uint32)).sum())
