"""SURVEY.md 8(f)-2 on the GPU: align_frame_init_gpu / align_frame_eval_gpu through the C-ABI against the oracle
(oracle/orc_align.c) on the same key-frames."""
import numpy as np
import pytest

from align_scene import keyframes

pytestmark = pytest.mark.gpu


def _compare(ro, jo, rg, jg):
    fo, fg = np.isfinite(ro), np.isfinite(rg)
    assert np.mean(fo != fg) < 2e-3  # a projection within rounding of the image border may fall on the other side
    m = fo & fg
    assert m.mean() > 0.5
    assert np.abs(ro[m] - rg[m]).max() < 5e-4 * max(1.0, np.abs(ro[m]).max())  # sqrt(log(1 + r)) of small r: libm vs device log/exp/sincos
    if jo is not None:
        scale = np.abs(jo[m]).max(axis=0) + 1e-12  # per parameter
        # the sqrt-Cauchy factor 0.5 / sqrt(log(1 + r)) amplifies rounding differences of r where r -> 0 (r is a
        # difference of interpolated quantities there): tight where the residual is not tiny, looser elsewhere
        big = m & (ro > 0.02)
        assert big.sum() > 100
        assert (np.abs(jo[big] - jg[big]).max(axis=0) / scale).max() < 2e-3
        mid = m & (ro > 2e-3)
        assert (np.abs(jo[mid] - jg[mid]).max(axis=0) / scale).max() < 3e-2
        # below that, r straddles the FLT_EPSILON switch of the loss (align_frame.cu:399: raw vs sqrt-Cauchy value, a factor
        # ~1e3 in the Jacobian) differently on the two sides for a few pixels; they must stay rare
        bad = (np.abs(jo[m] - jg[m]) / scale).max(axis=1) > 3e-2
        assert bad.mean() < 2e-2
        assert np.all(jg[~fg] == 0)  # no Jacobian where the residual is undefined (align_frame.cu:432)


@pytest.mark.parametrize("photo", [True, False])
def test_align_maps_match_oracle(orc, photo):
    from voldor_amd import kernels
    kf = keyframes(w=192, h=136, n=3, seed=1)
    images = kf["images"] if photo else None
    A = orc.Align(images, kf["depths"], kf["weights"], kf["K"], kf["vbf"], kf["crw"])
    rc, shape = kernels.align_frame_init_gpu(images, kf["depths"], kf["weights"], kf["K"], kf["vbf"], kf["crw"])
    assert rc == 0
    rng = np.random.default_rng(2)
    for (ref, tar) in ((0, 1), (2, 1), (1, 0)):
        pr, pt = kf["params"][ref].copy(), kf["params"][tar].copy()
        pr[:6] += rng.normal(0, 0.01, 6).astype(np.float32)
        pr[6:] = rng.normal(0, 0.05, 3)
        pt[6:] = rng.normal(0, 0.05, 3)
        for aw in (True, False):
            ro, jo = A.eval(ref, tar, pr, pt, True, aw)
            rc, rg, jg = kernels.align_frame_eval_gpu(shape, ref, tar, pr, pt, True, aw)
            assert rc == 0
            _compare(ro, jo, rg, jg)
            if not photo:
                assert np.all(jg[..., 7:] == 0)
    # a large rotation exercises the theta^(3/2) terms of d/d(rvec)
    pr = kf["params"][0].copy(); pr[:3] += [0.2, -0.1, 0.15]
    ro, jo = A.eval(0, 1, pr, kf["params"][1])
    rc, rg, jg = kernels.align_frame_eval_gpu(shape, 0, 1, pr, kf["params"][1])
    _compare(ro, jo, rg, jg)


def test_align_null_protocol_and_errors(orc):
    from voldor_amd import kernels
    kf = keyframes(w=96, h=64, n=2, seed=4)
    rc, shape = kernels.align_frame_init_gpu(kf["images"], kf["depths"], kf["weights"], kf["K"], kf["vbf"], kf["crw"])
    assert rc == 0
    rc, r1, j1 = kernels.align_frame_eval_gpu(shape, 0, 1, kf["params"][0], kf["params"][1])
    rc2, r2, _ = kernels.align_frame_eval_gpu(shape, 0, 1, None, None, want_jacobian=False)  # NULL params: keep the previous ones (:425-428)
    assert rc == 0 and rc2 == 0
    np.testing.assert_allclose(r1, r2, rtol=0, atol=1e-4, equal_nan=True)  # the residual-only kernel is a separate instantiation
    assert kernels.align_frame_eval_gpu(shape, 0, 5, kf["params"][0], kf["params"][1])[0] != 0  # frame id out of range
