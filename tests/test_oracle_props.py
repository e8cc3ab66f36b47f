"""Property tests of the oracle's restatement: invariants the reference formulas imply."""
import numpy as np
import pytest


def test_rigidness_half_at_lambda_flow(orc):
    # err = lambda*|flow|  ->  p == mu  ->  rigidness == 0.5 exactly (residual_model.h:39-41)
    L = orc.lib()
    for mag in (0.5, 3.0, 10.0, 40.0, 250.0):
        lam = 0.15
        r = L.orc_fun_rigidness(mag + lam * mag, 0.0, mag, 0.0, lam, 1.0)
        assert abs(r - 0.5) < 1e-5
    assert L.orc_fun_rigidness(5.0, 1.0, 5.0, 1.0, 0.15, 1.0) > 0.99  # zero error (the log-logistic pdf is finite at 0 for c<1)
    errs = [L.orc_fun_rigidness(8.0 + e, 0.0, 8.0, 0.0, 0.15, 1.0) for e in (0.0, 0.2, 0.5, 1.0, 2.0, 5.0)]
    assert all(a >= b for a, b in zip(errs, errs[1:]))  # monotone in the end-point error


def test_rng_support(orc):
    L = orc.lib()
    u = np.array([L.orc_u01(L.orc_rng(233, i, 7)) for i in range(20000)])
    assert u.min() > 0 and u.max() <= 1.0  # (0,1] like curand_uniform
    assert abs(u.mean() - 0.5) < 0.01 and abs(np.mean(u < 0.25) - 0.25) < 0.015
    assert L.orc_u01(0xFFFFFFFF) == 1.0


def test_fb_smooth_constant_map_and_monotonicity(orc):
    m = np.full((1, 40, 50), 0.7, np.float32)
    s = orc.fb_smooth(m)
    assert np.ptp(s[0, 15:-15, 15:-15]) < 1e-4 and s[0, 20, 25] > 0.7  # deep interior is flat; agreeing neighbours reinforce
    rng = np.random.default_rng(0)
    a = rng.uniform(0.05, 0.95, (2, 33, 47)).astype(np.float32)
    sa = orc.fb_smooth(a)
    assert (sa > 0).all() and (sa < 1).all()
    b = a.copy(); b[0, 10:20, 10:30] = np.minimum(b[0, 10:20, 10:30] + 0.04, 0.99)
    sb = orc.fb_smooth(b)
    assert (sb[0] >= sa[0] - 1e-6).all() and (sb[0, 10:20, 10:30] > sa[0, 10:20, 10:30]).all()  # monotone in the emissions
    np.testing.assert_array_equal(sb[1], sa[1])  # maps are independent


def test_p3p_reprojection_and_known_pose(orc):
    from voldor_amd import synth
    rng = np.random.default_rng(1)
    fx = fy = 400.0; cx, cy = 320.0, 240.0
    for use_double in (False, True):
        good = 0
        for _ in range(200):
            X = rng.uniform([-3, -2, 3], [3, 2, 15], (4, 3)).astype(np.float32)
            rv, t = rng.normal(0, 0.1, 3), rng.normal(0, 0.5, 3)
            Xc = X @ synth.rodrigues(rv).T + t
            y = np.stack([fx * Xc[:, 0] / Xc[:, 2] + cx, fy * Xc[:, 1] / Xc[:, 2] + cy], -1).astype(np.float32)
            ok, R, tt = orc.lambdatwist_p4p(y, X, fx, fy, cx, cy, use_double)
            assert ok
            P = X @ R.T + tt
            rep = np.stack([fx * P[:, 0] / P[:, 2] + cx, fy * P[:, 1] / P[:, 2] + cy], -1)
            assert np.abs(rep - y)[:3].max() < (5e-2 if not use_double else 1e-3)  # the three points P3P solves for
            good += np.abs(R - synth.rodrigues(rv)).max() < 5e-3 and np.abs(tt - t).max() < 5e-2
        assert good >= 190  # fp32 loses the true root on a few ill-conditioned triples
    ok, R, tt = orc.ap3p_p4p(y, X, fx, fy, cx, cy)
    assert ok and np.abs(tt - t).max() < 0.1


def test_rodrigues_round_trip(orc):
    from voldor_amd import synth
    rng = np.random.default_rng(2)
    for scale in (1e-6, 1e-3, 0.3, 2.5, np.pi - 1e-3):
        v = rng.normal(size=3); v = v / np.linalg.norm(v) * scale
        R = synth.rodrigues(v).astype(np.float32)
        assert np.abs(orc.rodrigues(R) - v).max() < 2e-4 * max(1.0, scale) / (1.0 if scale < 3 else 0.05)
        assert np.abs(orc.rvec_to_rotmat(v.astype(np.float32)) - R).max() < 1e-6
    np.testing.assert_array_equal(orc.rvec_to_rotmat(np.zeros(3, np.float32)), np.eye(3, dtype=np.float32))


def test_meanshift_planted_mode(orc):
    rng = np.random.default_rng(3)
    mode = np.array([0.2, -0.1, 0.05, 0.3, 0.0, 1.0])
    pts = np.concatenate([mode + rng.normal(0, 0.05, (4000, 6)), rng.uniform(-3, 3, (4000, 6))]).astype(np.float32)
    for ext in (True, False):
        m, conf, it = orc.meanshift(pts, 0.1, np.zeros(6, np.float32), ext)
        assert np.abs(m - mode).max() < 0.02 and 0.2 < conf < 0.8 and 1 <= it <= 100


def test_robust_gaussian_reports_failure_on_rank_deficient_cloud(orc):
    rng = np.random.default_rng(4)
    pts = np.zeros((2000, 6), np.float32); pts[:, 0] = rng.normal(size=2000)
    assert orc.fit_robust_gaussian(pts, np.zeros(6, np.float32), np.zeros((6, 6), np.float32))[0] != 0
    cov = np.diag([1.0, 2.0, 0.5, 1.5, 0.7, 1.2])
    good = rng.multivariate_normal(np.zeros(6), cov, 8000).astype(np.float32)
    rc, mean, c, dens, it = orc.fit_robust_gaussian(good, np.zeros(6, np.float32), (np.eye(6) * 4).astype(np.float32), max_iters=3)
    assert rc == 0 and np.abs(mean).max() < 0.1 and np.all(np.diag(c) > 0) and 0.5 < dens <= 1.0


def test_gblur_preserves_constants_and_mass_at_borders(orc):
    rc, o = orc.gblur(np.full((1, 20, 30), 3.0, np.float32), 2.0)
    assert rc == 0 and np.allclose(o, 3.0, atol=1e-5)  # border re-normalisation (gblur.cu:19-40)
    assert orc.gblur(np.zeros((1, 8, 8), np.float32), 100.0)[0] != 0


def test_window_recovers_ground_truth(orc):
    from voldor_amd import synth
    sc = synth.make_scene(w=200, h=150, n_flows=4, fx=100, fy=100, cx=100, cy=75, seed=21)
    fx, fy, cx, cy = sc["K"]
    out = orc.voldor(sc["flows"], fx, fy, cx, cy, config="--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 5")
    assert out["n_registered"] == 4
    gt = sc["poses_gt"].copy(); gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
    rot, tr = synth.pose_errors(out["poses"], gt)
    assert rot.max() < 4e-3 and tr.max() < 0.08
    # config grammar: unknown key / missing value are errors (config.h:245-248, :101-108)
    with pytest.raises(RuntimeError):
        orc.voldor(sc["flows"], fx, fy, cx, cy, config="--bogus 1")
    with pytest.raises(RuntimeError):
        orc.voldor(sc["flows"], fx, fy, cx, cy, config="--max_iters")
