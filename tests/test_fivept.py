"""The five-point minimal solver (voldor_amd/csrc/vk_fivept.hpp): the solver behind the reference's
cv::findEssentialMat(pts1, pts2, K, LMEDS, 0.999, 1.0) (voldor/geometry.cpp:316-326).  OpenCV is not in the reference tree and not on this
box, so these are PROPERTY tests of a restatement of the published algorithm (Nister 2004), on the host build of the very source the
bootstrap kernel compiles (no GPU needed):
  * every returned matrix satisfies the five epipolar constraints and the essential-matrix constraints det E = 0,
    2 E E^T E - trace(E E^T) E = 0 (two equal singular values, one zero);
  * the planted essential matrix is among the solutions (300 random two-view geometries, incl. pure forward motion and small baselines);
  * the LMedS bootstrap on top of it (192 samples x up to 10 models, median squared Sampson distance, cheirality vote, cam.t = R t)
    recovers a planted pose from a flow field with 40 % gross outliers, and agrees with the 8-point bootstrap within a tolerance that
    the 8-point estimator itself shows between two scenes' worth of sampling."""
import numpy as np
import pytest

from conftest import K9


def _rodrigues(rv):
    th = np.linalg.norm(rv)
    if th < 1e-12:
        return np.eye(3)
    k = rv / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx


def _geometry(rng, forward=False, baseline=1.0):
    R = _rodrigues(rng.normal(0, 0.05, 3))
    t = np.array([0.0, 0.0, 1.0]) if forward else rng.normal(0, 1, 3)
    t = baseline * t / np.linalg.norm(t)
    X = np.stack([rng.uniform(-2, 2, 5), rng.uniform(-1.5, 1.5, 5), rng.uniform(3, 9, 5)], 1)
    X2 = X @ R.T + t
    q1 = X[:, :2] / X[:, 2:]; q2 = X2[:, :2] / X2[:, 2:]
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return q1, q2, E / np.linalg.norm(E) * np.sqrt(2.0), R, t


def test_solutions_are_essential_matrices_and_contain_the_planted_one():
    from voldor_amd import kernels
    rng = np.random.default_rng(11)
    n_trials = 1000
    n_sol, resid, dist = [], [], []
    for i in range(n_trials):
        q1, q2, E0, _, _ = _geometry(rng, forward=(i % 5 == 0), baseline=(0.05 if i % 7 == 0 else 1.0))
        Es = kernels.fivept_solve(q1, q2)
        n_sol.append(len(Es))
        assert len(Es) <= 10
        h1 = np.concatenate([q1, np.ones((5, 1))], 1); h2 = np.concatenate([q2, np.ones((5, 1))], 1)
        for E in Es:
            assert abs(np.linalg.norm(E) - np.sqrt(2.0)) < 1e-9
            assert np.abs(np.einsum("ki,ij,kj->k", h2, E, h1)).max() < 1e-8  # q'^T E q = 0 for the five correspondences (E lies in the null space by construction)
            resid.append(np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max())
        dist.append(min([min(np.abs(E - E0).max(), np.abs(E + E0).max()) for E in Es] + [9.0]))
    resid, dist = np.array(resid), np.array(dist)
    print(f"solutions per sample: mean {np.mean(n_sol):.2f}, max {max(n_sol)}; essential-matrix constraint residual: median {np.median(resid):.1e}, 99th percentile "
          f"{np.percentile(resid, 99):.1e}; planted E found to 1e-9 in {(dist < 1e-9).sum()} of {n_trials} samples")
    # measured: residual 2e-16 / 9e-16 (median / 99th percentile), planted E found in 999 of 1000 samples (984 when the system was solved in one basis of the null space only: a root of the degree-10
    # polynomial lost in a cluster of roots -- small baselines and pure forward motion make up a third of these samples)
    assert np.percentile(resid, 99) < 1e-10 and np.mean(resid < 1e-6) > 0.995
    assert (dist < 1e-9).sum() >= 0.99 * n_trials
    assert max(n_sol) >= 6 and np.mean(n_sol) > 3.0  # several real roots per sample, as the degree-10 polynomial allows


def _angle(Ra, Rb):
    return float(np.arccos(np.clip((np.trace(Ra.T @ Rb) - 1) / 2, -1, 1)))


@pytest.mark.parametrize("seed,outliers", [(3, 0.4), (5, 0.4), (7, 0.25)])
def test_five_point_lmeds_bootstrap_recovers_a_planted_pose_under_outliers(seed, outliers):
    from voldor_amd import kernels, synth
    sc = synth.make_scene(w=320, h=240, n_flows=1, fx=160, fy=160, cx=160, cy=120, seed=seed, outlier_frac=0.0, moving_patch=False)
    flow = sc["flows"][0].copy()
    rng = np.random.default_rng(seed)
    m = rng.uniform(size=flow.shape[:2]) < outliers  # gross outliers: uniform flow up to +-30 pixels
    flow[m] = rng.uniform(-30, 30, (int(m.sum()), 2)).astype(np.float32)
    K = K9(*sc["K"])
    ok5, R5, t5 = kernels.estimate_pose_epipolar5(flow, K)
    ok8, R8, t8 = kernels.estimate_pose_epipolar(flow, K)
    assert ok5 and ok8
    rv, tg = sc["poses_gt"][0, :3].astype(np.float64), sc["poses_gt"][0, 3:].astype(np.float64)
    Rg = _rodrigues(rv)
    dirg = tg / np.linalg.norm(tg)
    for name, R, t in (("5-point", R5, t5), ("8-point", R8, t8)):
        d = t / np.linalg.norm(t)
        print(f"{name}: rotation error {_angle(R.astype(np.float64), Rg):.2e} rad, translation direction error {np.arccos(np.clip(d @ dirg, -1, 1)):.2e} rad")
    # measured (seeds 3, 5, 7) with the 134 subsets OpenCV's LMedS draws at confidence 0.999 (round 6; 192 subsets in rounds 4-5 gave 2.0 / 3.2 / 0.9 mrad):
    # five-point rotation error 4.8 / 2.1 / 0.9 mrad and translation-direction error 0.033 / 0.046 / 0.005 rad against the 8-point bootstrap's (256 subsets)
    # 2.5 / 4.0 / 2.6 mrad and 0.027 / 0.095 / 0.048 rad on the same noisy flows (sub-pixel flow noise + the gross outliers): at 40 % outliers one subset
    # in thirteen is clean, and the least median over ~10 clean subsets is a noisy pick either way
    e5r, e5t = _angle(R5.astype(np.float64), Rg), np.arccos(np.clip((t5 / np.linalg.norm(t5)) @ dirg, -1, 1))
    e8r, e8t = _angle(R8.astype(np.float64), Rg), np.arccos(np.clip((t8 / np.linalg.norm(t8)) @ dirg, -1, 1))
    assert e5r < 5e-3 and e5t < 7e-2
    assert e5r <= 2.0 * e8r + 1e-3 and e5t <= 1.5 * e8t + 5e-3  # in the class of the 8-point estimator it sits next to
    # the two bootstraps estimate the same pose: their disagreement stays within the sum of what they show against ground truth
    assert _angle(R5.astype(np.float64), R8.astype(np.float64)) <= e5r + e8r + 1e-6
    assert np.arccos(np.clip((t5 / np.linalg.norm(t5)) @ (t8 / np.linalg.norm(t8)), -1, 1)) <= e5t + e8t + 1e-6


@pytest.mark.gpu
def test_five_point_bootstrap_kernels_give_the_host_bits(small_scene):
    """k_boot_hyp5 + k_boot_score + k_boot_select on the MI355X against the host build of the same source (fp64, no contraction), and the
    window pipeline with --bootstrap_points 5 registers the whole window."""
    from voldor_amd import kernels, pyvoldor
    K = K9(*small_scene["K"])
    ok, Rh, th = kernels.estimate_pose_epipolar5(small_scene["flows"][0], K)
    Rg, tg, _ = kernels.bootstrap_gpu(small_scene["flows"][0], K, points=5)
    assert ok
    np.testing.assert_array_equal(Rg, Rh)
    np.testing.assert_array_equal(tg, th)
    fx, fy, cx, cy = small_scene["K"]
    kernels.set_rand_epoch(0)
    a = pyvoldor.voldor(small_scene["flows"], fx, fy, cx, cy, config="--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 4 --bootstrap_points 5")
    kernels.set_rand_epoch(0)
    b = pyvoldor.voldor(small_scene["flows"], fx, fy, cx, cy, config="--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 4")
    assert a["n_registered"] == b["n_registered"] == small_scene["flows"].shape[0]
    from voldor_amd import synth
    gt = small_scene["poses_gt"].copy()
    gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))  # monocular windows are normalised to mean |t| = 1
    (r5, t5), (r8, t8) = synth.pose_errors(a["poses"], gt), synth.pose_errors(b["poses"], gt)
    print(f"window from the five-point bootstrap: rot {r5.max():.2e} rad, rel. trans {t5.max():.2e}; from the 8-point bootstrap: {r8.max():.2e}, {t8.max():.2e}")
    # the bootstrap only seeds the EM: both windows end at the same accuracy against ground truth (160x120, 4 iterations: a noisy estimate either way)
    assert r5.max() <= 1.5 * r8.max() + 1e-3 and t5.max() <= 1.5 * t8.max() + 2e-2, (r5, t5, r8, t8)


def test_solution_sets_equal_those_of_an_independent_solver():
    """oracle/orc_fivept.py solves the same ten cubic constraints by the action-matrix / eigenvector method of Stewenius et al. (numpy SVD + eig): no
    code and no formulation shared with vk_fivept.hpp (degree-10 polynomial, Laguerre roots with deflation).  (a) PRECISION: every essential matrix
    the product returns is one the oracle returns (to 1e-6, up to sign) -- no spurious models; (b) RECALL: on generic geometries the two sets are the
    same set in >= 97 % of the samples and the product finds >= 99 % of the oracle's matrices; on the hard ones (pure forward motion with a 5 %
    baseline: near-multiple roots) the eigen-solver keeps roots the deflation loses -- measured and printed, the LMedS on top draws 192 samples."""
    from oracle import orc_fivept
    from voldor_amd import kernels
    rng = np.random.default_rng(12)

    def partner(E, S):
        return min([min(np.abs(E - F).max(), np.abs(E + F).max()) for F in S] + [9.0])
    stats = {False: dict(samples=0, same=0, prod=0, spurious=0, orc=0, found=0, planted_p=0, planted_o=0), True: None}
    stats[True] = dict(stats[False])
    for i in range(400):
        hard = i % 5 == 0
        q1, q2, E0, _, _ = _geometry(rng, forward=hard, baseline=(0.05 if hard else 1.0))
        P = kernels.fivept_solve(q1, q2)
        O = orc_fivept.solve(q1, q2)
        st = stats[hard]
        st["samples"] += 1
        tol = 1e-4 if hard else 1e-6  # the hard samples are ill conditioned: a root is good to ~1e-5 in either solver
        sp = sum(1 for E in P if partner(E, O) > tol); fo = sum(1 for F in O if partner(F, P) <= tol)
        st["prod"] += len(P); st["spurious"] += sp; st["orc"] += len(O); st["found"] += fo
        st["same"] += int(len(P) == len(O) and sp == 0 and fo == len(O))
        st["planted_p"] += int(partner(E0, P) < tol); st["planted_o"] += int(partner(E0, O) < tol)
    for hard in (False, True):
        st = stats[hard]
        print(f"{'hard' if hard else 'generic'}: {st['samples']} samples, identical sets {st['same']}, product models {st['prod']} (not in the oracle's set: {st['spurious']}), "
              f"oracle models {st['orc']} (found by the product: {st['found']}), planted E found by product / oracle: {st['planted_p']} / {st['planted_o']}")
    g, h = stats[False], stats[True]
    # measured: generic 320 / 320 identical sets (1588 models each), planted matrix 320 / 320 in both; hard (80 samples): product 409 models, 7 without
    # a partner at 1e-4 (none at 1e-2), 394 of the oracle's 410 found, planted matrix 76 (product) / 78 (oracle).  Solving in ONE basis of the null
    # space the product found the planted matrix in 62 of the 80 -- the degree-10 polynomial loses near-multiple roots; the union over two bases is
    # what closed the gap to the eigenvalue method (vk_fivept.hpp solve)
    assert g["spurious"] == 0 and g["same"] == g["samples"] and g["found"] == g["orc"]
    assert g["planted_o"] == g["samples"] and g["planted_p"] == g["samples"]
    assert h["spurious"] <= 0.03 * h["prod"] and h["found"] >= 0.93 * h["orc"]
    assert h["planted_o"] >= 0.95 * h["samples"] and h["planted_p"] >= 0.9 * h["samples"]


def _flow_field(w, h, f, depth, R, t):
    """exact flow of a depth map under the motion X2 = R X + t (intrinsics f, principal point at the image centre)"""
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    X = np.stack([(xs - w / 2) / f * depth, (ys - h / 2) / f * depth, depth], -1)
    Y = X @ R.T + t
    return np.stack([f * Y[..., 0] / Y[..., 2] + w / 2 - xs, f * Y[..., 1] / Y[..., 2] + h / 2 - ys], -1).astype(np.float32)


def _median_sampson_px2(flow, f, R, t, step=4):
    """median squared Sampson distance (pixels^2) of the flow's correspondences under the epipolar geometry of (R, t)"""
    h, w, _ = flow.shape
    ys, xs = np.mgrid[0:h:step, 0:w:step].astype(np.float64)
    fl = flow[::step, ::step].astype(np.float64)
    q1 = np.stack([(xs - w / 2) / f, (ys - h / 2) / f, np.ones_like(xs)], -1).reshape(-1, 3)
    q2 = np.stack([(xs + fl[..., 0] - w / 2) / f, (ys + fl[..., 1] - h / 2) / f, np.ones_like(xs)], -1).reshape(-1, 3)
    t = np.asarray(t, np.float64); R = np.asarray(R, np.float64)
    E = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]) @ R
    Eq1 = q1 @ E.T; Etq2 = q2 @ E
    num = np.einsum("ki,ki->k", q2, Eq1) ** 2
    den = Eq1[:, 0] ** 2 + Eq1[:, 1] ** 2 + Etq2[:, 0] ** 2 + Etq2[:, 1] ** 2
    return float(np.median(num / den)) * f * f


def test_a_single_plane_separates_the_five_point_from_the_eight_point_estimator():
    """what the reference's estimator (five-point inside LMedS, voldor/geometry.cpp:316-326) can do and the 8-point LMedS of rounds 1-5
    cannot.  All points on ONE plane: the linear 8-point system is rank deficient (a three-dimensional solution space: its answer is arbitrary), the
    five-point solver is not degenerate there -- the planted essential matrix is among its solutions (and among the independent solver's,
    oracle/orc_fivept.py), with the one ambiguity planar scenes have (a second motion / plane pair that explains the same flow; OpenCV's LMedS has it too:
    the least median cannot tell two exact fits apart).  Host build of the bootstrap's own source, no GPU."""
    from oracle import orc_fivept
    from voldor_amd import kernels
    w, h, f = 320, 240, 160.0
    K = K9(f, f, w / 2, h / 2)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    ray = np.stack([(xs - w / 2) / f, (ys - h / 2) / f, np.ones_like(xs)], -1)
    planted5 = fits5 = broken8 = 0
    for seed in range(6):
        rng = np.random.default_rng(seed)
        R = _rodrigues(rng.normal(0, 0.03, 3)); t = rng.normal(0, 1, 3); t = 0.3 * t / np.linalg.norm(t)
        n = np.array([0.2 * rng.normal(), 0.3 * rng.normal(), 1.0]); n /= np.linalg.norm(n)
        depth = 6.0 / (ray @ n)
        flow = _flow_field(w, h, f, depth, R, t) + rng.normal(0, 0.05, (h, w, 2)).astype(np.float32)
        # (a) the minimal solver on five points of the plane: the planted matrix is a solution, in the product and in the oracle
        idx = rng.choice(w * h, 5, replace=False); py, px = idx // w, idx % w
        q1 = np.stack([(px - w / 2) / f, (py - h / 2) / f], 1)
        X2 = (ray[py, px] * depth[py, px, None]) @ R.T + t
        q2 = X2[:, :2] / X2[:, 2:]
        tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]]); E0 = tx @ R; E0 *= np.sqrt(2.0) / np.linalg.norm(E0)
        near = lambda S: min([min(np.abs(E - E0).max(), np.abs(E + E0).max()) for E in S] + [9.0])
        assert near(kernels.fivept_solve(q1, q2)) < 1e-6 and near(orc_fivept.solve(q1, q2)) < 1e-6
        # (b) eight points of the plane: the 8 x 9 epipolar system has a null space of dimension 3
        idx8 = rng.choice(w * h, 8, replace=False); py, px = idx8 // w, idx8 % w
        a1 = np.concatenate([np.stack([(px - w / 2) / f, (py - h / 2) / f], 1), np.ones((8, 1))], 1)
        Y = (ray[py, px] * depth[py, px, None]) @ R.T + t; a2 = Y / Y[:, 2:]
        sv = np.linalg.svd(np.stack([np.outer(a2[i], a1[i]).ravel() for i in range(8)]), compute_uv=False)
        assert sv[6] < 1e-9 * sv[0] and sv[5] > 1e-6 * sv[0], sv  # rank 6
        # (c) the two bootstraps on the whole flow field
        ok5, R5, t5 = kernels.estimate_pose_epipolar5(flow, K); ok8, R8, t8 = kernels.estimate_pose_epipolar(flow, K)
        assert ok5 and ok8
        dirg = t / np.linalg.norm(t)
        e5 = (_angle(R5.astype(np.float64), R), np.arccos(np.clip(t5 / np.linalg.norm(t5) @ dirg, -1, 1)))
        e8 = (_angle(R8.astype(np.float64), R), np.arccos(np.clip(t8 / np.linalg.norm(t8) @ dirg, -1, 1)))
        m5, m8 = _median_sampson_px2(flow, f, R5, t5), _median_sampson_px2(flow, f, R8, t8)
        print(f"plane, seed {seed}: five-point rot {e5[0]:.1e} rad, t direction {e5[1]:.1e} rad, median Sampson {m5:.1e} px^2 | 8-point {e8[0]:.1e}, {e8[1]:.1e}, {m8:.1e} px^2")
        planted5 += int(e5[0] < 2e-3 and e5[1] < 0.15); fits5 += int(m5 < 0.1); broken8 += int(e8[0] > 2e-2 and e8[1] > 0.5)
    # measured (6 scenes): the five-point bootstrap returns the planted pose in 4 (rotation 3e-4 - 6e-4 rad, translation direction 0.01 - 0.1 rad) and the
    # plane's second interpretation in 2 (0.04 rad / 1 rad) -- a fit of the flow either way (median Sampson distance 1e-3 - 7e-2 px^2 at 0.05 px flow noise);
    # the 8-point bootstrap never returns the planted pose: off by 0.04 - 0.05 rad in rotation and 0.9 - 1.9 rad in the translation direction in all 6,
    # with medians just as small -- on a plane a small residual proves nothing, which is the degeneracy
    assert fits5 == 6 and planted5 >= 3 and broken8 == 6


def test_forward_motion_bootstrap():
    """The hard case of the degree-10 polynomial (near-multiple roots: why the system is solved in two bases): forward motion over a scene with relief.  The
    five-point LMedS bootstrap recovers the planted pose as the 8-point one does."""
    from voldor_amd import kernels
    w, h, f = 320, 240, 160.0
    K = K9(f, f, w / 2, h / 2)
    ys, xs = np.mgrid[0:h, 0:w].astype(np.float64)
    depth = 4 + 3 * np.sin(xs / 40) ** 2 + 2 * (ys / h)
    for seed in range(4):
        rng = np.random.default_rng(100 + seed)
        R = _rodrigues(rng.normal(0, 0.01, 3)); t = np.array([0.02, 0.01, 0.3])
        flow = _flow_field(w, h, f, depth, R, t) + rng.normal(0, 0.05, (h, w, 2)).astype(np.float32)
        ok5, R5, t5 = kernels.estimate_pose_epipolar5(flow, K); ok8, R8, t8 = kernels.estimate_pose_epipolar(flow, K)
        assert ok5 and ok8
        dirg = t / np.linalg.norm(t)
        e5 = (_angle(R5.astype(np.float64), R), np.arccos(np.clip(t5 / np.linalg.norm(t5) @ dirg, -1, 1)))
        e8 = (_angle(R8.astype(np.float64), R), np.arccos(np.clip(t8 / np.linalg.norm(t8) @ dirg, -1, 1)))
        print(f"forward motion, seed {seed}: five-point rot {e5[0]:.1e} rad, t direction {e5[1]:.1e} rad | 8-point {e8[0]:.1e}, {e8[1]:.1e}")
        assert e5[0] < 2e-3 and e5[1] < 4e-2 and e8[0] < 3e-3 and e8[1] < 4e-2  # measured: 0 - 7e-4 / 8e-3 - 2e-2 (five-point), 4e-4 - 1.3e-3 / 7e-3 - 1.2e-2 (8-point)
