"""Seeded inputs shared by tests/golden/gen_golden_kernels.py (reference kernels on the CPU -> ref_kernels.npz) and
tests/test_oracle_vs_ref_kernels.py (oracle vs those goldens).  Everything is derived from fixed seeds with numpy's
PCG64 and voldor_amd.synth, so only the reference OUTPUTS are stored."""
import numpy as np

from voldor_amd import synth

import align_scene

OD_DEFAULTS = dict(abs_resize_factor=1.0, basefocal=0.0, n_rand_samples=10, global_prop_step=8, local_prop_width=32,
                   lambda_=0.15, omega=0.15, disp_delta=-1.0, delta=0.5, fb_smooth=1, s0_ems_prob=0.5,
                   no_change_prob=0.9, range_factor=1.0, update_rigidness_only=0)


def fb_cases():
    rng = np.random.default_rng(11)
    yield "64x48x3", rng.uniform(0.01, 0.99, (3, 48, 64)).astype(np.float32), 0.5, 0.9
    yield "131x77x2", rng.uniform(0.0, 1.0, (2, 77, 131)).astype(np.float32), 0.4, 0.8
    m = rng.uniform(0.3, 0.7, (1, 20, 300)).astype(np.float32)  # rows longer than the reference's 128-thread chain block
    m[0, 3, :] = 0.0
    m[0, :, 150] = 1.0
    yield "300x20x1", m, 0.5, 0.97


def _scene(w, h, n, seed, basefocal=30.0):
    f = 0.5 * w
    sc = synth.make_scene(w=w, h=h, n_flows=n, fx=f, fy=f, cx=0.5 * w, cy=0.5 * h, seed=seed, basefocal=basefocal)
    K = np.array([f, 0, 0.5 * w, 0, f, 0.5 * h, 0, 0, 1], np.float32)
    gt = sc["poses_gt"]
    Rs = np.stack([synth.rodrigues(gt[i, :3]) for i in range(n)]).astype(np.float32).reshape(n, 9)
    ts = np.ascontiguousarray(gt[:, 3:], np.float32)
    return sc, K, Rs, ts


def _depth_base(w, h, n, seed):
    sc, K, Rs, ts = _scene(w, h, n, seed)
    rng = np.random.default_rng(seed + 1)
    depth = (sc["depth_gt"] * (1 + rng.normal(0, 0.2, (h, w)))).astype(np.float32)
    rig = rng.uniform(0.2, 1.0, (n, h, w)).astype(np.float32)
    priors = np.stack([30.0 / sc["disparity"], sc["depth_gt"] * 1.02]).astype(np.float32)
    priors[1, :5, :] = 0  # invalid prior band
    pconfs = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32)
    confs = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32)
    dpRs = np.stack([np.eye(3), synth.rodrigues([0.002, -0.001, 0.001])]).astype(np.float32).reshape(2, 9)
    dpts = np.array([[0, 0, 0], [0.01, 0, -0.05]], np.float32)
    return dict(flows=np.ascontiguousarray(sc["flows"], np.float32), rig=rig, depth=depth, K=K, Rs=Rs, ts=ts,
                _priors=(priors, pconfs, confs, dpRs, dpts))


def _with(base, with_priors, rand_epoch=7, **over):
    c = dict(base)
    pri = c.pop("_priors")
    names = ("priors", "pconfs", "confs", "dpRs", "dpts")
    for k, v in zip(names, pri if with_priors else (None,) * 5):
        c[k] = v
    c["kw"] = dict(OD_DEFAULTS, **over)
    c["rand_epoch"] = rand_epoch
    return c


def depth_cases():
    """(name, case): every pixel pass of optimize_depth_gpu in isolation, then chained, then with depth priors."""
    b = _depth_base(64, 48, 3, 13)
    off = dict(n_rand_samples=0, global_prop_step=0, local_prop_width=0, fb_smooth=0)
    yield "cost", _with(b, False, **off)
    yield "rand", _with(b, False, **dict(off, n_rand_samples=10))
    yield "global", _with(b, False, **dict(off, global_prop_step=8))
    yield "local", _with(b, False, **dict(off, local_prop_width=32))
    yield "all", _with(b, False)
    yield "all_priors", _with(b, True, basefocal=30.0, disp_delta=1.0, delta=0.2)
    yield "update_only", _with(b, True, basefocal=30.0, update_rigidness_only=1)
    b2 = _depth_base(100, 75, 2, 29)  # ragged against the 16x16 / 64 / 16*32 launch shapes
    yield "ragged_all", _with(b2, False, rand_epoch=123, global_prop_step=5, local_prop_width=8, range_factor=0.7, abs_resize_factor=0.5)


def collect_cases():
    b = _with(_depth_base(64, 48, 3, 13), False)
    rng = np.random.default_rng(5)
    b["rig"] = np.clip(b["rig"] + rng.normal(0, 0.3, b["rig"].shape), 0, 1).astype(np.float32)
    b["depth"] = b["depth"].copy()
    b["depth"][10:14, 20:30] = 0.05  # below sample_min_depth
    std = dict(rigidness_thresh=0.5, rigidness_sum_thresh=1.0, sample_min_depth=0.1, sample_max_depth=0.0, max_trace_on_flow=0)
    yield "cam0", b, 0, std
    yield "cam1_trace2", b, 1, dict(std, max_trace_on_flow=2, sample_max_depth=20.0)
    yield "cam2", b, 2, dict(std, rigidness_thresh=0.3)
    yield "cam2_sumthresh", b, 2, dict(std, rigidness_sum_thresh=5.5, max_trace_on_flow=1)


def _pnp_points(seed, n_pts, noise_px):
    rng = np.random.default_rng(seed)
    K = np.array([320, 0, 320, 0, 320, 240, 0, 0, 1], np.float32)
    X = np.stack([rng.uniform(-4, 4, n_pts), rng.uniform(-3, 3, n_pts), rng.uniform(3, 20, n_pts)], 1)
    R = synth.rodrigues([0.01, -0.02, 0.005])
    t = np.array([0.05, -0.01, 0.3])
    Y = X @ R.T + t
    uv = np.stack([320 * Y[:, 0] / Y[:, 2] + 320, 320 * Y[:, 1] / Y[:, 2] + 240], 1) + rng.normal(0, noise_px, (n_pts, 2))
    out = rng.random(n_pts) < 0.1
    uv[out] += rng.uniform(-40, 40, (int(out.sum()), 2))
    return X.astype(np.float32), uv.astype(np.float32), K


def solve_cases():
    X, uv, K = _pnp_points(3, 700, 0.3)
    yield "lambdatwist", X, uv, K, 2048, False
    yield "ap3p", X, uv, K, 2048, True
    X2, uv2, K2 = _pnp_points(4, 37, 0.0)  # few points: repeated indices inside a hypothesis are likely
    yield "lambdatwist_small", X2, uv2, K2, 512, False
    yield "ap3p_small", X2, uv2, K2, 512, True


def _pose_pool(seed, n, inlier_frac, sigma):
    rng = np.random.default_rng(seed)
    centre = np.array([0.2, -0.4, 0.1, 0.05, -0.01, 0.3])
    n_in = int(n * inlier_frac)
    pool = np.concatenate([centre + rng.normal(0, sigma, (n_in, 6)), centre + rng.uniform(-1.5, 1.5, (n - n_in, 6))])
    rng.shuffle(pool)
    return pool.astype(np.float32), centre.astype(np.float32)


MS_DEFAULTS = dict(epsilon=1e-5, max_iters=100, max_init_trials=20, good_init_confidence=0.5)


def meanshift_cases():
    pool, centre = _pose_pool(21, 4096, 0.6, 0.03)
    yield "ext_init", pool, 1e-2, centre + np.float32(0.05), True, MS_DEFAULTS
    yield "rand_init", pool, 1e-2, np.zeros(6, np.float32), False, MS_DEFAULTS
    yield "rand_init_strict", pool, 1e-2, np.zeros(6, np.float32), False, dict(MS_DEFAULTS, good_init_confidence=0.9, max_init_trials=7)
    pool2, c2 = _pose_pool(22, 1000, 0.3, 0.1)  # N not a multiple of the 512-element reduction blocks
    yield "n1000_cap3", pool2, 4e-2, c2, True, dict(MS_DEFAULTS, max_iters=3)


RG_DEFAULTS = dict(trunc_sigma=3.0, covar_reg_lambda=1e-3, epsilon=1e-5, max_iters=100)


def rg_cases():
    pool, centre = _pose_pool(21, 4096, 0.6, 0.03)
    cov = (np.eye(6) * 1e-2).astype(np.float32)
    yield "wide_start", pool, centre + np.float32(0.01), cov, RG_DEFAULTS
    yield "no_reg", pool, centre, cov, dict(RG_DEFAULTS, covar_reg_lambda=0.0, trunc_sigma=2.5)
    pool2, c2 = _pose_pool(22, 1000, 0.3, 0.1)
    yield "n1000_cap2", pool2, c2, (np.eye(6) * 4e-2).astype(np.float32), dict(RG_DEFAULTS, max_iters=2)
    flat = np.tile(centre, (600, 1)).astype(np.float32)  # zero spread: singular covariance after the first M-step
    yield "singular", flat, centre, cov, dict(RG_DEFAULTS, covar_reg_lambda=0.0)


def align_cases():
    """(name, keyframe set, photometric?, [(eval name, ref_fid, tar_fid, params_ref, params_tar, want_jacobian, apply_weights)])"""
    kf = align_scene.keyframes(w=64, h=48, n=3, seed=2)
    rng = np.random.default_rng(8)
    P = kf["params"].copy()
    pert = P + rng.normal(0, 1, P.shape).astype(np.float32) * np.array([2e-3] * 3 + [2e-2] * 3 + [0, 0, 0], np.float32)
    scaled = pert.copy()
    scaled[:, 6] = [0.02, -0.03, 0.01]   # log depth scale
    scaled[:, 7] = [0.05, 0.0, -0.04]    # log colour scale
    scaled[:, 8] = [0.01, -0.02, 0.0]    # colour offset
    yield "photo", kf, True, [("f0_f1_jac", 0, 1, pert[0], pert[1], True, True),
                              ("f2_f0_scaled_jac", 2, 0, scaled[2], scaled[0], True, True),
                              ("f1_f2_truth_res", 1, 2, P[1], P[2], False, False)]
    yield "depth_only", kf, False, [("f1_f0_jac", 1, 0, scaled[1], scaled[0], True, False),
                                    ("f0_f2_res", 0, 2, pert[0], pert[2], False, True)]


def gblur_cases():
    """(name, src [d,h,w], sigma, ksize): gblur_gpu (gblur.cu:47-72); ksize 0 = max(ceil(6 sigma), 3)"""
    rng = np.random.default_rng(4)
    yield "2x37x53_s1.5", rng.uniform(0, 5, (2, 37, 53)).astype(np.float32), 1.5, 0
    yield "1x64x80_s3", rng.uniform(0, 5, (1, 64, 80)).astype(np.float32), 3.0, 0
    yield "1x20x30_s0.4", rng.uniform(0, 5, (1, 20, 30)).astype(np.float32), 0.4, 0   # kernel clamps to 3 taps
    yield "1x16x16_k9", rng.uniform(0, 5, (1, 16, 16)).astype(np.float32), 2.0, 9
    yield "too_wide", rng.uniform(0, 5, (1, 8, 8)).astype(np.float32), 50.0, 0       # half kernel > 128 taps: error
