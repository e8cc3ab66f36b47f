"""The HIP window call (pyvoldor.voldor -> py_voldor_wrapper of libvoldor_hip.so) against the REFERENCE's own
py_voldor_wrapper outputs (tests/golden/ref_window.npz: voldor/*.cpp + gpu-kernels/*.cu executed on the CPU, see
tests/golden/gen_golden_window.py).  No oracle in between.

The product keeps the documented deviations (D1 RNG and D2 bilinear are in the goldens too; D3b re-draws the hypotheses, D4
uses one depth buffer, rodrigues uses the exact polar factor), so the two runs are two samples of the same estimator: same
registered frame count, poses within its sampling noise (north_star: 1e-3 rad; translation against the noise floor of the
8192-hypothesis mean-shift mode, DESIGN.md parity budget), confident depth within a few percent.
"""
import os

import numpy as np
import pytest

import ref_window_cases as cases

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window.npz")
CASES = list(cases.window_cases())


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


@pytest.mark.parametrize("name,c", CASES, ids=[n for n, _ in CASES])
def test_window_vs_reference_pipeline(gold, name, c):
    from voldor_amd import kernels, pyvoldor, synth
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                        depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"])
    n = int(gold[f"{name}/n_registered"])
    assert g["n_registered"] == n
    rot, tr = synth.pose_errors(g["poses"], gold[f"{name}/poses"])
    if name == "low_density":  # camera 1 rests on ~20 pixels whose membership (rigidness > 0.9995) flips with fp32 rounding: compared bit for bit
        rot, tr = rot[:1], tr[:1]  # in strict mode (tests/test_gpu_strict.py::test_low_density_window_keeps_the_reference_pool)
    big = not c["exact"]
    rot_tol, tr_tol = (1e-3, 3e-2) if big else (2e-3, 8e-2)  # small windows: a few thousand correspondences per camera
    assert rot.max() < rot_tol and tr.max() < tr_tol, (rot, tr)
    if big:
        ref_depth, ref_conf = gold[f"{name}/depth_sub2"], gold[f"{name}/depth_conf_sub2"]
        depth, conf = g["depth"][::2, ::2], g["depth_conf"][::2, ::2]
    else:
        ref_depth, ref_conf = gold[f"{name}/depth"], gold[f"{name}/depth_conf"]
        depth, conf = g["depth"], g["depth_conf"]
    s = np.mean(np.linalg.norm(g["poses"][:, 3:], axis=1)) / np.mean(np.linalg.norm(gold[f"{name}/poses"][:, 3:], axis=1))
    if name == "low_density":
        s = 1.0  # a stereo window has a metric scale; camera 1's translation (see above) must not rescale the comparison
    m = (conf > 0.5) & (ref_conf > 0.5)
    rel = np.abs(depth[m] / s - ref_depth[m]) / ref_depth[m]
    assert m.mean() > 0.3 and np.median(rel) < 3e-2, (m.mean(), np.median(rel))
    # against analytic ground truth too (monocular windows up to scale)
    if name.startswith(("mono_320", "stereo_312")):
        gt = c["poses_gt"].copy()
        if c["disparity"] is None:
            gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
        r2, t2 = synth.pose_errors(g["poses"], gt[:n])
        r3, t3 = synth.pose_errors(gold[f"{name}/poses"], gt[:n])
        assert r2.max() < 3e-3 and t2.max() < 5e-2
        assert r3.max() < 3e-3 and t3.max() < 5e-2  # and so is the reference itself


def test_accuracy_distribution_matches_the_reference(gold):
    """Two samples of one estimator cannot be compared pose by pose below its sampling noise, but their ACCURACY can: over 8
    independent 320x240 monocular windows (40 poses) the HIP path's error against analytic ground truth must have the same
    size as the reference pipeline's (tests/golden/ref_window.npz "ens*").  Measured: RMS rotation error 6.41e-4 rad (HIP)
    vs 6.42e-4 (reference), RMS relative translation error 1.62e-2 vs 1.44e-2; HIP vs reference pairwise 2.9e-4 rad /
    1.2e-2 -- the two runs share the flow noise but not the hypothesis draws (D3b), so they are as far from each other as
    each is from the truth."""
    from voldor_amd import kernels, pyvoldor, synth
    err = {"hip": [], "ref": [], "pair": []}
    for name, c in cases.ensemble_cases():
        fx, fy, cx, cy = c["K"]
        kernels.set_rand_epoch(0)
        g = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"])
        ref_poses = gold[f"{name}/poses"]
        assert g["n_registered"] == int(gold[f"{name}/n_registered"]) == 5
        gt = c["poses_gt"].copy()
        gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))  # monocular windows are normalised to mean |t| = 1
        err["hip"].append(np.stack(synth.pose_errors(g["poses"], gt)))
        err["ref"].append(np.stack(synth.pose_errors(ref_poses, gt)))
        err["pair"].append(np.stack(synth.pose_errors(g["poses"], ref_poses)))
    rms = {k: np.sqrt(np.mean(np.concatenate(v, axis=1) ** 2, axis=1)) for k, v in err.items()}  # [rot, rel-trans]
    worst = {k: np.concatenate(v, axis=1).max(axis=1) for k, v in err.items()}
    # same accuracy: RMS error of 40 poses within a factor 1.5 of the reference's, for rotation and translation (measured 1.00
    # and 1.13; the bar leaves room for the sampling noise of another 40-pose draw, ~ +-15 %)
    assert np.all(rms["hip"] < 1.5 * rms["ref"]), (rms["hip"], rms["ref"])
    assert np.all(rms["hip"] > 0.5 * rms["ref"]), (rms["hip"], rms["ref"])  # and not suspiciously better either
    assert worst["hip"][0] < 3e-3 and worst["hip"][1] < 5e-2 and worst["ref"][0] < 3e-3 and worst["ref"][1] < 5e-2
    # pairwise: north_star's 1e-3 rad holds for every pose; translation: the worst of 40 is held to the bound each run is held to
    # against the truth (a different summation order in the mode kernel moves it between 2.9e-2 and 3.1e-2: it sits at the noise floor)
    assert worst["pair"][0] < 1e-3 and worst["pair"][1] < 5e-2, worst["pair"]
    print("rms rot/trans  hip", rms["hip"], " ref", rms["ref"], " hip-vs-ref", rms["pair"])


def test_baseline_cfg2_window_vs_reference_pipeline(gold):
    """BASELINE configs[1] itself -- 640x480, N_flow=5, 8 EM iterations, monocular, the window bench.py times -- against the
    reference pipeline's poses for the same flows (north_star: 1e-3 rad; translation at the estimator's noise floor)."""
    from voldor_amd import kernels, pyvoldor, synth
    name, c = cases.cfg2_case()
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"])
    assert g["n_registered"] == int(gold[f"{name}/n_registered"]) == 5
    rot, tr = synth.pose_errors(g["poses"], gold[f"{name}/poses"])
    assert rot.max() < 1e-3 and tr.max() < 3e-2, (rot, tr)
    gt = c["poses_gt"].copy()
    gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
    for poses in (g["poses"], gold[f"{name}/poses"]):
        r, t = synth.pose_errors(poses, gt)
        assert r.max() < 2e-3 and t.max() < 4e-2, (r, t)


# ---- strict math + the reference's draw + the reference's SVD: a HIP window IS the reference's window ---------------------------
STRICT_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window_strict.npz")
ENSEMBLE_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_ensemble.npz")
REFERENCE_MODE = " --strict_math 1 --reference_draw 1 --reference_svd 1"


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("name", ["mono_nonexclusive", "stereo_default", "stereo_ap3p", "depth_priors"])
def test_strict_window_equals_the_reference(name):
    """tests/golden/ref_window_strict.npz = the REFERENCE's own pipeline (voldor/*.cpp + gpu-kernels/*.cu, executed on the CPU) with
    its libm calls served by vk_strict_math.h.  `--strict_math 1 --reference_draw 1 --reference_svd 1` switches the product's three
    computational deviations (hardware transcendentals / re-associated sums, D3b rejection draw, D8 exact polar factor) to the
    reference's behaviour; D1 (counter RNG), D2 (exact bilinear) and D5 (injected two-view pose) are in the golden run too.  Then
    the window must equal the reference's in EVERY bit of the registered count, depth map, confidence map and covariances -- no
    oracle in between.  Poses: the reference returns Camera::pose6() = cv::Rodrigues(cv::Rodrigues(rvec)) (utils.h:44-53; OpenCV,
    not in the reference tree), a round trip through a 3x3 float matrix that moves an rvec by at most ~2 ulp: < 1e-9."""
    from voldor_amd import kernels, pyvoldor
    g = np.load(STRICT_GOLD)
    c = dict(CASES)[name]
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                        depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"] + REFERENCE_MODE)
    assert o["n_registered"] == int(g[f"{name}/n_registered"])
    for k in ("depth", "depth_conf", "poses_covar"):
        neq = _bits(o[k]) != _bits(g[f"{name}/{k}"])
        assert not neq.any(), f"{name}/{k}: {int(neq.sum())} of {neq.size} values differ from the reference"
    assert np.abs(o["poses"].astype(np.float64) - g[f"{name}/poses"]).max() < 1e-9
    # the switch matters: without the reference's SVD the same window is a different rounding of the same estimate
    kernels.set_rand_epoch(0)
    o2 = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                         depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"],
                         config=c["config"] + " --strict_math 1 --reference_draw 1")
    assert (_bits(o2["depth"]) != _bits(g[f"{name}/depth"])).any()


def test_strict_cfg2_window_equals_the_reference():
    """BASELINE configs[1] at full size (640x480, N=5, 8 EM iterations, monocular, refit on the last iteration): the reference
    pipeline's strict-math window (tests/golden/gen_golden_ensemble.py, `cfg2/s233/strict`: covariances, poses, sha256 of the two
    maps and every 8th pixel of them) against the HIP window in reference mode."""
    import hashlib
    from voldor_amd import kernels, pyvoldor
    if not os.path.exists(ENSEMBLE_GOLD):
        pytest.skip("tests/golden/ref_ensemble.npz not generated")
    g = np.load(ENSEMBLE_GOLD)
    name, c = cases.cfg2_case()
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"] + REFERENCE_MODE)
    p = "cfg2/s233/strict/"
    assert o["n_registered"] == int(g[p + "n_registered"]) == 5
    assert not (_bits(o["depth"][::8, ::8]) != _bits(g[p + "depth_sub"])).any()
    assert not (_bits(o["depth_conf"][::8, ::8]) != _bits(g[p + "conf_sub"])).any()
    assert hashlib.sha256(np.ascontiguousarray(o["depth"]).tobytes()).digest() == g[p + "depth_sha256"].tobytes()
    assert hashlib.sha256(np.ascontiguousarray(o["depth_conf"]).tobytes()).digest() == g[p + "conf_sha256"].tobytes()
    assert not (_bits(o["poses_covar"]) != _bits(g[p + "poses_covar"])).any()
    assert np.abs(o["poses"].astype(np.float64) - g[p + "poses"]).max() < 1e-9
