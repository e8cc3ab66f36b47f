"""The HIP window call (pyvoldor.voldor -> py_voldor_wrapper of libvoldor_hip.so) against the REFERENCE's own
py_voldor_wrapper outputs (tests/golden/ref_window.npz: voldor/*.cpp + gpu-kernels/*.cu executed on the CPU, see
tests/golden/gen_golden_window.py).  No oracle in between.

The product's default mode keeps D1 (counter RNG) and D2 (exact bilinear) -- both are in the goldens too --, D4 (one depth buffer), D8
(exact polar factor) and its fast arithmetic (hardware transcendentals, re-associated sums); the hypothesis draw is the reference's.
Two kinds of statement, neither with a hand-set tolerance:
  * fast mode vs the reference: the distance of a window is ranked inside the reference's OWN distances under 1-ulp jitter of its libm
    (rank-sum test over the windows; tests/golden/ref_window_noise.npz), the accuracy against ground truth is compared as a distribution
    (Kolmogorov-Smirnov); the 24-window cfg2 / 8-window cfg3 ensembles are tests/test_gpu_ensemble.py;
  * reference mode (--strict_math 1 --reference_draw 1 --reference_svd 1) vs the reference in strict math: EQUALITY OF BITS.
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


NOISE = os.path.join(os.path.dirname(__file__), "golden", "ref_window_noise.npz")


def test_windows_vs_reference_pipeline_rank_test(gold):
    """Every window of tests/ref_window_cases.py (stereo, AP3P, monocular, depth priors, truncated, low density, CPU P3P, the two larger
    ones) through the fast HIP path against the reference's own run of it.  No tolerance: tests/golden/ref_window_noise.npz holds, per
    window, the reference's run under eight other seeds of its random streams (why seeds: see tests/golden/gen_golden_window_noise.py);
    the distance {HIP vs reference} is RANKED inside that window's sample of distances {re-seeded reference vs reference} -- two
    independent draws of the reference's own estimator on the same flows --, and the ranks of all windows are summed
    (exact null distribution; one-sided p > 0.01) for rotation, translation, 90th-percentile depth difference, fraction of the confident
    pixels within 1e-3, and covariance trace.  Registered counts must be equal; a gross-error guard (10x the largest self-distance of the
    window) names the window if a code path is wrong."""
    import stat_helpers as sh
    from voldor_amd import kernels, pyvoldor
    nz = np.load(NOISE)
    vals = {k: [] for k in sh.METRICS}
    samples = {k: [] for k in sh.METRICS}
    for name, c in CASES:
        fx, fy, cx, cy = c["K"]
        kernels.set_rand_epoch(0)
        o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                            depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"])
        sub = 2 if c["exact"] else 4
        hip = {"n_registered": o["n_registered"], "poses": o["poses"], "poses_covar": o["poses_covar"], "depth": o["depth"][::sub, ::sub], "depth_conf": o["depth_conf"][::sub, ::sub]}
        def ref(salt):
            p = f"{name}/s{salt}/"
            return {"n_registered": int(nz[p + "n_registered"]), "poses": nz[p + "poses"], "poses_covar": nz[p + "poses_covar"], "depth": nz[p + "depth"], "depth_conf": nz[p + "depth_conf"]}
        r0 = ref(0)
        assert hip["n_registered"] == r0["n_registered"] == int(gold[f"{name}/n_registered"]), name
        if name.startswith("truncated"):
            assert 0 < hip["n_registered"] < c["flows"].shape[0]
        if c["stat_only"] or name.endswith("_b1"):
            continue  # the reference drew with libc rand() (cfg1) / ran with its stale device depth (B-1, D4): not the same estimator
        d = sh.window_distance(hip, r0)
        self_d = [sh.window_distance(ref(k), r0) for k in range(1, 9)]
        self_d = [x for x in self_d if x is not None]
        for k in sh.METRICS:
            smp = [x[k] for x in self_d]
            if k == "within_1e-3":  # larger is better: rank the shortfall
                vals[k].append(-d[k]); samples[k].append([-x for x in smp])
            else:
                vals[k].append(d[k]); samples[k].append(smp)
                if np.isfinite(d[k]) and len(smp):
                    assert d[k] <= 10 * np.nanmax(smp) + 1e-12, (name, k, d[k], np.nanmax(smp))
    for k in sh.METRICS:
        p = sh.rank_sum_pvalue(vals[k], samples[k])
        print(f"{k:12s} rank-sum p = {p:.3f} over {len(vals[k])} windows")
        assert p > 0.01, (k, p, vals[k])


def test_accuracy_distribution_matches_the_reference(gold):
    """Accuracy against analytic ground truth over 8 independent 320x240 monocular windows (40 poses): the per-pose rotation and
    translation errors of the HIP path and of the reference pipeline (tests/golden/ref_window.npz "ens*") must come from one
    distribution (two-sample KS, p > 0.01), and so must the pairwise distances HIP-vs-reference when set against ... the reference's
    own error (both are draws of one estimator around the truth: the pairwise distance cannot be larger in distribution than
    sqrt(2) x the error)."""
    import stat_helpers as sh
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
    e = {k: np.concatenate(v, axis=1) for k, v in err.items()}  # [rot | rel-trans] x 40 poses
    for i, what in enumerate(("rotation", "translation")):
        p = sh.ks_pvalue(e["hip"][i], e["ref"][i])
        p2 = sh.ks_pvalue(e["pair"][i], np.sqrt(2.0) * e["ref"][i])
        print(f"{what}: rms error hip {np.sqrt(np.mean(e['hip'][i] ** 2)):.2e} ref {np.sqrt(np.mean(e['ref'][i] ** 2)):.2e} (KS p = {p:.3f}); "
              f"pairwise rms {np.sqrt(np.mean(e['pair'][i] ** 2)):.2e} (vs sqrt2 x ref error: one-sided check below)")
        assert p > 0.01, (what, p)
        # pairwise distances must not be stochastically LARGER than sqrt(2) x the reference's error (smaller is fine: shared flow noise)
        assert np.median(e["pair"][i]) <= np.median(np.sqrt(2.0) * e["ref"][i]) or p2 > 0.01, (what, p2)


def test_baseline_cfg2_window_vs_reference_pipeline(gold):
    """BASELINE configs[1] itself -- 640x480, N_flow=5, 8 EM iterations, monocular, the window bench.py times -- against the reference
    pipeline's poses for the same flows: no further away than the reference's own runs are from each other under 1-ulp jitter over the
    24-window ensemble, no less accurate than the reference's worst window (tests/golden/ref_ensemble_bounds.json)."""
    import json
    from voldor_amd import kernels, pyvoldor, synth
    B = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_ensemble_bounds.json")))["cfg2"]
    name, c = cases.cfg2_case()
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"])
    assert g["n_registered"] == int(gold[f"{name}/n_registered"]) == 5
    rot, tr = synth.pose_errors(g["poses"], gold[f"{name}/poses"])
    # regression guard at twice the ensemble's extreme (a 49th draw exceeds the largest of 48 with probability 2 %); this very window
    # is one of the 24 of the distribution test tests/test_gpu_ensemble.py, which is the parity statement
    assert rot.max() <= 2 * B["self"]["rot"]["max"] and tr.max() <= 2 * B["self"]["trans"]["max"], (rot, tr)
    gt = c["poses_gt"].copy()
    gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
    for poses in (g["poses"], gold[f"{name}/poses"]):
        r, t = synth.pose_errors(poses, gt)
        assert r.max() <= 2 * B["gt"]["rot"]["max"] and t.max() <= 2 * B["gt"]["trans"]["max"], (r, t)


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
    oracle in between.  Poses: the reference returns Camera::pose6() = cv::Rodrigues(cv::Rodrigues(rvec)) (utils.h:44-53), a round trip
    through the 3x3 float matrix that moves an rvec by an ulp or two; since round 4 the strict kernels take the same round trip
    (voldor_amd/csrc/vk_ref_cv.h, shared with the OpenCV stand-in of the emulated reference) -- it is also what the next mean shift starts
    from (geometry.cpp:184), which decides windows whose mean shift runs into its iteration cap -- and the poses are compared bit for bit."""
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
    assert not (_bits(o["poses"]) != _bits(g[f"{name}/poses"])).any()  # Camera::pose6(): the strict kernels take its round trip through the float matrix (vk_ref_cv.h)
    # the switch matters: without the reference's SVD the same window is a different rounding of the same estimate
    kernels.set_rand_epoch(0)
    o2 = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                         depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"],
                         config=c["config"] + " --strict_math 1 --reference_draw 1")
    assert (_bits(o2["depth"]) != _bits(g[f"{name}/depth"])).any()


CUDA_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window_cuda.npz")
CUDA_MODE = REFERENCE_MODE + " --reference_rng 1 --reference_tex 1"


@pytest.mark.parametrize("name", ["mono_nonexclusive", "stereo_default", "stereo_ap3p", "depth_priors", "cfg2_640x480"])
def test_window_with_xorwow_and_texture_filter_equals_the_reference(name):
    """Reference mode with NOTHING stood in for but OpenCV's two-view bootstrap: tests/golden/ref_window_cuda.npz = the
    reference's own pipeline in strict math whose curand_init / curand_uniform draw from cuRAND's XORWOW streams and whose at_tex is CUDA's
    linear filter over the stacked layers (both restated in voldor_amd/csrc/vk_ref_cuda.h from their published definitions: KATs in
    tests/test_reference_cuda.py).  `--reference_rng 1 --reference_tex 1` on top of the reference mode: every output bit of the window --
    four small windows (monocular, stereo, AP3P, depth priors from other poses) and BASELINE cfg2 at full size."""
    import hashlib
    from voldor_amd import kernels, pyvoldor
    if not os.path.exists(CUDA_GOLD):
        pytest.skip("tests/golden/ref_window_cuda.npz not generated")
    g = np.load(CUDA_GOLD)
    c = cases.cfg2_case()[1] if name == "cfg2_640x480" else dict(CASES)[name]
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                        depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"] + CUDA_MODE)
    assert o["n_registered"] == int(g[f"{name}/n_registered"])
    if name == "cfg2_640x480":
        for k in ("depth", "depth_conf"):
            assert not (_bits(o[k][::8, ::8]) != _bits(g[f"{name}/{k}_sub8"])).any(), k
            assert hashlib.sha256(np.ascontiguousarray(o[k]).tobytes()).digest() == g[f"{name}/{k}_sha256"].tobytes(), k
    else:
        for k in ("depth", "depth_conf"):
            neq = _bits(o[k]) != _bits(g[f"{name}/{k}"])
            assert not neq.any(), f"{name}/{k}: {int(neq.sum())} of {neq.size} values differ from the reference"
    assert not (_bits(o["poses_covar"]) != _bits(g[f"{name}/poses_covar"])).any()
    assert not (_bits(o["poses"]) != _bits(g[f"{name}/poses"])).any()  # Camera::pose6(): the strict kernels take its round trip through the float matrix (vk_ref_cv.h)
    # each switch matters: with only one of them the window is another one
    for only in (" --reference_rng 1", " --reference_tex 1"):
        kernels.set_rand_epoch(0)
        o2 = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                             depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"] + REFERENCE_MODE + only)
        assert (_bits(o2["depth"]) != _bits(o["depth"])).any(), only


@pytest.mark.parametrize("name", ["mono_nonexclusive", "stereo_default"])
def test_texture_filter_window_does_not_read_flow_layers_that_are_not_up_yet(name):
    """host-resident flows go up frame by frame, each right before its first reader.  CUDA's linear filter over the STACK of layers
    (`--reference_tex 1`) blends the bottom row of layer i with the top row of layer i + 1, so camera i's trace reads one layer more than the fast
    path's -- the staggered upload has to be one frame ahead there.  The device copy of the flows is poisoned by a window of NaN flows of the same
    geometry on the same context; a reader that is ahead of its upload then sees NaNs (or the stale window) and the bits move."""
    from voldor_amd import kernels, pyvoldor
    if not os.path.exists(CUDA_GOLD):
        pytest.skip("tests/golden/ref_window_cuda.npz not generated")
    g = np.load(CUDA_GOLD)
    c = dict(CASES)[name]
    fx, fy, cx, cy = c["K"]
    kw = dict(basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"], depth_prior_poses=c["depth_prior_poses"],
              depth_prior_pconfs=c["depth_prior_pconfs"])
    for _ in range(3):
        kernels.set_rand_epoch(0)
        try:
            pyvoldor.voldor(np.full_like(c["flows"], np.nan), fx, fy, cx, cy, config=c["config"] + CUDA_MODE, **kw)  # leaves NaN layers on the device
        except Exception:
            pass  # (whatever the window makes of NaN flows: the layers are up)
        kernels.set_rand_epoch(0)
        o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"] + CUDA_MODE, **kw)
        assert o["n_registered"] == int(g[f"{name}/n_registered"])
        for k in ("depth", "depth_conf", "poses", "poses_covar"):
            assert not (_bits(o[k]) != _bits(g[f"{name}/{k}"])).any(), k


DEFAULT_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window_default.npz")


@pytest.mark.parametrize("name", ["mono_default_b1", "truncated_b1", "cfg2_640x480_default"])
def test_window_in_the_references_default_exclusive_mode_equals_the_reference(name):
    """Deviation D4 as a switch (round 4): tests/golden/ref_window_default.npz = the reference's own pipeline in strict math under its DEFAULT
    `exclusive_gpu_context 1`, where a monocular window shows SURVEY Appendix B-1 -- optimize_depth.cu keeps searching from a device copy of the
    depth map that never saw normalize_world_scale() (voldor.cpp:250-291, :309-317).  `--reference_stale_depth 1` on top of the reference mode
    keeps that second copy: every output bit of two small windows (one truncated) and of BASELINE cfg2 at full size.  Without the switch the
    window is the one the reference computes with `--exclusive_gpu_context 0` (the other tests of this file)."""
    import hashlib
    from voldor_amd import kernels, pyvoldor
    if not os.path.exists(DEFAULT_GOLD):
        pytest.skip("tests/golden/ref_window_default.npz not generated")
    g = np.load(DEFAULT_GOLD)
    big = name == "cfg2_640x480_default"
    c = cases.cfg2_case()[1] if big else dict(CASES)[name]
    fx, fy, cx, cy = c["K"]

    def run(extra):
        kernels.set_rand_epoch(0)
        return pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                               depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"] + REFERENCE_MODE + extra)
    o = run(" --reference_stale_depth 1")
    assert o["n_registered"] == int(g[f"{name}/n_registered"])
    if big:
        for k in ("depth", "depth_conf"):
            assert not (_bits(o[k][::8, ::8]) != _bits(g[f"{name}/{k}_sub8"])).any(), k
            assert hashlib.sha256(np.ascontiguousarray(o[k]).tobytes()).digest() == g[f"{name}/{k}_sha256"].tobytes(), k
    else:
        for k in ("depth", "depth_conf"):
            neq = _bits(o[k]) != _bits(g[f"{name}/{k}"])
            assert not neq.any(), f"{name}/{k}: {int(neq.sum())} of {neq.size} values differ from the reference"
    assert not (_bits(o["poses_covar"]) != _bits(g[f"{name}/poses_covar"])).any()
    assert not (_bits(o["poses"]) != _bits(g[f"{name}/poses"])).any()  # Camera::pose6(): the strict kernels take its round trip through the float matrix (vk_ref_cv.h)
    # the switch matters (the stale copy is real), and it needs the reference's default exclusive mode to do anything
    o2 = run("")
    assert (_bits(o2["depth"]) != _bits(o["depth"])).any()
    o3 = run(" --reference_stale_depth 1 --exclusive_gpu_context 0")
    assert not (_bits(o3["depth"]) != _bits(o2["depth"])).any()


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
    assert not (_bits(o["poses"]) != _bits(g[p + "poses"])).any()  # Camera::pose6(): the strict kernels take its round trip through the float matrix (vk_ref_cv.h)
