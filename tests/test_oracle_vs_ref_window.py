"""Pins the oracle's whole-window schedule (oracle/orc_voldor.c) against the REFERENCE's own py_voldor_wrapper.

tests/golden/ref_window.npz holds outputs of voldor/{py_export,voldor,geometry,utils}.cpp compiled in place (OpenCV calls
served by oracle/ref_stubs/minicv) on top of the reference's kernel files run on the CPU (tests/golden/gen_golden_window.py).

Two bars:
  * "reference mode": the oracle's three deliberate numerical deviations are switched to the reference's behaviour --
    ORC_REFERENCE_DRAW=1 (hypothesis indices into the compacted list instead of D3b's rejection draw), the reference's own
    approximate-SVD rodrigues() installed through orc_set_rodrigues_hook (instead of the exact polar factor), and, for runs
    of the reference in its default exclusive_gpu_context mode, ORC_EMULATE_B1=1 (the stale un-normalised device depth of
    SURVEY Appendix B-1 instead of D4's single depth buffer).  Then every output of the window -- registered count, depth map,
    confidence map, covariances, and (since round 4: the oracle takes the Rodrigues(Rodrigues(r)) round trip of Camera::rvec() /
    pose6() through the float matrix itself, utils.h:44-53, voldor_amd/csrc/vk_ref_cv.h) the poses -- must be BIT-IDENTICAL to the
    reference's.  This pins the EM schedule, the pose-pool
    scaling, the truncation rule, the world-scale normalisation and the depth-prior initialisation.
  * default mode (what the HIP path is compared with): same registered count, poses within the estimator's own sampling
    noise of the reference's (the deviations re-draw the hypotheses; see DESIGN.md parity budget).
"""
import ctypes as C
import os

import numpy as np
import pytest

import ref_window_cases as cases
from oracle import orc
from voldor_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window.npz")
CASES = list(cases.window_cases())


@pytest.fixture(scope="module")
def gold():
    return np.load(GOLD)


def run_oracle(c):
    fx, fy, cx, cy = c["K"]
    return orc.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                      depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"])


@pytest.fixture(params=["restated_svd", "hooked_reference_rodrigues"])
def reference_mode(monkeypatch, request):
    """The reference's rodrigues() either as restated in voldor_amd/csrc/vk_ref_svd.h (orc_set_reference_svd: what the HIP path
    runs under --reference_svd 1; needs nothing but the oracle) or as the reference's own code hooked in from oracle/_ref."""
    monkeypatch.setenv("ORC_REFERENCE_DRAW", "1")
    if request.param == "restated_svd":
        orc.set_reference_svd(True)
        yield monkeypatch
        orc.set_reference_svd(False)
        return
    ref = orc.ref()
    if ref is None or not hasattr(ref, "ref_rodrigues"):
        pytest.skip("oracle/_ref (the reference's rodrigues.h compiled in place) is not built on this box")
    orc.lib().orc_set_rodrigues_hook(C.cast(ref.ref_rodrigues, C.c_void_p))
    yield monkeypatch
    orc.lib().orc_set_rodrigues_hook(None)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


EXACT = [(n, c) for n, c in CASES if c["exact"] and not c["stat_only"]]


@pytest.mark.parametrize("name,c", EXACT, ids=[n for n, _ in EXACT])
def test_window_bit_identical_in_reference_mode(gold, reference_mode, name, c):
    if c["b1"]:
        reference_mode.setenv("ORC_EMULATE_B1", "1")
    o = run_oracle(c)
    n = int(gold[f"{name}/n_registered"])
    assert o["n_registered"] == n
    if name.startswith("truncated"):
        assert 0 < n < c["flows"].shape[0]  # the noise flows were cut off (voldor.cpp:187-194)
    assert np.array_equal(bits(o["depth"]), bits(gold[f"{name}/depth"])), f"depth differs at {np.mean(o['depth'] != gold[f'{name}/depth']):.4f} of the pixels"
    assert np.array_equal(bits(o["depth_conf"]), bits(gold[f"{name}/depth_conf"]))
    assert np.array_equal(bits(o["poses_covar"]), bits(gold[f"{name}/poses_covar"]))
    assert np.array_equal(bits(o["poses"]), bits(gold[f"{name}/poses"]))  # pose6() = Camera::rvec() of the float matrix: the oracle takes the same round trip (vk_ref_cv.h, round 4)


def test_b1_is_the_only_difference_of_the_default_exclusive_mode(gold, reference_mode):
    """Without ORC_EMULATE_B1 the oracle (D4: one normalised depth buffer) must NOT reproduce the reference's default-mode
    monocular window, and must reproduce the --exclusive_gpu_context 0 one: the stale device depth is real (SURVEY B-1)."""
    c = dict(CASES)["mono_default_b1"]
    o = run_oracle(c)
    assert o["n_registered"] == int(gold["mono_default_b1/n_registered"])
    assert np.mean(o["depth"] != gold["mono_default_b1/depth"]) > 0.5
    c2 = dict(c, config=dict(CASES)["mono_nonexclusive"]["config"])
    o2 = run_oracle(c2)
    assert np.array_equal(bits(o2["depth"]), bits(gold["mono_nonexclusive/depth"]))


@pytest.mark.parametrize("name", ["stereo_default", "mono_nonexclusive", "depth_priors", "truncated_b1", "cfg1_cpu_p3p"])
def test_default_oracle_within_sampling_noise_of_the_reference(gold, name):
    c = dict(CASES)[name]
    o = run_oracle(c)
    n = int(gold[f"{name}/n_registered"])
    assert o["n_registered"] == n
    rot, tr = synth.pose_errors(o["poses"], gold[f"{name}/poses"])
    # 128x96 / 160x120 windows: a few thousand correspondences per camera, mean-shift mode of 8192 re-drawn hypotheses
    assert rot.max() < 2e-3 and tr.max() < 8e-2, (rot, tr)
    m = (o["depth_conf"] > 0.5) & (gold[f"{name}/depth_conf"] > 0.5)
    s = np.mean(np.linalg.norm(o["poses"][:, 3:], axis=1)) / np.mean(np.linalg.norm(gold[f"{name}/poses"][:, 3:], axis=1))
    rel = np.abs(o["depth"][m] / s - gold[f"{name}/depth"][m]) / gold[f"{name}/depth"][m]
    assert m.mean() > 0.3 and np.median(rel) < 3e-2, (m.mean(), np.median(rel))


# ---- strict math on the reference's own code --------------------------------------------------------------------------------
STRICT_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window_strict.npz")


@pytest.mark.parametrize("name", ["mono_nonexclusive", "stereo_default", "stereo_ap3p", "depth_priors"])
def test_strict_oracle_bit_identical_to_the_reference_in_strict_math(reference_mode, name):
    """tests/golden/ref_window_strict.npz = the reference pipeline with its libm calls served by vk_strict_math.h (oracle/ref_stubs/emul/
    cuda_emul.h, ref_set_math_mode(1)).  The oracle with orc_set_strict_math(1) must give the same bits: "strict math" is nothing but
    a libm swap in the reference's own code, and the HIP path is held to this oracle bit for bit on the GPU (tests/test_gpu_strict.py)."""
    g = np.load(STRICT_GOLD)
    c = dict(CASES)[name]
    orc.lib().orc_set_strict_math(1)
    if orc.ref() is not None:
        orc.ref().ref_set_math_mode(1)  # the hooked rodrigues() is the reference's own code: its atan2f must be the strict one too
    try:
        o = run_oracle(c)
    finally:
        orc.lib().orc_set_strict_math(0)
        if orc.ref() is not None:
            orc.ref().ref_set_math_mode(0)
    assert o["n_registered"] == int(g[f"{name}/n_registered"])
    assert np.array_equal(bits(o["depth"]), bits(g[f"{name}/depth"]))
    assert np.array_equal(bits(o["depth_conf"]), bits(g[f"{name}/depth_conf"]))
    assert np.array_equal(bits(o["poses_covar"]), bits(g[f"{name}/poses_covar"]))
    assert np.array_equal(bits(o["poses"]), bits(g[f"{name}/poses"]))
    # and it is a different rounding of the same window than the glibc run
    assert not np.array_equal(bits(g[f"{name}/depth"]), bits(np.load(GOLD)[f"{name}/depth"]))


# ---- round 4: the last two stand-ins switched off -- cuRAND XORWOW streams and CUDA's texture filter (vk_ref_cuda.h) ------------------
CUDA_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window_cuda.npz")


@pytest.mark.parametrize("name", ["mono_nonexclusive", "stereo_default", "stereo_ap3p", "depth_priors"])
def test_oracle_with_xorwow_and_texture_filter_equals_the_reference(reference_mode, name):
    """tests/golden/ref_window_cuda.npz = the reference pipeline in strict math with ref_set_reference_rng(1) / ref_set_reference_tex(1):
    its curand_init / curand_uniform calls served by the XORWOW restatement, its at_tex by the 8-bit-fraction filter over the stacked
    layers (oracle/ref_stubs/emul/cuda_emul.h <- voldor_amd/csrc/vk_ref_cuda.h).  The oracle with orc_set_reference_rng(1) /
    orc_set_reference_tex(1) implements the same two rules at ITS call sites (per-pixel states that persist across the sample passes, the
    solver's re-seeded streams, every fetch of flows / priors / confidences): same bits in every output."""
    if not os.path.exists(CUDA_GOLD):
        pytest.skip("tests/golden/ref_window_cuda.npz not generated")
    g = np.load(CUDA_GOLD)
    c = dict(CASES)[name]
    L = orc.lib()
    L.orc_set_strict_math(1); L.orc_set_reference_rng(1); L.orc_set_reference_tex(1)
    if orc.ref() is not None:
        orc.ref().ref_set_math_mode(1)
    try:
        o = run_oracle(c)
    finally:
        L.orc_set_strict_math(0); L.orc_set_reference_rng(0); L.orc_set_reference_tex(0)
        if orc.ref() is not None:
            orc.ref().ref_set_math_mode(0)
    assert o["n_registered"] == int(g[f"{name}/n_registered"])
    assert np.array_equal(bits(o["depth"]), bits(g[f"{name}/depth"]))
    assert np.array_equal(bits(o["depth_conf"]), bits(g[f"{name}/depth_conf"]))
    assert np.array_equal(bits(o["poses_covar"]), bits(g[f"{name}/poses_covar"]))
    assert np.array_equal(bits(o["poses"]), bits(g[f"{name}/poses"]))
    # the switches matter: another window than the strict one with the stand-ins
    assert not np.array_equal(bits(g[f"{name}/depth"]), bits(np.load(STRICT_GOLD)[f"{name}/depth"]))


DEFAULT_GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_window_default.npz")


@pytest.mark.parametrize("name", ["mono_default_b1", "truncated_b1"])
def test_strict_oracle_reproduces_the_default_exclusive_mode(reference_mode, name):
    """tests/golden/ref_window_default.npz = the reference pipeline in strict math under its DEFAULT `exclusive_gpu_context 1`: the monocular
    runs that show SURVEY Appendix B-1.  The oracle with ORC_EMULATE_B1=1 gives the same bits (and the product with
    `--reference_stale_depth 1`: tests/test_gpu_vs_ref_window.py)."""
    if not os.path.exists(DEFAULT_GOLD):
        pytest.skip("tests/golden/ref_window_default.npz not generated")
    g = np.load(DEFAULT_GOLD)
    c = dict(CASES)[name]
    reference_mode.setenv("ORC_EMULATE_B1", "1")
    orc.lib().orc_set_strict_math(1)
    if orc.ref() is not None:
        orc.ref().ref_set_math_mode(1)
    try:
        o = run_oracle(c)
    finally:
        orc.lib().orc_set_strict_math(0)
        if orc.ref() is not None:
            orc.ref().ref_set_math_mode(0)
    assert o["n_registered"] == int(g[f"{name}/n_registered"])
    for k in ("depth", "depth_conf", "poses_covar"):
        assert np.array_equal(bits(o[k]), bits(g[f"{name}/{k}"])), k
    assert np.array_equal(bits(o["poses"]), bits(g[f"{name}/poses"]))

