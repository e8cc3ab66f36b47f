"""The fp32 pre-filter of the strict sample pass and of the strict table pass (vk_depth_impl.hpp "the strict sample pass behind an fp32 filter", round 6).

The filter discards a hypothesis only when a lower bound of its strict cost -- the filter's own sum less SF_REL of it and SF_ABS per unit weight --
is at or above the pixel's current cost, so it changes no result as long as the filter's -log(rigidness) stays within that margin of the strict
one for the same inputs.  Here: (1) the margin is MEASURED over the input range and held a factor 10 below SF_ABS / SF_REL, (2) windows with the
filter on and off are equal in every bit while the counters show that the filter did discard most of what it saw, and that strict arithmetic
did see the rest (reference: gpu-kernels/optimize_depth.cu:201-207 `cost < best`, :269-284 the sample loop, residual_model.h:34-49)."""
import numpy as np
import pytest

import hooks
from test_gpu_strict import assert_bits, strict  # noqa: F401  (fixture)

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("eight_point_bootstrap")]

SF_ABS = SF_REL = 1e-4  # vk_device.hpp
COUNTERS = ("sf_samples", "sf_sample_survivors", "sf_table_tiles", "sf_table_queued")


@pytest.mark.parametrize("arf,lam", [(1.0, 0.15), (2.0, 0.15), (0.5, 0.3), (1.0, 0.05)])
def test_filter_margin_is_ten_times_the_measured_deviation(arf, lam):
    rng = np.random.default_rng(int(arf * 10 + lam * 1000))
    n = 1 << 21
    worst = 0.0
    for regime in range(4):
        # observed flow: magnitude log-uniform from 1e-4 to 2000 px (and exactly zero); rigid flow = observed + an error of 1e-6 .. 3000 px
        mag = np.exp(rng.uniform(np.log(1e-4), np.log(2000.0), n)); ang = rng.uniform(0, 2 * np.pi, n)
        ox = (mag * np.cos(ang)).astype(np.float32); oy = (mag * np.sin(ang)).astype(np.float32)
        if regime == 1:
            ox[::7] = 0; oy[::7] = 0
        lo, hi = [(1e-6, 3000.0), (1e-3, 30.0), (1e-9, 1e-3), (10.0, 1e5)][regime]
        em = np.exp(rng.uniform(np.log(lo), np.log(hi), n)); ea = rng.uniform(0, 2 * np.pi, n)
        dx1 = (ox + em * np.cos(ea)).astype(np.float32); dy1 = (oy + em * np.sin(ea)).astype(np.float32)
        if regime == 2:
            dx1[::5] = ox[::5]; dy1[::5] = oy[::5]  # exactly no error
        s, f = hooks.filter_pair(dx1, dy1, ox, oy, lam, arf)
        ok = np.isfinite(s) & np.isfinite(f)
        assert ok.mean() > 0.95, ok.mean()  # (errors of 1e4 px and more against a small strictness term: the ratio leaves the fp32 range on both sides)
        assert (s[ok] >= 0).all() and (f[ok] >= 0).all()  # every term of the chain is >= 0: what the bound's monotonicity rests on
        dev = np.abs(s[ok].astype(np.float64) - f[ok]) / (SF_ABS + SF_REL * np.maximum(s[ok], f[ok]))
        worst = max(worst, float(dev.max()))
        bad = ~ok
        # where one side is not finite the other must not be small: the filter may only discard what strict arithmetic would not take either
        assert not (np.isfinite(f[bad]) & ~np.isfinite(s[bad]) & (f[bad] < 20)).any()
        assert not (np.isfinite(s[bad]) & ~np.isfinite(f[bad]) & (s[bad] < 20)).any()
    print(f"arf {arf} lambda {lam}: worst deviation = {worst:.4f} of the margin")
    assert worst <= 0.1, worst


@pytest.mark.parametrize("basefocal,omega,arf", [(386.1, 0.15, 1.0), (40.0, 0.15, 2.0), (480.0, 0.3, 1.0)])
def test_filter_margin_of_a_depth_prior_term(basefocal, omega, arf):
    rng = np.random.default_rng(int(basefocal))
    n = 1 << 21
    worst = 0.0
    for regime in range(3):
        d2 = np.exp(rng.uniform(np.log(0.05), np.log(5000.0), n)).astype(np.float32)  # the prior's depth
        rel = [np.exp(rng.uniform(np.log(1e-7), np.log(10.0), n)), np.exp(rng.uniform(np.log(1e-3), np.log(1.0), n)), np.zeros(n)][regime]
        d1 = (d2 * (1 + rng.choice([-1, 1], n) * np.minimum(rel, 0.999))).astype(np.float32)  # the hypothesis seen from the prior's camera
        s, f = hooks.filter_pair_depth(d1, d2, basefocal, omega, arf)
        ok = np.isfinite(s) & np.isfinite(f)
        assert ok.mean() > 0.95, ok.mean()
        assert (s[ok] >= 0).all() and (f[ok] >= 0).all()
        dev = np.abs(s[ok].astype(np.float64) - f[ok]) / (SF_ABS + SF_REL * np.maximum(s[ok], f[ok]))
        worst = max(worst, float(dev.max()))
        bad = ~ok
        assert not (np.isfinite(f[bad]) & ~np.isfinite(s[bad]) & (f[bad] < 20)).any()
        assert not (np.isfinite(s[bad]) & ~np.isfinite(f[bad]) & (s[bad] < 20)).any()
    print(f"basefocal {basefocal} omega {omega} arf {arf}: worst deviation = {worst:.4f} of the margin")
    assert worst <= 0.1, worst


def _windows():
    from voldor_amd import synth
    mono = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --strict_math 1"
    out = {}
    sc = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=11)
    out["mono_320x240"] = (sc, mono + " --max_iters 3", {})
    sc = synth.make_scene(w=323, h=241, n_flows=3, fx=160, fy=160, cx=160, cy=120, seed=21)
    out["odd_323x241"] = (sc, mono + " --max_iters 3", {})
    sc = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=16)
    rng = np.random.default_rng(2)
    h, w = sc["depth_gt"].shape
    kw = dict(depth_priors=(sc["depth_gt"][None] * (1 + rng.normal(0, 0.03, (2, h, w)))).astype(np.float32),
              depth_prior_poses=np.array([[0.002, -0.001, 0.0005, 0.01, 0.0, -0.02], [0, 0, 0, 0, 0, 0]], np.float32),
              depth_prior_pconfs=rng.uniform(0.3, 1.0, (2, h, w)).astype(np.float32))
    out["two_priors"] = (sc, "--silent --meanshift_kernel_var 0.1 --delta 0.5 --max_iters 3 --strict_math 1", kw)
    sc = synth.make_scene(w=1241, h=376, n_flows=8, fx=718.856, fy=718.856, cx=607.19, cy=185.22, seed=501, basefocal=386.1)
    out["wide_1241_disparity"] = (sc, "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 2 --strict_math 1", dict(basefocal=386.1, disparity=sc["disparity"]))
    sc = synth.make_scene(w=320, h=240, n_flows=5, fx=160, fy=160, cx=160, cy=120, seed=17)
    fl = sc["flows"].copy()
    fl[3:] = np.random.default_rng(4).uniform(-25, 25, fl[3:].shape).astype(np.float32)
    out["truncated"] = (dict(sc, flows=fl), mono + " --max_iters 5", {})
    sc = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=11)
    out["reference_rng_tex"] = (sc, mono + " --max_iters 3 --reference_draw 1 --reference_svd 1 --reference_rng 1 --reference_tex 1", {})
    return out


@pytest.mark.parametrize("name", ["mono_320x240", "odd_323x241", "two_priors", "wide_1241_disparity", "truncated", "reference_rng_tex"])
def test_filter_changes_no_bit_and_discards_most(strict, name):
    from voldor_amd import kernels, pyvoldor
    sc, cfg, kw = _windows()[name]
    fx, fy, cx, cy = sc["K"]
    out, seen = {}, {}
    try:
        for mode in (2, 0):
            hooks.debug_switch("strict_filter", mode); hooks.debug_switch("strict_table_filter", 2)  # (the table pass's filter is on from 1 M pixels by default: forced here)
            kernels.set_rand_epoch(0)
            for k in COUNTERS:
                hooks.debug_counter(k)
            out[mode] = pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=cfg, **kw)
            seen[mode] = {k: hooks.debug_counter(k) for k in COUNTERS}
    finally:
        hooks.debug_switch("strict_filter", 1); hooks.debug_switch("strict_table_filter", 1)
    a, b = out[2], out[0]
    assert a["n_registered"] == b["n_registered"]
    for k in ("depth", "depth_conf", "poses", "poses_covar"):
        assert_bits(a[k], b[k], f"{name}: {k}")
    s = seen[2]
    print(name, s)
    assert all(v == 0 for v in seen[0].values()), seen[0]  # filter off: the filtered kernels did not run
    assert s["sf_samples"] > 0
    assert 0 < s["sf_sample_survivors"] < (0.6 if name == "two_priors" else 0.05) * s["sf_samples"], s  # (two strong priors on a small scene: a third of the samples beat a poor incumbent)
    assert s["sf_table_tiles"] > 0 and 0 < s["sf_table_queued"] < (0.9 if name == "two_priors" else 0.7) * 256 * s["sf_table_tiles"], s  # (the accepts and the near-ties: strict arithmetic saw them, and only them)
