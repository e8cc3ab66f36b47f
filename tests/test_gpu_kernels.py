"""Parity of every FAST HIP stage against the CPU oracle (glibc), through the C-ABI (B-inner).

Bit-exact parity is the strict mode's job (tests/test_gpu_strict.py: every stage and whole windows equal the oracle bit for bit).
The fast kernels differ from the oracle by fp32 rounding: hardware v_log/v_exp/v_rcp instead of glibc powf/expf/logf and IEEE
division, the rigid chain folded into one projective map per frame (positions move by ~1e-5 px), fused multiply-adds.  A rigidness
is a steep function of a small end-point error, so a map agrees to ~1e-5 typically and to ~1e-3 at its worst pixel; the discrete
`cost < best` decisions of the depth search can flip on near-ties, so depth maps are compared on the fraction of pixels that agree
(>= 99 %).
"""
import numpy as np
import pytest

import hooks

from conftest import K9

pytestmark = pytest.mark.gpu


def _state(scene, rng, noise=0.1):
    from voldor_amd import synth
    flows = scene["flows"]
    N, h, w, _ = flows.shape
    gt = scene["poses_gt"]
    Rs = np.stack([synth.rodrigues(gt[i, :3]) for i in range(N)]).astype(np.float32)
    ts = gt[:, 3:].astype(np.float32)
    depth = (scene["depth_gt"] * (1 + rng.normal(0, noise, (h, w)))).astype(np.float32)
    rig = rng.uniform(0.2, 1.0, (N, h, w)).astype(np.float32)
    return flows, Rs, ts, depth, rig


def _od_kwargs(**over):
    kw = dict(abs_resize_factor=1.0, basefocal=0.0, n_rand_samples=10, global_prop_step=8, local_prop_width=32,
              lambda_=0.15, omega=0.15, disp_delta=-1.0, delta=0.5, fb_smooth=1, s0_ems_prob=0.5, no_change_prob=0.9,
              range_factor=1.0, update_rigidness_only=0)
    kw.update(over)
    return kw


def _run_both(orc, scene, K, flows, Rs, ts, depth, rig, priors=None, pconfs=None, confs=None, dp_Rs=None, dp_ts=None, epoch=5, **over):
    from voldor_amd import kernels
    N, h, w, _ = flows.shape
    N_dp = 0 if priors is None else priors.shape[0]
    kw = _od_kwargs(**over)
    p = orc.make_od_params(N, N_dp, w, h, K, Rs, ts, dp_Rs, dp_ts, **kw)
    o_depth, o_rig, o_confs, o_cost, o_ep = orc.optimize_depth(p, flows, rig, depth, priors, pconfs, confs, rand_epoch=epoch)
    kernels.set_rand_epoch(epoch)
    g_depth, g_rig, g_confs = kernels.optimize_depth_gpu(
        flows, rig, priors, pconfs, confs, depth, K, Rs, ts, dp_Rs, dp_ts, kw["abs_resize_factor"], N, N_dp, w, h,
        kw["basefocal"], kw["n_rand_samples"], kw["global_prop_step"], kw["local_prop_width"], kw["lambda_"], kw["omega"],
        kw["disp_delta"], kw["delta"], kw["fb_smooth"], kw["s0_ems_prob"], kw["no_change_prob"], kw["range_factor"],
        kw["update_rigidness_only"])
    assert kernels.get_rand_epoch() == o_ep
    return (o_depth, o_rig, o_confs), (g_depth, g_rig, g_confs)


def _assert_map_close(a, b, typical=2e-5, worst=3e-3):
    d = np.abs(a - b)
    assert np.percentile(d, 99) < 10 * typical and np.median(d) < typical and d.max() < worst, (np.median(d), np.percentile(d, 99), d.max())


def test_update_rigidness_only_matches_oracle(orc, small_scene):
    rng = np.random.default_rng(0)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = _state(small_scene, rng)
    (od, orig, _), (gd, grig, _) = _run_both(orc, small_scene, K, flows, Rs, ts, depth, rig, update_rigidness_only=1)
    np.testing.assert_array_equal(od, gd)  # depth untouched
    _assert_map_close(orig, grig)


def test_fb_smooth_and_cost_path(orc, small_scene):
    # no random samples, no propagation: fb_smooth -> cost map -> E-step only; depth must not move
    rng = np.random.default_rng(1)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = _state(small_scene, rng)
    (od, orig, _), (gd, grig, _) = _run_both(orc, small_scene, K, flows, Rs, ts, depth, rig, n_rand_samples=0,
                                             global_prop_step=0, local_prop_width=0)
    np.testing.assert_array_equal(od, gd)
    _assert_map_close(orig, grig)


@pytest.mark.parametrize("stage", ["rand", "global", "local", "all"])
def test_depth_search_stages(orc, small_scene, stage):
    rng = np.random.default_rng(2)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = _state(small_scene, rng, noise=0.3)
    over = dict(n_rand_samples=0, global_prop_step=0, local_prop_width=0, fb_smooth=0)
    if stage in ("rand", "all"): over["n_rand_samples"] = 10
    if stage in ("global", "all"): over["global_prop_step"] = 8
    if stage in ("local", "all"): over["local_prop_width"] = 32
    if stage == "all": over["fb_smooth"] = 1
    (od, orig, _), (gd, grig, _) = _run_both(orc, small_scene, K, flows, Rs, ts, depth, rig, **over)
    agree = np.mean(np.abs(od - gd) <= 1e-5 * np.abs(od))
    assert agree >= 0.99, f"{stage}: only {agree:.4f} of depth pixels agree"
    same = np.abs(od - gd) <= 1e-5 * np.abs(od)
    _assert_map_close(orig[:, same], grig[:, same])


def test_ragged_size_and_small_segments(orc):
    from voldor_amd import synth
    sc = synth.make_scene(w=83, h=67, n_flows=3, fx=40, fy=40, cx=41, cy=33, seed=11)  # not multiples of 64/32/8
    rng = np.random.default_rng(3)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=0.3)
    (od, orig, _), (gd, grig, _) = _run_both(orc, sc, K, flows, Rs, ts, depth, rig, global_prop_step=3, local_prop_width=7)
    agree = np.mean(np.abs(od - gd) <= 1e-5 * np.abs(od))
    assert agree >= 0.985


def test_global_step1_serial_chain(orc):
    from voldor_amd import synth
    sc = synth.make_scene(w=64, h=48, n_flows=2, fx=32, fy=32, cx=32, cy=24, seed=12)
    rng = np.random.default_rng(4)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=0.3)
    (od, _, _), (gd, _, _) = _run_both(orc, sc, K, flows, Rs, ts, depth, rig, global_prop_step=1, local_prop_width=0, n_rand_samples=2, fb_smooth=0)
    agree = np.mean(np.abs(od - gd) <= 1e-5 * np.abs(od))
    assert agree >= 0.97  # one flipped decision propagates along a serial chain


def _gpu_only(flows, Rs, ts, depth, rig, K, epoch=5, priors=None, pconfs=None, confs=None, dp_Rs=None, dp_ts=None, **over):
    from voldor_amd import kernels
    N, h, w, _ = flows.shape
    N_dp = 0 if priors is None else priors.shape[0]
    kw = _od_kwargs(**over)
    kernels.set_rand_epoch(epoch)
    return kernels.optimize_depth_gpu(flows, rig, priors, pconfs, confs, depth, K, Rs, ts, dp_Rs, dp_ts, kw["abs_resize_factor"], N, N_dp, w, h,
                                      kw["basefocal"], kw["n_rand_samples"], kw["global_prop_step"], kw["local_prop_width"], kw["lambda_"], kw["omega"],
                                      kw["disp_delta"], kw["delta"], kw["fb_smooth"], kw["s0_ems_prob"], kw["no_change_prob"], kw["range_factor"],
                                      kw["update_rigidness_only"])


@pytest.mark.parametrize("width,noise,n_flows,n_dp", [(32, 0.3, 4, 0), (32, 0.02, 5, 0), (33, 0.1, 5, 1), (7, 0.3, 3, 0), (64, 0.3, 9, 0),
                                                        (65, 0.05, 13, 2), (5, 0.0, 2, 0), (32, 0.1, 16, 0), (32, 0.2, 6, 5), (32, 0.2, 8, 0), (33, 0.1, 7, 1), (32, 0.2, 10, 0), (32, 0.1, 12, 1)])
def test_local_runs_equal_the_step_by_step_chain(width, noise, n_flows, n_dp):
    """The local-propagation kernel of the fast mode (cost table + chain automata whose run evaluations are planned two runs ahead, four
    lanes per pixel, two chains per wave up to width 33) against the literal step-by-step chain of the same arithmetic
    (vk_debug_switch "local_serial"): identical depth and rigidness maps, bit for bit, at replacement rates from ~50 % (noisy start) to ~0, for
    every way the frames and depth priors fall onto the four lanes of a pixel."""
    from voldor_amd import kernels, synth
    sc = synth.make_scene(w=211, h=97, n_flows=n_flows, fx=100, fy=100, cx=105, cy=48, seed=21, basefocal=40.0 if n_dp else 0.0)  # ragged: 211 = 6 * 32 + 19
    rng = np.random.default_rng(int(width * 100 + noise * 1000))
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=noise)
    if noise == 0.0:
        depth = sc["depth_gt"].astype(np.float32).copy()
    h, w = depth.shape
    extra = {}
    if n_dp:
        pri = np.stack([(sc["depth_gt"] * (1 + rng.normal(0, 0.05, (h, w)))).astype(np.float32) for _ in range(n_dp)])
        pri[:, ::7, ::5] = 0.0  # holes
        extra = dict(priors=pri, pconfs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32), confs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32),
                     dp_Rs=np.tile(np.eye(3, dtype=np.float32), (n_dp, 1, 1)), dp_ts=(rng.normal(0, 0.02, (n_dp, 3)) * np.arange(n_dp)[:, None]).astype(np.float32))
    over = dict(n_rand_samples=3, global_prop_step=5, local_prop_width=width, fb_smooth=0, basefocal=40.0 if n_dp else 0.0, disp_delta=1.0 if n_dp else -1.0)
    try:
        hooks.set_local_serial(True)
        d1, r1, c1 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    finally:
        hooks.set_local_serial(False)
    d2, r2, c2 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    assert np.mean(d1 != depth) > (0.05 if noise >= 0.1 else 0.0)  # the passes did replace depths
    np.testing.assert_array_equal(d1.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32))
    if n_dp:
        np.testing.assert_array_equal(c1.view(np.uint32), c2.view(np.uint32))


@pytest.mark.parametrize("noise,n_flows,n_dp,n_rand", [(0.3, 5, 0, 10), (0.02, 5, 0, 10), (0.2, 8, 1, 10), (0.1, 10, 1, 7), (0.3, 13, 2, 10), (0.0, 3, 0, 12), (0.2, 1, 0, 10), (0.2, 16, 0, 3)])
def test_sample_pass_equals_the_plain_sequential_form(noise, n_flows, n_dp, n_rand):
    """k_cost_rand_q -- exact early rejection after frame 0 and the priors, survivors compacted into an LDS queue, winner by a 64-bit
    atomicMin on (cost bits, sample index) -- against the plain loop over the samples in the same fast arithmetic
    (vk_debug_switch "cost_rand_plain"): IDENTICAL depth maps and rigidness maps, bit for bit, from noisy starts (most samples rejected late) to
    converged ones, with and without depth priors (identity and non-identity poses), sample counts that are not a multiple of the round."""
    from voldor_amd import kernels, synth
    sc = synth.make_scene(w=211, h=97, n_flows=n_flows, fx=100, fy=100, cx=105, cy=48, seed=29, basefocal=40.0 if n_dp else 0.0)
    rng = np.random.default_rng(int(n_flows * 100 + noise * 1000))
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=noise)
    if noise == 0.0:
        depth = sc["depth_gt"].astype(np.float32).copy()
    h, w = depth.shape
    extra = {}
    if n_dp:
        pri = np.stack([(sc["depth_gt"] * (1 + rng.normal(0, 0.05, (h, w)))).astype(np.float32) for _ in range(n_dp)])
        pri[:, ::7, ::5] = 0.0  # holes
        extra = dict(priors=pri, pconfs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32), confs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32),
                     dp_Rs=np.tile(np.eye(3, dtype=np.float32), (n_dp, 1, 1)), dp_ts=(rng.normal(0, 0.02, (n_dp, 3)) * np.arange(n_dp)[:, None]).astype(np.float32))
    over = dict(n_rand_samples=n_rand, global_prop_step=0, local_prop_width=0, fb_smooth=0, basefocal=40.0 if n_dp else 0.0, disp_delta=1.0 if n_dp else -1.0)
    try:
        hooks.set_cost_rand_plain(True)
        d1, r1, c1 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    finally:
        hooks.set_cost_rand_plain(False)
    d2, r2, c2 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    assert np.mean(d1 != depth) > (0.05 if noise >= 0.1 else 0.0)  # samples were accepted
    np.testing.assert_array_equal(d1.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32))


@pytest.mark.parametrize("step,noise,n_flows,n_dp", [(8, 0.3, 5, 0), (2, 0.3, 4, 0), (5, 0.1, 8, 1), (8, 0.2, 10, 0), (3, 0.2, 13, 2), (8, 0.05, 16, 0), (7, 0.3, 1, 0)])
def test_global_split_equals_one_lane_per_site(step, noise, n_flows, n_dp):
    """The global-propagation passes with a site evaluated by a group of lanes (k_global_prop_split_lean, the default) against one lane
    per site (vk_debug_switch "global_split"(0)): identical depth / rigidness maps, bit for bit, for every way the frames and depth priors fall
    onto the lanes of a group (4 lanes up to 8 frames, 8 beyond)."""
    from voldor_amd import kernels, synth
    sc = synth.make_scene(w=211, h=97, n_flows=n_flows, fx=100, fy=100, cx=105, cy=48, seed=23, basefocal=40.0 if n_dp else 0.0)
    rng = np.random.default_rng(int(step * 100 + noise * 1000))
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=noise)
    h, w = depth.shape
    extra = {}
    if n_dp:
        pri = np.stack([(sc["depth_gt"] * (1 + rng.normal(0, 0.05, (h, w)))).astype(np.float32) for _ in range(n_dp)])
        pri[:, ::7, ::5] = 0.0  # holes
        extra = dict(priors=pri, pconfs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32), confs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32),
                     dp_Rs=np.tile(np.eye(3, dtype=np.float32), (n_dp, 1, 1)), dp_ts=(rng.normal(0, 0.02, (n_dp, 3)) * np.arange(n_dp)[:, None]).astype(np.float32))
    over = dict(n_rand_samples=2, global_prop_step=step, local_prop_width=0, fb_smooth=0, basefocal=40.0 if n_dp else 0.0, disp_delta=1.0 if n_dp else -1.0)
    try:
        hooks.set_global_split(False)
        d1, r1, c1 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    finally:
        hooks.set_global_split(True)
    d2, r2, c2 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    assert np.mean(d1 != depth) > 0.01  # the passes did replace depths
    np.testing.assert_array_equal(d1.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32))


@pytest.mark.parametrize("n_flows,n_dp,w,h", [(5, 0, 211, 97), (1, 0, 64, 4), (8, 1, 211, 97), (10, 2, 130, 61), (16, 0, 211, 97), (3, 0, 63, 3)])
def test_estep_with_two_pixels_per_lane_equals_one(n_flows, n_dp, w, h):
    """vk_debug_switch "estep_pairs": the E-step with two pixels per lane on packed fp32 (k_update_rigidness_pairs) against one pixel per lane:
    identical rigidness maps, prior confidences and depth, bit for bit (the packed instructions round each half like the scalar ones; the
    wave sums cover the same rows in the same order), ragged tiles included."""
    from voldor_amd import synth
    sc = synth.make_scene(w=w, h=h, n_flows=n_flows, fx=100, fy=100, cx=w / 2, cy=h / 2, seed=29, basefocal=40.0 if n_dp else 0.0)
    rng = np.random.default_rng(n_flows * 10 + n_dp)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=0.2)
    extra = {}
    if n_dp:
        pri = np.stack([(sc["depth_gt"] * (1 + rng.normal(0, 0.05, (h, w)))).astype(np.float32) for _ in range(n_dp)])
        pri[:, ::7, ::5] = 0.0
        extra = dict(priors=pri, pconfs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32), confs=rng.uniform(0.3, 1.0, (n_dp, h, w)).astype(np.float32),
                     dp_Rs=np.tile(np.eye(3, dtype=np.float32), (n_dp, 1, 1)), dp_ts=(rng.normal(0, 0.02, (n_dp, 3)) * np.arange(n_dp)[:, None]).astype(np.float32))
    over = dict(update_rigidness_only=1, basefocal=40.0 if n_dp else 0.0, disp_delta=1.0 if n_dp else -1.0)
    try:
        hooks.debug_switch("estep_pairs", 2)
        d1, r1, c1 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
        hooks.debug_switch("estep_pairs", 0)
        d2, r2, c2 = _gpu_only(flows, Rs, ts, depth, rig, K, **extra, **over)
    finally:
        hooks.debug_switch("estep_pairs", 1)
    assert not np.array_equal(r2, rig)
    np.testing.assert_array_equal(d1.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32))
    if n_dp:
        np.testing.assert_array_equal(np.asarray(c1).view(np.uint32), np.asarray(c2).view(np.uint32))


def test_local_runs_with_the_tiled_table_equal_the_step_by_step_chain():
    """Above 400k pixels the candidate-cost table of a pass comes from its own tiled kernel (below, every chain tabulates its own steps at
    the head of the runs kernel): the same equality at 832x512."""
    from voldor_amd import kernels, synth
    sc = synth.make_scene(w=832, h=512, n_flows=3, fx=400, fy=400, cx=416, cy=256, seed=22)
    rng = np.random.default_rng(7)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=0.2)
    over = dict(n_rand_samples=2, global_prop_step=8, local_prop_width=32, fb_smooth=0)
    try:
        hooks.set_local_serial(True)
        d1, r1, _ = _gpu_only(flows, Rs, ts, depth, rig, K, **over)
    finally:
        hooks.set_local_serial(False)
    d2, r2, _ = _gpu_only(flows, Rs, ts, depth, rig, K, **over)
    assert np.mean(d1 != depth) > 0.05
    np.testing.assert_array_equal(d1.view(np.uint32), d2.view(np.uint32))
    np.testing.assert_array_equal(r1.view(np.uint32), r2.view(np.uint32))


def test_depth_priors_and_disparity(orc):
    from voldor_amd import synth
    sc = synth.make_scene(w=128, h=96, n_flows=3, fx=64, fy=64, cx=64, cy=48, seed=13, basefocal=30.0)
    rng = np.random.default_rng(5)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = _state(sc, rng, noise=0.2)
    h, w = depth.shape
    priors = np.stack([30.0 / sc["disparity"], sc["depth_gt"] * 1.02]).astype(np.float32)
    priors[1, :5, :] = 0  # invalid prior region (target_depth <= 0)
    pconfs = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32)
    confs = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32)
    dp_Rs = np.stack([np.eye(3), synth.rodrigues([0.002, -0.001, 0.001])]).astype(np.float32)
    dp_ts = np.array([[0, 0, 0], [0.01, 0.0, -0.05]], np.float32)
    (od, orig, ocf), (gd, grig, gcf) = _run_both(orc, sc, K, flows, Rs, ts, depth, rig, priors, pconfs, confs, dp_Rs, dp_ts,
                                                basefocal=30.0, disp_delta=1.0, delta=0.2)
    agree = np.mean(np.abs(od - gd) <= 1e-5 * np.abs(od))
    assert agree >= 0.99
    same = np.abs(od - gd) <= 1e-5 * np.abs(od)
    _assert_map_close(orig[:, same], grig[:, same])
    _assert_map_close(ocf[:, same], gcf[:, same])


def test_only_depth_priors_N0(orc):
    from voldor_amd import synth
    sc = synth.make_scene(w=96, h=64, n_flows=2, fx=48, fy=48, cx=48, cy=32, seed=14)
    rng = np.random.default_rng(6)
    K = K9(*sc["K"])
    h, w = sc["depth_gt"].shape
    priors = (sc["depth_gt"][None] * np.array([1.0, 1.05])[:, None, None]).astype(np.float32)
    pconfs = np.ones((2, h, w), np.float32)
    confs = np.ones((2, h, w), np.float32)
    dp_Rs = np.stack([np.eye(3)] * 2).astype(np.float32)
    dp_ts = np.zeros((2, 3), np.float32)
    flows = sc["flows"][:0]
    depth = priors[0].copy()
    rig = np.zeros((0, h, w), np.float32)
    (od, _, ocf), (gd, _, gcf) = _run_both(orc, sc, K, flows, None, None, depth, rig, priors, pconfs, confs, dp_Rs, dp_ts, delta=0.5,
                                           basefocal=24.0)  # basefocal 0 would make every prior cost tie exactly
    agree = np.mean(np.abs(od - gd) <= 1e-5 * np.abs(od))
    # two priors that agree to 5 %: flat cost valleys, i.e. many near-ties between candidates, decided by the last bits of the
    # projection (fast path: one projective map + v_rcp; oracle: un-fused chain + IEEE division).  Strict mode has no such slack
    # (tests/test_gpu_strict.py::test_strict_optimize_depth_with_priors_and_ragged_size_bits).
    assert agree >= 0.94
    same = np.abs(od - gd) <= 1e-5 * np.abs(od)
    _assert_map_close(ocf[:, same], gcf[:, same])


def test_null_protocol_reuses_device_copies(orc, small_scene):
    """Second call with NULL inputs must reuse what the first call left on the device
    (optimize_depth.cu:372-459), exactly like two chained oracle calls."""
    from voldor_amd import kernels
    rng = np.random.default_rng(7)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = _state(small_scene, rng, noise=0.3)
    N, h, w, _ = flows.shape
    kw = _od_kwargs()
    p = orc.make_od_params(N, 0, w, h, K, Rs, ts, **kw)
    d1, r1, _, _, ep = orc.optimize_depth(p, flows, rig, depth, rand_epoch=0)
    d2, r2, _, _, ep = orc.optimize_depth(p, flows, r1, d1, rand_epoch=ep)
    kernels.set_rand_epoch(0)
    args = (kw["abs_resize_factor"], N, 0, w, h, kw["basefocal"], kw["n_rand_samples"], kw["global_prop_step"], kw["local_prop_width"],
            kw["lambda_"], kw["omega"], kw["disp_delta"], kw["delta"], kw["fb_smooth"], kw["s0_ems_prob"], kw["no_change_prob"],
            kw["range_factor"], kw["update_rigidness_only"])
    kernels.optimize_depth_gpu(flows, rig, None, None, None, depth, K, Rs, ts, None, None, *args, download=False)
    g2, gr2, _ = kernels.optimize_depth_gpu(None, None, None, None, None, None, None, None, None, None, None, *args)
    agree = np.mean(np.abs(d2 - g2) <= 1e-5 * np.abs(d2))
    assert agree >= 0.98


def test_collect_and_compaction(orc, small_scene):
    from voldor_amd import kernels
    rng = np.random.default_rng(8)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = _state(small_scene, rng, noise=0.05)
    rig[:, 10:30, 20:60] = 0.1  # low rigidness region -> rejected
    depth[50:60, :] = 0.01      # below sample_min_depth
    N, h, w, _ = flows.shape
    for active in range(N):
        o2, o3 = orc.collect_p3p(flows, rig, depth, K, Rs, ts, active)
        g2, g3 = kernels.collect_p3p_instances(flows, rig, depth, K, Rs, ts, N, w, h, active)
        fin_o, fin_g = np.isfinite(o2[..., 0]), np.isfinite(g2[..., 0])
        assert np.mean(fin_o != fin_g) < 1e-3
        both = fin_o & fin_g
        assert both.sum() > 1000
        assert np.abs(o2[both] - g2[both]).max() < 2e-3
        assert np.abs(o3[both] - g3[both]).max() < 1e-4 * max(1.0, np.abs(o3[both]).max())
        # ordered compaction == row-major host scan (geometry.cpp:68-80)
        c2, c3 = kernels.get_compacted_points(w * h)
        assert c2.shape[0] == fin_g.sum()
        np.testing.assert_array_equal(c2, g2[fin_g])
        np.testing.assert_array_equal(c3, g3[fin_g])


def _corr(orc, scene, active=1):
    rng = np.random.default_rng(9)
    K = K9(*scene["K"])
    flows, Rs, ts, depth, rig = _state(scene, rng, noise=0.02)
    rig[:] = 1.0
    p2, p3 = orc.collect_p3p(flows, rig, depth, K, Rs, ts, active)
    return orc.compact_p3p(p2, p3) + (K,)


@pytest.mark.parametrize("solver", ["lambdatwist", "ap3p", "lambdatwist_f64"])
def test_pose_sampling(orc, small_scene, solver):
    from voldor_amd import kernels
    pts2, pts3, K = _corr(orc, small_scene)
    n = 4096
    if solver == "lambdatwist":
        orv, otv = orc.solve_batch_p3p(pts3, pts2, K, n)
        grv, gtv = kernels.solve_batch_p3p_lambdatwist_gpu(pts3, pts2, K, n)
    elif solver == "lambdatwist_f64":
        orv, otv = orc.solve_batch_p3p(pts3, pts2, K, n, use_double=True)
        grv, gtv = kernels.solve_batch_p3p_lambdatwist_f64_gpu(pts3, pts2, K, n)
    else:
        orv, otv = orc.solve_batch_p3p(pts3, pts2, K, n, use_ap3p=True)
        grv, gtv = kernels.solve_batch_p3p_ap3p_gpu(pts3, pts2, K, n)
    fo = np.isfinite(orv.sum(1) + otv.sum(1))
    fg = np.isfinite(grv.sum(1) + gtv.sum(1))
    assert fo.mean() > 0.5
    assert np.mean(fo != fg) < 0.02
    both = fo & fg
    err = np.maximum(np.abs(orv - grv).max(1), np.abs(otv - gtv).max(1))[both]
    # minimal solvers are ill-conditioned on some 4-tuples: compare the bulk, not the tail
    tol = 1e-6 if solver == "lambdatwist_f64" else 2e-3
    assert np.mean(err < tol) > (0.99 if solver == "lambdatwist_f64" else 0.93), np.percentile(err, [50, 90, 99])


def test_meanshift_and_robust_gaussian(orc, small_scene):
    from voldor_amd import kernels
    pts2, pts3, K = _corr(orc, small_scene)
    rv, tv = orc.solve_batch_p3p(pts3, pts2, K, 8192)
    fin = np.isfinite(rv.sum(1) + tv.sum(1))
    pool = np.concatenate([rv[fin] * 25.0, tv[fin]], 1).astype(np.float32)
    init = np.zeros(6, np.float32)
    for ext in (True, False):
        om, oc, oi = orc.meanshift(pool, 0.2, init, ext)
        gm, gc, gi = kernels.meanshift_gpu(pool, 0.2, init, ext)
        assert np.abs(om - gm).max() < 2e-4
        assert abs(oc - gc) < 1e-4 * max(1, oc)
        assert abs(oi - gi) <= 2
    cov0 = (np.eye(6) * 0.2 * 1e4).astype(np.float32)
    orc_rc, omean, ocov, odens, oit = orc.fit_robust_gaussian(pool * 100, om * 100, cov0)
    rc, gmean, gcov, gdens, git = kernels.fit_robust_gaussian(pool * 100, om * 100, cov0)
    assert rc == orc_rc == 0
    assert abs(odens - gdens) < 2e-3
    assert np.abs(omean - gmean).max() < 5e-2  # pose x100 units
    assert np.abs(ocov - gcov).max() < 2e-2 * np.abs(ocov).max()


def _pose_pool(orc, scene, active, n=8192):
    pts2, pts3, K = _corr(orc, scene, active)
    rv, tv = orc.solve_batch_p3p(pts3, pts2, K, n)
    return rv, tv


@pytest.mark.parametrize("active,n", [(1, 8192), (0, 8192), (2, 3000)])
def test_pipeline_mode_kernel_matches_oracle(orc, small_scene, active, n):
    """vk_pose_mode_pool: the mode kernel of the WINDOW PIPELINE (k_mode_trials + k_pose_mode<REFIT,512>: packed pairs in registers,
    Cholesky-whitened gate, pool re-dealt by distance) against the oracle's meanshift + fit_robust_gaussian glued as
    geometry.cpp:156-263 does -- the stage test of the kernels the timed path runs (the host-pointer meanshift_gpu /
    fit_robust_gaussian are other kernels).  Tolerances as for those: the hard 3-sigma gate makes the refit sensitive to last bits."""
    from voldor_amd import kernels
    rv, tv = _pose_pool(orc, small_scene, active, n)
    rs, var, sc = 25.0, 0.2, 100.0
    fin = np.isfinite(rv.sum(1) + tv.sum(1))
    assert fin.sum() > 0.5 * n
    pool = np.concatenate([rv[fin] * rs, tv[fin]], 1).astype(np.float32)
    init = np.zeros(6, np.float32)
    for ext in (True, False):
        om, oc, oi = orc.meanshift(pool, var, init, ext)
        g = hooks.pose_mode_pool(rv, tv, init, use_external_init_mean=ext, refit=False, kernel_var=var, rvec_scale=rs)
        assert g["success"] == 1 and g["sample_count"] == fin.sum()
        gm = np.concatenate([g["pose6"][:3] * rs, g["pose6"][3:]])
        assert np.abs(om - gm).max() < 2e-4, (ext, om, gm)
        assert abs(oc - g["density"]) < 1e-4 * max(1, oc) and abs(oi - g["ms_iters"]) <= 2
        assert not g["covar"].any()
    cov0 = (np.eye(6) * var * sc * sc).astype(np.float32)
    orc_rc, omean, ocov, odens, oit = orc.fit_robust_gaussian(pool * sc, om * sc, cov0)
    g = hooks.pose_mode_pool(rv, tv, init, use_external_init_mean=False, refit=True, kernel_var=var, rvec_scale=rs, rg_pose_scaling=sc)
    assert orc_rc == 0 and g["success"] == 1
    gmean = np.concatenate([g["pose6"][:3] * rs, g["pose6"][3:]]) * sc
    unit = np.array([rs] * 3 + [1.0] * 3)
    gcov = g["covar"] * sc * sc * unit[:, None] * unit[None, :]  # geometry.cpp:224-233 undone
    assert abs(odens - g["density"]) < 2e-3
    assert np.abs(omean - gmean).max() < 5e-2  # pose x100 units
    assert np.abs(ocov - gcov).max() < 2e-2 * np.abs(ocov).max()
    assert abs(oit - g["gu_iters"]) <= 3, (oit, g["gu_iters"])


@pytest.mark.parametrize("active,n", [(1, 8192), (2, 8192), (0, 5000)])
def test_refit_partition_changes_no_sum(orc, small_scene, active, n):
    """vk_debug_switch "refit_partition": with the pool re-dealt by distance a gate pass leaves out samples whose weight is exactly 0, so every sum has
    the same terms in another order.  Both forms must therefore agree to the rounding of a 28-value float reduction over <= 8192 terms
    (the gate is hard: a sample within that rounding of the 3-sigma surface may fall on the other side, which moves the fit by one
    sample in a few thousand)."""
    from voldor_amd import kernels
    rv, tv = _pose_pool(orc, small_scene, active, n)
    init = np.zeros(6, np.float32)
    out = {}
    try:
        for part in (1, 0):
            hooks.set_refit_partition(bool(part))
            out[part] = hooks.pose_mode_pool(rv, tv, init, use_external_init_mean=False, refit=True, kernel_var=0.2, rvec_scale=25.0)
            again = hooks.pose_mode_pool(rv, tv, init, use_external_init_mean=False, refit=True, kernel_var=0.2, rvec_scale=25.0)
            for k in ("pose6", "covar"):
                np.testing.assert_array_equal(out[part][k], again[k])  # either form is the same from run to run
    finally:
        hooks.set_refit_partition(True)
    a, b = out[1], out[0]
    assert a["success"] == b["success"] == 1 and a["sample_count"] == b["sample_count"] and a["ms_iters"] == b["ms_iters"]
    # measured on these pools: pose 1.5e-8, covariance 3e-7 relative, the same iteration count and density
    assert np.abs(a["pose6"] - b["pose6"]).max() < 1e-6
    assert np.abs(a["covar"] - b["covar"]).max() < 2e-5 * np.abs(b["covar"]).max()
    assert abs(a["density"] - b["density"]) < 1e-5 and abs(a["gu_iters"] - b["gu_iters"]) <= 1


def test_robust_gaussian_rejects_degenerate(orc):
    from voldor_amd import kernels
    rng = np.random.default_rng(10)
    pool = np.zeros((2048, 6), np.float32)
    pool[:, 0] = rng.normal(0, 1, 2048)  # rank-1 cloud -> singular covariance
    cov0 = np.zeros((6, 6), np.float32)
    rc_o = orc.fit_robust_gaussian(pool, np.zeros(6, np.float32), cov0)[0]
    rc_g = kernels.fit_robust_gaussian(pool, np.zeros(6, np.float32), cov0)[0]
    assert rc_o != 0 and rc_g != 0


def test_meanshift_generic_dims(orc):
    from voldor_amd import kernels
    rng = np.random.default_rng(11)
    space = np.concatenate([rng.normal(0.3, 0.05, (3000, 3)), rng.uniform(-2, 2, (3000, 3))]).astype(np.float32)
    om, oc, oi = orc.meanshift(space, 0.05, np.zeros(3, np.float32), True)
    gm, gc, gi = kernels.meanshift_gpu(space, 0.05, np.zeros(3, np.float32), True)
    assert np.abs(om - gm).max() < 1e-4 and np.abs(gm - 0.3).max() < 0.02


def test_gblur(orc):
    from voldor_amd import kernels
    rng = np.random.default_rng(12)
    src = rng.uniform(0, 1, (2, 37, 53)).astype(np.float32)
    for sigma, ks in ((1.5, 0), (3.0, 9)):
        rc_o, o = orc.gblur(src, sigma, ks)
        rc_g, g = kernels.gblur_gpu(src, sigma, ks)
        assert rc_o == 0 and rc_g == 0
        assert np.abs(o - g).max() < 1e-5
    assert kernels.gblur_gpu(src, 100.0, 0)[0] != 0  # half kernel > 128 taps -> error like the reference


def test_bootstrap_pieces(orc, small_scene):
    from voldor_amd import kernels
    K = K9(*small_scene["K"])
    ok_o, Ro, to = orc.estimate_pose_epipolar(small_scene["flows"][0], K)
    ok_g, Rg, tg = kernels.estimate_pose_epipolar(small_scene["flows"][0], K)
    assert ok_o and ok_g
    assert np.abs(Ro - Rg).max() < 1e-5 and np.abs(to - tg).max() < 1e-5
    Rb, tb, db = kernels.bootstrap_gpu(small_scene["flows"][0], K)  # the kernels the window pipeline runs
    np.testing.assert_array_equal(Rb, Rg)  # same source, fp64, no contraction: identical bits host vs device
    np.testing.assert_array_equal(tb, tg)
    do = orc.estimate_depth_closed_form(small_scene["flows"][0], K, Ro, to)
    np.testing.assert_array_equal(db, do)
    dg = kernels.estimate_depth_closed_form(small_scene["flows"][0], K, Ro, to)
    assert np.mean(np.abs(do - dg) <= 1e-3 * np.abs(do)) > 0.99


@pytest.mark.parametrize("w,h", [(160, 120), (161, 123), (64, 16), (37, 15), (640, 480)])
def test_fb_smooth_alone_matches_oracle(orc, w, h):
    """fb_smooth.h:72-108 on its own, sizes that exercise the 16-step register batches, their tails and the
    unaligned-row path.  The HIP recurrence is the Moebius regrouping of the same step with
    v_rcp_f32: a few ulp per step, contractive."""
    from voldor_amd import kernels
    rng = np.random.default_rng(w * 1000 + h)
    maps = rng.uniform(0.02, 0.98, (3, h, w)).astype(np.float32)
    maps[1, :, : w // 2] = 0.999
    maps[2, h // 3:, :] = 1e-3
    o = orc.fb_smooth(maps, 0.5, 0.9)
    rc, g = kernels.fb_smooth_gpu(maps, 0.5, 0.9)
    assert rc == 0
    assert np.isfinite(g).all()
    assert np.abs(o - g).max() < 2e-5


@pytest.mark.parametrize("w,h", [(640, 480), (1241, 376), (333, 777)])
def test_fb_smooth_segment_lengths_agree(orc, w, h):
    """vk_debug_switch "fb_segment": the 20- and 40-step segmentations of the fast fb_smooth (chosen by image size, fb_smooth_device) are the same
    recurrence cut differently; each stays within the stage tolerance of the oracle and they agree with each other to rounding."""
    from voldor_amd import kernels
    rng = np.random.default_rng(w + 7 * h)
    maps = rng.uniform(0.02, 0.98, (2, h, w)).astype(np.float32)
    maps[1, :, w // 3:] = 0.9999
    o = orc.fb_smooth(maps, 0.5, 0.9)
    out = {}
    try:
        for seg in (20, 40):
            hooks.set_fb_segment(seg)
            rc, out[seg] = kernels.fb_smooth_gpu(maps, 0.5, 0.9)
            assert rc == 0 and np.isfinite(out[seg]).all()
            assert np.abs(o - out[seg]).max() < 2e-5, seg
    finally:
        hooks.set_fb_segment(0)
    assert np.abs(out[20] - out[40]).max() < 1e-5
    assert not np.array_equal(out[20], out[40]) or w * h < 1000  # the switch really changes the launch
    with pytest.raises(Exception):
        hooks.set_fb_segment(30)


@pytest.mark.parametrize("with_priors,n_rand", [(False, 10), (True, 10), (False, 23), (False, 3), (False, 0)])
def test_sample_pass_with_survivor_queue_matches_strict(small_scene, with_priors, n_rand):
    """The fast cost-map + random-sample pass (k_cost_rand_q: exact early rejection after frame 0 / the priors, survivors compacted
    into an LDS queue, winner per pixel by a 64-bit atomicMin on (cost, sample index)) against the strict kernel, which walks the
    samples one by one like optimize_depth.cu:269-277: the same depth on all but the near-ties that fp32 rounding decides
    (measured: 4-6 pixels in 100 000) -- also when the samples do not fill a round of the queue (3) or need several (23) -- and the
    same result on every run (the queue order is whatever the LDS atomics make it)."""
    from voldor_amd import kernels
    rng = np.random.default_rng(21)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = _state(small_scene, rng, noise=0.3)
    N, h, w, _ = flows.shape
    pri = pc = cf = dR = dt = None
    if with_priors:
        pri = np.stack([small_scene["depth_gt"] * 1.03, small_scene["depth_gt"] * 0.98]).astype(np.float32)
        pri[1, :7] = 0
        pc = rng.uniform(0.5, 1, pri.shape).astype(np.float32); cf = rng.uniform(0.5, 1, pri.shape).astype(np.float32)
        dR = np.stack([np.eye(3), np.eye(3)]).astype(np.float32); dt = np.array([[0, 0, 0], [0.01, 0, -0.02]], np.float32)
    kw = _od_kwargs(n_rand_samples=n_rand, basefocal=40.0 if with_priors else 0.0, disp_delta=1.0 if with_priors else -1.0, global_prop_step=0,
                    local_prop_width=0, fb_smooth=0)
    out = []
    try:
        for strict in (True, False, False):
            kernels.set_strict_math(strict)
            kernels.set_rand_epoch(9)
            out.append(kernels.optimize_depth_gpu(flows, rig, pri, pc, cf, depth, K, Rs, ts, dR, dt, kw["abs_resize_factor"], N, 0 if pri is None else 2, w, h,
                                                  kw["basefocal"], kw["n_rand_samples"], kw["global_prop_step"], kw["local_prop_width"], kw["lambda_"],
                                                  kw["omega"], kw["disp_delta"], kw["delta"], kw["fb_smooth"], kw["s0_ems_prob"], kw["no_change_prob"],
                                                  kw["range_factor"], kw["update_rigidness_only"]))
    finally:
        kernels.set_strict_math(False)
    (sd, sr, _), (fd, fr, _), (fd2, fr2, _) = out
    np.testing.assert_array_equal(fd, fd2); np.testing.assert_array_equal(fr, fr2)
    if n_rand > 0:
        assert np.mean(sd != depth) > 0.05  # the samples do replace depths
    assert np.mean(fd != sd) <= 5e-4, np.mean(fd != sd)
    same = fd == sd
    _assert_map_close(sr[:, same], fr[:, same])
