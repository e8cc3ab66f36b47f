"""Strict-math mode: the HIP path reproduces the CPU oracle BIT FOR BIT -- stages and whole windows, up to BASELINE cfg2.

Chain of evidence: the oracle reproduces the reference's own code bit for bit (tests/test_oracle_vs_ref_*.py);
strict mode swaps the transcendental library calls for voldor_amd/csrc/vk_strict_math.h on both sides (oracle: orc_set_strict_math,
product: --strict_math 1) and runs every HIP stage in the reference's operation order (vk_strict.hip); `--reference_draw 1` /
ORC_REFERENCE_DRAW=1 select the reference's index draw (geometry.cpp:68-88 + solve_batch_lambdatwist.cu:16-19) on both sides.
What is asserted here is EQUALITY OF BITS, not a tolerance."""
import os

import numpy as np
import pytest

from conftest import K9
import test_gpu_kernels as tk

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("eight_point_bootstrap")]  # (fast windows here are held against the oracle / the reference ensembles: same two-view pose, conftest.py)


def _same_bits(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def assert_bits(a, b, what):
    eq = _same_bits(a, b)
    assert eq.all(), f"{what}: {(~eq).sum()} of {eq.size} values differ"


@pytest.fixture()
def strict(orc):
    from voldor_amd import kernels
    L = orc.lib()
    L.orc_set_strict_math(1); kernels.set_strict_math(True)
    old = os.environ.get("ORC_REFERENCE_DRAW")
    os.environ["ORC_REFERENCE_DRAW"] = "1"
    yield
    L.orc_set_strict_math(0); kernels.set_strict_math(False)
    if old is None:
        os.environ.pop("ORC_REFERENCE_DRAW", None)
    else:
        os.environ["ORC_REFERENCE_DRAW"] = old


def test_device_arithmetic_gives_the_host_bits():
    """IEEE + - * / sqrt, conversions and every vk_strict_math.h function: gfx950 == host, on 2^18 inputs each."""
    import hooks
    rng = np.random.default_rng(1)
    n = 1 << 18
    a = (rng.choice([-1, 1], n) * np.exp(rng.uniform(-12, 12, n))).astype(np.float32)
    b = (rng.choice([-1, 1], n) * np.exp(rng.uniform(-6, 6, n))).astype(np.float32)
    a2 = rng.normal(0, 3, n).astype(np.float32); b2 = rng.normal(0, 3, n).astype(np.float32)
    for op in range(hooks.lib().vkt_probe_ops_count()):
        for x, y in ((a, b), (a2, b2)):
            if op == 14:
                x = np.clip(x, -1e9, 1e9)
            h = hooks.probe(op, x, y, False); d = hooks.probe(op, x, y, True)
            neq = (h.view(np.uint64) != d.view(np.uint64)) & ~(np.isnan(h) & np.isnan(d))
            assert not neq.any(), (op, int(neq.sum()))


def test_strict_fb_smooth_bits(orc, strict):
    from voldor_amd import kernels
    rng = np.random.default_rng(3)
    for (n, h, w) in ((3, 120, 160), (1, 37, 101), (2, 240, 41)):
        m = rng.uniform(0.02, 0.98, (n, h, w)).astype(np.float32)
        rc, g = kernels.fb_smooth_gpu(m.copy())
        assert rc == 0
        assert_bits(orc.fb_smooth(m.copy()), g, f"fb_smooth {n}x{h}x{w}")


@pytest.mark.parametrize("case", ["rigidness_only", "cost_rand", "global", "local", "full", "full_no_smooth_step4_width7"])
def test_strict_optimize_depth_bits(orc, small_scene, strict, case):
    over = {"rigidness_only": dict(update_rigidness_only=1), "cost_rand": dict(global_prop_step=0, local_prop_width=0, fb_smooth=0),
            "global": dict(n_rand_samples=0, local_prop_width=0, fb_smooth=0), "local": dict(n_rand_samples=0, global_prop_step=0, fb_smooth=0),
            "full": dict(), "full_no_smooth_step4_width7": dict(fb_smooth=0, global_prop_step=4, local_prop_width=7, n_rand_samples=3)}[case]
    rng = np.random.default_rng(0)
    K = K9(*small_scene["K"])
    flows, Rs, ts, depth, rig = tk._state(small_scene, rng)
    (od, orig, _), (gd, grig, _) = tk._run_both(orc, small_scene, K, flows, Rs, ts, depth, rig, **over)
    assert_bits(od, gd, "depth"); assert_bits(orig, grig, "rigidness")


def test_strict_optimize_depth_with_priors_and_ragged_size_bits(orc, strict):
    from voldor_amd import synth
    sc = synth.make_scene(w=157, h=93, n_flows=6, fx=80, fy=75, cx=77, cy=45, seed=21)
    rng = np.random.default_rng(5)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = tk._state(sc, rng)
    N, h, w, _ = flows.shape
    pri = (sc["depth_gt"][None] * (1 + rng.normal(0, 0.05, (2, h, w)))).astype(np.float32)
    pri[1, :5] = 0.0  # invalid prior pixels: confidence left untouched (optimize_depth.cu:129)
    pc = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32); cf = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32)
    dpR = np.stack([np.eye(3), synth.rodrigues([0.01, -0.02, 0.005])]).astype(np.float32)
    dpt = np.array([[0, 0, 0], [0.05, 0.01, -0.1]], np.float32)
    (od, orig, oc), (gd, grig, gc) = tk._run_both(orc, sc, K, flows, Rs, ts, depth, rig, priors=pri, pconfs=pc, confs=cf, dp_Rs=dpR, dp_ts=dpt,
                                                   basefocal=40.0, disp_delta=1.0)
    assert_bits(od, gd, "depth"); assert_bits(orig, grig, "rigidness"); assert_bits(oc, gc, "prior confidences")


@pytest.mark.parametrize("solver", ["lambdatwist", "ap3p", "lambdatwist_f64"])
def test_strict_pose_hypotheses_bits(orc, small_scene, strict, solver):
    """index draw + minimal solver + 4th-point selection + nearest rotation + angle-axis: rotation vectors AND translations"""
    from voldor_amd import kernels
    pts2, pts3, K = tk._corr(orc, small_scene)
    kw, fn = {"lambdatwist": ({}, kernels.solve_batch_p3p_lambdatwist_gpu), "ap3p": (dict(use_ap3p=True), kernels.solve_batch_p3p_ap3p_gpu),
              "lambdatwist_f64": (dict(use_double=True), kernels.solve_batch_p3p_lambdatwist_f64_gpu)}[solver]
    orv, otv = orc.solve_batch_p3p(pts3, pts2, K, 8192, **kw)
    grv, gtv = fn(pts3, pts2, K, 8192)
    assert np.isfinite(otv.sum(1)).mean() > 0.5
    assert_bits(orv, grv, "rvecs"); assert_bits(otv, gtv, "tvecs")


def _window_both(orc, sc, cfg, **extra):
    from voldor_amd import kernels, pyvoldor
    fx, fy, cx, cy = sc["K"]
    o = orc.voldor(sc["flows"], fx, fy, cx, cy, config=cfg, **extra)
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=cfg + " --strict_math 1 --reference_draw 1", **extra)
    return o, g


def _assert_window_bits(o, g):
    assert o["n_registered"] == g["n_registered"]
    n = o["n_registered"]
    assert_bits(o["poses"][:n], g["poses"][:n], "poses")
    assert_bits(o["poses_covar"][:n], g["poses_covar"][:n], "pose covariances")
    assert_bits(o["depth"], g["depth"], "depth")
    assert_bits(o["depth_conf"], g["depth_conf"], "depth confidence")


WINDOWS = {
    "mono_1iter_norefit": (dict(seed=11), "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 1 --rg_refine 0"),
    "mono_3iters_refit": (dict(seed=11), "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 3"),
    "mono_ap3p": (dict(seed=13), "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2 --lambdatwist 0"),
    "mono_cpu_p3p_f64": (dict(seed=14), "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2 --cpu_p3p 1"),
    "mono_refit_every_iteration": (dict(seed=15), "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 3 --rg_refine_last_only 0"),
    "stereo": (dict(seed=12, basefocal=80.0), "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 3"),
}


@pytest.mark.parametrize("name", sorted(WINDOWS))
def test_strict_window_bits(orc, strict, name):
    from voldor_amd import synth
    kw, cfg = WINDOWS[name]
    sc = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, **kw)
    extra = dict(basefocal=kw["basefocal"], disparity=sc["disparity"]) if "basefocal" in kw else {}
    o, g = _window_both(orc, sc, cfg, **extra)
    assert o["n_registered"] == 4
    _assert_window_bits(o, g)


def test_strict_window_with_depth_priors_bits(orc, strict):
    """the SLAM driver's call shape (voldor_slam.py:447-457): depth priors with poses and confidences from earlier windows"""
    from voldor_amd import synth
    sc = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=16)
    rng = np.random.default_rng(2)
    h, w = sc["depth_gt"].shape
    pri = (sc["depth_gt"][None] * (1 + rng.normal(0, 0.03, (2, h, w)))).astype(np.float32)
    poses = np.array([[0.002, -0.001, 0.0005, 0.01, 0.0, -0.02], [0, 0, 0, 0, 0, 0]], np.float32)
    pconf = rng.uniform(0.3, 1.0, (2, h, w)).astype(np.float32)
    o, g = _window_both(orc, sc, "--silent --meanshift_kernel_var 0.1 --delta 0.5 --max_iters 3", depth_priors=pri, depth_prior_poses=poses,
                        depth_prior_pconfs=pconf)
    _assert_window_bits(o, g)


def test_strict_window_truncation_bits(orc, strict):
    """a window that loses its last frames (voldor.cpp:187-194): same truncation point, same outputs"""
    from voldor_amd import synth
    sc = synth.make_scene(w=320, h=240, n_flows=5, fx=160, fy=160, cx=160, cy=120, seed=17)
    fl = sc["flows"].copy()
    rng = np.random.default_rng(4)
    fl[3:] = rng.uniform(-25, 25, fl[3:].shape).astype(np.float32)  # frames 3.. carry no rigid motion at all
    sc = dict(sc, flows=fl)
    o, g = _window_both(orc, sc, "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 5")
    assert o["n_registered"] < 5
    _assert_window_bits(o, g)


def test_strict_baseline_cfg2_window_bits(orc, strict):
    """BASELINE cfg2 (640x480, N=5, 8 EM iterations, refit on): every output of the window, bit for bit"""
    from voldor_amd import synth
    sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320, fy=320, cx=320, cy=240, seed=233)
    o, g = _window_both(orc, sc, "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8")
    assert o["n_registered"] == 5
    _assert_window_bits(o, g)


# ---- round 4: strict arithmetic on the parallel launch structures == strict arithmetic on the plain ones -------------------------
def _plain_vs_parallel(run):
    """run() on the parallel structures (default: the mode kernel as 16 cooperating single-wave workgroups), once more with the mode kernel as
    ONE 512-thread workgroup (vk_debug_switch "strict_pose_coop" = 0), and once with vk_debug_switch "strict_plain" = 1 (rounds 1-3: one lane
    per chain / line / site, every sample in full, one 256-thread workgroup walking the sum tree block by block).  Returns (default, plain)
    after asserting that the single-workgroup form gives what the cooperative one gives."""
    import hooks
    from voldor_amd import kernels
    out = {}
    try:
        # "gave_up" (round 5): the cooperative form with a poll bound of ONE -- its workgroups give up their meetings almost at once, nothing of the
        # camera record is written and the single-workgroup kernel launched behind it takes the camera over (vk_strict.hip coop_wait): same bits
        for key, (plain, coop, polls) in {"coop": (0, 1, 0), "one_wg": (0, 0, 0), "gave_up": (0, 1, 1), "plain": (1, 1, 0)}.items():
            hooks.set_strict_plain(plain); hooks.set_strict_pose_coop(coop); hooks.set_strict_coop_max_polls(polls)
            kernels.set_rand_epoch(0)
            hooks.debug_counter("strict_coop_fallbacks")
            out[key] = run()
            _FALLBACKS[key] = _FALLBACKS.get(key, 0) + hooks.debug_counter("strict_coop_fallbacks")
    finally:
        hooks.set_strict_plain(0); hooks.set_strict_pose_coop(1); hooks.set_strict_coop_max_polls(0)
    a = out["coop"]
    for other in ("one_wg", "gave_up"):
        b = out[other]
        for k in a:
            if isinstance(a[k], np.ndarray):
                assert_bits(a[k], b[k], f"cooperative vs {other} mode kernel: {k}")
            elif isinstance(a[k], (int, float, np.integer, np.floating)):
                assert np.float32(a[k]).view(np.uint32) == np.float32(b[k]).view(np.uint32) if isinstance(a[k], (float, np.floating)) else a[k] == b[k], (other, k)
    return out["coop"], out["plain"]


_FALLBACKS = {}  # per variant of _plain_vs_parallel: cameras the single-workgroup kernel took over (vk_debug_counter)


@pytest.mark.parametrize("name", ["mono_320x240", "stereo_priors", "truncated", "refit_every_iteration", "ap3p", "cfg2", "wide_1241", "odd_323x241"])
def test_strict_parallel_structures_equal_the_plain_ones(strict, name):
    """strict mode now runs on the fast launch structures -- survivor queue with an exact rejection bound in the
    reference's own rounding (k_cost_rand_q_strict), lanes per site (k_global_prop_split_lean<.,.,true>), table + planned runs with the
    lane-split strict evaluation (k_local_runs_lean<.,.,.,true>), chunked fb_smooth lines, and the reference's 512-wide float tree as the
    XOR butterfly it is (k_pose_strict_par) -- and must give the bits of the plain structures in every output of a window: registered
    count, depth map, confidence map, poses, covariances, per-camera iteration counts and densities."""
    from voldor_amd import pyvoldor, synth
    mono = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --strict_math 1"
    kw = {}
    if name == "cfg2":
        sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320, fy=320, cx=320, cy=240, seed=233); cfg = mono + " --max_iters 8"
    elif name == "wide_1241":  # the separate table kernel (> 400k pixels), 8 frames, a disparity prior at the identity pose
        sc = synth.make_scene(w=1241, h=376, n_flows=8, fx=718.856, fy=718.856, cx=607.19, cy=185.22, seed=501, basefocal=386.1)
        cfg = "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 2 --strict_math 1"; kw = dict(basefocal=386.1, disparity=sc["disparity"])
    elif name == "odd_323x241":  # ragged chunks of the fb_smooth lines, partial tiles, partial chains
        sc = synth.make_scene(w=323, h=241, n_flows=3, fx=160, fy=160, cx=160, cy=120, seed=21); cfg = mono + " --max_iters 3"
    else:
        sc = synth.make_scene(w=320, h=240, n_flows=5 if name == "truncated" else 4, fx=160, fy=160, cx=160, cy=120, seed=dict(mono_320x240=11, stereo_priors=16, truncated=17, refit_every_iteration=15, ap3p=13)[name])
        cfg = mono + " --max_iters 3"
        if name == "stereo_priors":
            rng = np.random.default_rng(2)
            h, w = sc["depth_gt"].shape
            kw = dict(depth_priors=(sc["depth_gt"][None] * (1 + rng.normal(0, 0.03, (2, h, w)))).astype(np.float32),
                      depth_prior_poses=np.array([[0.002, -0.001, 0.0005, 0.01, 0.0, -0.02], [0, 0, 0, 0, 0, 0]], np.float32),
                      depth_prior_pconfs=rng.uniform(0.3, 1.0, (2, h, w)).astype(np.float32))
            cfg = "--silent --meanshift_kernel_var 0.1 --delta 0.5 --max_iters 3 --strict_math 1"
        elif name == "truncated":
            fl = sc["flows"].copy()
            fl[3:] = np.random.default_rng(4).uniform(-25, 25, fl[3:].shape).astype(np.float32)
            sc = dict(sc, flows=fl); cfg = mono + " --max_iters 5"
        elif name == "refit_every_iteration":
            cfg += " --rg_refine_last_only 0"
        elif name == "ap3p":
            cfg += " --lambdatwist 0"
    fx, fy, cx, cy = sc["K"]

    def run():
        o = pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=cfg, **kw)
        o["stats"] = pyvoldor.last_camera_stats(sc["flows"].shape[0]) if hasattr(pyvoldor, "last_camera_stats") else None
        return o
    a, b = _plain_vs_parallel(run)
    assert a["n_registered"] == b["n_registered"]
    for k in ("depth", "depth_conf", "poses", "poses_covar"):
        assert_bits(a[k], b[k], f"{name}: {k}")
    if a["stats"] is not None:
        for k in a["stats"]:
            np.testing.assert_array_equal(np.asarray(a["stats"][k]), np.asarray(b["stats"][k]), err_msg=f"{name}: {k}")


@pytest.mark.parametrize("n,nan_every,refit", [(8192, 0, True), (8192, 37, True), (8192, 3, False), (8192, 1, True), (8192, 600, False), (5000, 11, True), (700, 0, True), (500, 7, True), (6, 2, False), (1, 0, False), (2, 0, True)])
def test_strict_mode_kernel_parallel_tree_equals_the_block_walk(orc, small_scene, strict, n, nan_every, refit):
    """The strict mode kernel alone (vk_pose_mode_pool under strict math): k_pose_strict_par -- wave w owns blocks 2w, 2w+1 of the reference's
    512-row reduction, rows dealt to the lanes in bit-reversed order so that one transposing wave reduction IS strides 32 .. 1 of
    reduce_vector_sum.h:12-61 -- against k_pose_strict, which walks the blocks one after the other.  Pools of every shape the tree
    distinguishes: full (16 blocks), a partial last block, a pool inside one block (no second level), two rows, ONE row (the reference's
    loop does not run), NaN hypotheses anywhere (ordered compaction); with and without the refit; external and sampled start."""
    import hooks
    import test_gpu_kernels as tk
    rv, tv = tk._pose_pool(orc, small_scene, 1, max(n, 8))
    rv, tv = rv[:n].copy(), tv[:n].copy()
    fin = np.isfinite(rv.sum(1) + tv.sum(1))
    rv[~fin] = 0.01; tv[~fin] = 0.02  # start from an all-finite pool, then plant the NaNs
    if nan_every == 600:  # fewer finite rows than one block of the tree (14 of 8192): fifteen of the sixteen workgroups own no block
        keep = np.zeros(n, bool); keep[::600] = True
        rv[~keep] = np.nan
    elif nan_every:  # (nan_every == 1: ONE finite row in a pool of 8192 -- the reference's loop does not run, every other workgroup has nothing)
        rv[::nan_every] = np.nan
        if n > 1:
            rv[0] = 0.01  # (keep one finite row in any case)
    init = np.array([0.01, -0.005, 0.002, 0.1, -0.2, 0.9], np.float32)
    for ext in (True, False):
        def run():
            return hooks.pose_mode_pool(rv, tv, init, use_external_init_mean=ext, refit=refit, kernel_var=0.2, rvec_scale=25.0)
        a, b = _plain_vs_parallel(run)
        assert a["success"] == b["success"] and a["sample_count"] == b["sample_count"] == int(np.isfinite(rv.sum(1) + tv.sum(1)).sum())
        assert a["ms_iters"] == b["ms_iters"] and a["gu_iters"] == b["gu_iters"], (a["ms_iters"], b["ms_iters"], a["gu_iters"], b["gu_iters"])
        assert_bits(a["pose6"], b["pose6"], "pose"); assert_bits(a["covar"], b["covar"], "covar"); assert_bits(np.float32(a["density"]), np.float32(b["density"]), "density")


def test_the_give_up_path_of_the_cooperative_mode_kernel_was_taken(strict):
    """Runs after the comparisons above (file order): with the poll bound at one, workgroups DID give up and the single-workgroup kernel DID take
    cameras over (else "gave_up" compared the cooperative form with itself); with the product's bound nobody gave up."""
    if not _FALLBACKS:
        pytest.skip("the comparison tests of this file did not run")
    assert _FALLBACKS.get("gave_up", 0) > 0, _FALLBACKS
    assert _FALLBACKS.get("coop", 0) == 0 and _FALLBACKS.get("one_wg", 0) == 0, _FALLBACKS


# ---- the low-density regime ---------------------------------------------------------------------------------
def test_low_density_window_keeps_the_reference_pool(orc, strict):
    """~1 % valid correspondences for camera 0, a few dozen pixels for camera 1 (tests/ref_window_cases.py low_density): the
    reference forms all n_poses_to_sample hypotheses by indexing its compacted list.  The product's default draw (rejection, D3b)
    must fall back to that draw there (rank select in k_solve) -- NO --reference_draw flag in this test: same registered count
    as the reference pipeline, the full pool for camera 1, and in strict mode the oracle's bits (the oracle applies the same
    fallback rule)."""
    import ref_window_cases as cases
    from voldor_amd import kernels, pyvoldor
    c = dict(cases.window_cases())["low_density"]
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_window.npz"))
    fx, fy, cx, cy = c["K"]
    kw = dict(basefocal=c["basefocal"], disparity=c["disparity"])
    os.environ.pop("ORC_REFERENCE_DRAW", None)  # default draw on both sides
    o = orc.voldor(c["flows"], fx, fy, cx, cy, config=c["config"], **kw)
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"] + " --strict_math 1", **kw)
    st = pyvoldor.last_camera_stats(4)
    n_ref = int(gold["low_density/n_registered"])
    assert n_ref == 2 and g["n_registered"] == n_ref == o["n_registered"]
    assert st["pose_sample_count"][1] > 0.8 * 8192, st["pose_sample_count"]  # rejection alone would leave (1-(1-18/49152)^256)^4 ~ 1e-4 of them
    _assert_window_bits(o, g)
    kernels.set_strict_math(False)
    kernels.set_rand_epoch(0)
    f = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"], **kw)  # fast mode: same behaviour
    assert f["n_registered"] == n_ref and pyvoldor.last_camera_stats(4)["pose_sample_count"][1] > 0.8 * 8192
    from voldor_amd import synth
    rot, tr = synth.pose_errors(f["poses"][:1], gold["low_density/poses"][:1])  # camera 0 (~460 points) is well determined
    assert rot.max() < 2e-3 and tr.max() < 5e-2, (rot, tr)


# (fast mode vs strict mode over the 24-window ensemble, held to the reference's self-noise DISTRIBUTION: tests/test_gpu_ensemble.py)


VARIANTS = [  # (config suffix, estimator noise comparable to the default configuration?)
    ("--n_poses_to_sample 1000", False),              # a pool that does not fill the mode kernel's registers (8x fewer hypotheses: noisier)
    ("--meanshift_max_init_trials 3", True),          # fewer initial-mode trials than a batch
    ("--meanshift_max_init_trials 100", True),        # more than k_mode_trials takes: the mode kernel runs them itself
    ("--rg_refine 0", True),                          # no robust-Gaussian refit: k_pose_mode<false> on the last iteration too
    ("--fb_smooth 0", True),                          # k_cum_poses as its own launch (no fb_smooth launch to ride on)
    ("--norm_world_scale 0", True),                   # no world-scale factor in the E-step kernel
    ("--depth_local_prop_width 70", True),            # chains longer than a wave: step-by-step local propagation
    ("--depth_local_prop_width 48", True),            # one chain per wave (HALF = 64)
    ("--depth_global_prop_step 1", True),             # serial global propagation
    ("--max_trace_on_flow 0 --lambdatwist 0", False), # full-length flow traces; AP3P hypotheses (a different, noisier minimal solver)
    ("--reference_draw 0", True),                     # the rejection draw D3b on both sides
]


def test_config_variants_fast_vs_strict():
    """Configurations that switch the fast pipeline onto its alternative kernels, each against the strict pipeline on the same
    window (same draws).  Registered counts equal; the pose distances of the variants whose estimator is as noisy as the default one
    are held to the reference's own self-distances on this window (tests/golden/ref_window_noise.npz: the reference under eight
    independent 1-ulp jitter patterns vs its glibc run): every such variant within 2x the LARGEST of them and the median over the variants
    within the largest -- plain bounds: all variants run on ONE window and share ONE reference sample, so a rank-sum over them (round 3)
    assumed an independence that is not there; every variant additionally stays within 10x the largest reference
    self-distance (a wrong code path loses the window or is off by orders of magnitude)."""
    import ref_window_cases as cases
    from voldor_amd import kernels, pyvoldor, synth
    noise = np.load(os.path.join(os.path.dirname(__file__), "golden", "ref_window_noise.npz"))
    name = "mono_320x240"
    ref = [synth.pose_errors(noise[f"{name}/s{k}/poses"], noise[f"{name}/s0/poses"]) for k in range(1, 9)]
    ref_rot, ref_tr = [r.max() for r, _ in ref], [t.max() for _, t in ref]
    c = dict(cases.window_cases())[name]
    fx, fy, cx, cy = c["K"]
    rots, trs = [], []
    for extra, comparable in VARIANTS:
        res = {}
        for mode in ("strict", "fast"):
            kernels.set_rand_epoch(0)
            res[mode] = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"] + " " + extra + (" --strict_math 1" if mode == "strict" else " --strict_math 0"))
        s, f = res["strict"], res["fast"]
        assert s["n_registered"] == f["n_registered"] == c["flows"].shape[0], extra
        rot, tr = synth.pose_errors(f["poses"], s["poses"])
        assert rot.max() <= 10 * max(ref_rot) and tr.max() <= 10 * max(ref_tr), (extra, rot.max(), tr.max())
        if comparable:
            rots.append(rot.max()); trs.append(tr.max())
    print(f"fast vs strict over {len(rots)} variants: rot median {np.median(rots):.2e} max {max(rots):.2e} (reference self-distance median {np.median(ref_rot):.2e} max {max(ref_rot):.2e}); "
          f"trans median {np.median(trs):.2e} max {max(trs):.2e} ({np.median(ref_tr):.2e} / {max(ref_tr):.2e})")
    assert max(rots) <= 2 * max(ref_rot) and max(trs) <= 2 * max(ref_tr), (rots, trs, max(ref_rot), max(ref_tr))
    assert np.median(rots) <= max(ref_rot) and np.median(trs) <= max(ref_tr), (np.median(rots), np.median(trs))


@pytest.mark.gpu
def test_reference_mode_with_four_windows_in_flight_equals_one_at_a_time():
    """Four windows in flight in reference mode: four cooperative mode kernels (16 single-wave workgroups each) meet in their own CoopGlobal records at
    the same time, next to the other windows' per-pixel kernels.  Every output bit equals the window run alone; three repetitions."""
    import torch
    import ref_window_cases as rc
    from voldor_amd import kernels, pyvoldor
    c = dict(rc.window_cases())["mono_320x240"]
    fx, fy, cx, cy = c["K"]
    cfg = c["config"] + " --strict_math 1 --reference_draw 1 --reference_svd 1"
    B = 4
    fl = [torch.from_numpy(np.ascontiguousarray(c["flows"] * (1.0 + 0.01 * b))).cuda() for b in range(B)]
    h, w = c["flows"].shape[1:3]
    bits = lambda a: np.ascontiguousarray(np.asarray(a), np.float32).view(np.uint32)
    one = []
    for b in range(B):
        kernels.set_rand_epoch(0)
        d = torch.empty(h, w, device="cuda"); cf = torch.empty(h, w, device="cuda")
        o = pyvoldor.voldor_device(fl[b], fx, fy, cx, cy, config=cfg, depth_out=d, depth_conf_out=cf)
        one.append((o, d.cpu().numpy(), cf.cpu().numpy()))
    for rep in range(3):
        kernels.set_rand_epoch(0)
        d = [torch.empty(h, w, device="cuda") for _ in range(B)]; cf = [torch.empty(h, w, device="cuda") for _ in range(B)]
        out = pyvoldor.voldor_device_batch(fl, fx, fy, cx, cy, config=cfg, depth_out=d, depth_conf_out=cf)
        for b in range(B):
            o, dd, cc = one[b]
            assert out[b]["n_registered"] == o["n_registered"] > 0
            assert np.array_equal(bits(out[b]["poses"]), bits(o["poses"])) and np.array_equal(bits(out[b]["poses_covar"]), bits(o["poses_covar"])), (rep, b)
            assert np.array_equal(bits(d[b].cpu().numpy()), bits(dd)) and np.array_equal(bits(cf[b].cpu().numpy()), bits(cc)), (rep, b)
