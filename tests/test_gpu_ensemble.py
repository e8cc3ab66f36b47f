"""The fast HIP path against the REFERENCE pipeline over an ensemble of windows, as distributions.

tests/golden/ref_ensemble.npz (tests/golden/gen_golden_ensemble.py) holds the reference's own pipeline (voldor/*.cpp +
gpu-kernels/*.cu executed on the CPU) on 24 independent BASELINE-cfg2 windows and 8 cfg3 windows, each under glibc (g) and under two
independent 1-ulp jitter patterns of its transcendentals (jA, jB).  {jA vs g, jB vs g} is what a last-bit change of expf / powf /
logf does to the reference's OWN output; the fast path (v_exp_f32 / v_log_f32, re-associated sums) is such a change.  Asserted:

  (i)  the distances {fast HIP vs g} and {jitter vs g} come from one distribution: two-sample Kolmogorov-Smirnov p > 0.01 for the
       worst rotation, worst relative translation, 90th-percentile relative depth, fraction of the confident pixels within 1e-3 and
       |log covariance-trace ratio| of a window; AND, since a KS test can only fail to reject (round 4): the 95 %
       bootstrap interval of mean(d_hip) / mean(d_ref), windows resampled as pairs (stat_helpers.paired_mean_ratio_ci), lies below 1.25
       (above 0.8 for the within-1e-3 fraction) -- equivalence within a margin, on 144 cfg2 / 48 cfg3 windows (24 / 8 in round 3, 72 / 48 in round 4).  The
       test bites: a 12-step cap on the Newton loop of the P3P cubic (vk_debug_switch "newton_cap", 0.13 ms faster) fails it;
  (ii) the errors against analytic ground truth of {fast HIP} and {reference g} come from one distribution (same test);
  (iii) every window registers the reference's frame count.

north_star's bars -- 1e-3 relative translation, 99 % of the confident pixels within 1e-3 -- are printed next to what the reference
achieves against itself: they are below the reference's own reproducibility."""
import os

import numpy as np
import pytest

import ensemble_cases as ens
import stat_helpers as sh

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("eight_point_bootstrap")]  # (fast windows here are held against the oracle / the reference ensembles: same two-view pose, conftest.py)
GOLD = os.path.join(os.path.dirname(__file__), "golden", "ref_ensemble.npz")
GOLD5 = os.path.join(os.path.dirname(__file__), "golden", "ref_ensemble5.npz")  # the cfg2 ensemble again, the reference pipeline started from the five-point two-view pose (round 6)
SUB = 8
ALPHA = 0.01
FS_SEEDS = ens.CFG2_SEEDS  # windows of the fast-vs-strict test (a strict window costs ~15 ms since round 4)


def _dump(name, obj):
    """raw per-window distances next to the verdict (gpurun_out/ is merged back from the GPU box): what the numbers in DESIGN.md are made from"""
    import json
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        json.dump(obj, open(os.path.join(d, name + ".json"), "w"))
    except OSError:
        pass


def _ref_run(g, kind, seed, mode):
    p = f"{kind}/s{seed}/{mode}/"
    return {"n_registered": int(g[p + "n_registered"]), "poses": g[p + "poses"], "poses_covar": g[p + "poses_covar"],
            "depth": g[p + "depth_sub"], "depth_conf": g[p + "conf_sub"]}


def _gt_errors(run, c, mono):
    gt = c["poses_gt"].copy()
    dgt = c["depth_gt"][::SUB, ::SUB]
    if mono:  # both pipelines normalise a monocular window to mean |t| = 1 (voldor.cpp:309-317)
        s = np.mean(np.linalg.norm(gt[:, 3:], axis=1))
        gt[:, 3:] /= s; dgt = dgt / s
    rot, tr = __import__("voldor_amd").synth.pose_errors(run["poses"], gt[:int(run["n_registered"])])
    m = run["depth_conf"] > 0.5
    return {"rot": float(np.sqrt(np.mean(rot ** 2))), "trans": float(np.sqrt(np.mean(tr ** 2))),
            "depth": float(np.median(np.abs(run["depth"][m] - dgt[m]) / dgt[m])) if m.sum() >= 50 else float("nan")}


@pytest.mark.parametrize("kind,seeds", [("cfg2", ens.CFG2_SEEDS), ("cfg3", ens.CFG3_SEEDS), ("cfg2-five-point", ens.CFG2_SEEDS)])
def test_fast_path_is_a_draw_from_the_reference_self_noise(kind, seeds):
    """cfg2 / cfg3: the fast path started from the 8-point two-view pose (the module's fixture) against the reference pipeline started from the same pose.
    cfg2-five-point (round 6): the fast path as it ships -- five-point LMedS bootstrap -- against the reference pipeline started from THAT pose
    (tests/golden/ref_ensemble5.npz: gen_golden_ensemble.py --five-point injects the pose of the product's own host build where the reference calls OpenCV)."""
    import hooks
    from voldor_amd import kernels, pyvoldor
    gold = GOLD
    if kind == "cfg2-five-point":
        gold, kind = GOLD5, "cfg2"
        prev = hooks.debug_switch("bootstrap_default", 0)  # the product's default (the fixture set 8 and restores its own previous value after the test)
        assert prev == 8
    if not os.path.exists(gold):
        pytest.skip(f"{os.path.relpath(gold)} not generated")
    g = np.load(gold)
    seeds = [s_ for s_ in seeds if f"{kind}/s{s_}/jB/n_registered" in g.files and f"{kind}/s{s_}/g/n_registered" in g.files and f"{kind}/s{s_}/jA/n_registered" in g.files]
    assert len(seeds) >= 24, len(seeds)
    mono = kind == "cfg2"
    d_hip = {k: [] for k in sh.METRICS}
    d_ref = {k: [] for k in sh.METRICS}
    e_hip = {k: [] for k in ("rot", "trans", "depth")}
    e_ref = {k: [] for k in ("rot", "trans", "depth")}
    for seed in seeds:
        c = ens.make(kind, seed)
        fx, fy, cx, cy = c["K"]
        kernels.set_rand_epoch(0)
        o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], config=c["config"])
        hip = {"n_registered": o["n_registered"], "poses": o["poses"], "poses_covar": o["poses_covar"],
               "depth": o["depth"][::SUB, ::SUB], "depth_conf": o["depth_conf"][::SUB, ::SUB]}
        rg = _ref_run(g, kind, seed, "g")
        assert hip["n_registered"] == rg["n_registered"] == c["flows"].shape[0], (kind, seed)  # (iii)
        d = sh.window_distance(hip, rg)
        for k in d_hip:
            d_hip[k].append(d[k])
        for mode in ("jA", "jB"):
            rj = _ref_run(g, kind, seed, mode)
            assert rj["n_registered"] == rg["n_registered"]
            dj = sh.window_distance(rj, rg)
            for k in d_ref:
                d_ref[k].append(dj[k])
        eh, er = _gt_errors(hip, c, mono), _gt_errors(rg, c, mono)
        for k in e_hip:
            e_hip[k].append(eh[k]); e_ref[k].append(er[k])
    _dump(f"ensemble_{kind}" + ("_five_point" if gold == GOLD5 else ""), {"d_hip": d_hip, "d_ref": d_ref, "e_hip": e_hip, "e_ref": e_ref, "seeds": list(seeds)})
    report = {}
    fails = []
    for k in sh.METRICS:  # (i) KS: "one distribution" cannot be rejected; and the EFFECT SIZE: median ratio inside the equivalence margin
        p = sh.ks_pvalue(d_hip[k], d_ref[k], f"{kind} {k}")
        ok, r, lo, hi = sh.equivalence_ok(k, d_hip[k], np.asarray(d_ref[k]).reshape(-1, 2))  # (window w: its two jittered reference runs)
        report[k] = (float(np.median(d_hip[k])), float(np.median(d_ref[k])), p, r, lo, hi)
        if not p > ALPHA:
            fails.append(f"{kind}: {k} distance to the reference's glibc run is not a draw from the reference's self-noise (KS p = {p:.4f}); medians {report[k][:2]}")
        if not ok:
            fails.append(f"{kind}: {k} paired ratio of means fast-HIP / reference-self-noise {r:.3f}, 95 % bootstrap interval [{lo:.3f}, {hi:.3f}] leaves the margin "
                         f"({'>= %.2f' % sh.RATIO_MIN if k == 'within_1e-3' else '<= %.2f' % sh.RATIO_MAX})")
    for k in e_hip:  # (ii)
        p = sh.ks_pvalue(e_hip[k], e_ref[k], f"{kind} gt {k}")
        r, lo, hi = sh.paired_mean_ratio_ci(e_hip[k], e_ref[k])
        report["gt_" + k] = (float(np.median(e_hip[k])), float(np.median(e_ref[k])), p, r, lo, hi)
        if not p > ALPHA:
            fails.append(f"{kind}: {k} error against ground truth differs in distribution (KS p = {p:.4f}); medians {np.median(e_hip[k]):.3e} vs {np.median(e_ref[k]):.3e}")
        if not hi <= sh.RATIO_MAX:
            fails.append(f"{kind}: {k} error against ground truth: paired ratio of means {r:.3f} [{lo:.3f}, {hi:.3f}] above {sh.RATIO_MAX}")
    print(f"\n{kind} ({len(seeds)} windows): median distance to the reference's glibc run, fast HIP | reference under 1-ulp jitter | KS p | paired ratio of means [95 % bootstrap interval over the windows]")
    for k, (a, b, p, r, lo, hi) in report.items():
        print(f"  {k:12s} {a:.3e} | {b:.3e} | {p:.3f} | {r:.3f} [{lo:.3f}, {hi:.3f}]")
    assert not fails, "\n".join(fails)
    print(f"  fraction of confident pixels within 1e-3: fast HIP {np.median(d_hip['within_1e-3']):.3f} | reference vs itself {np.median(d_ref['within_1e-3']):.3f} (north_star asks 0.99)")
    print(f"  worst relative translation: fast HIP {np.max(d_hip['trans']):.2e} | reference vs itself {np.max(d_ref['trans']):.2e} (north_star asks 1e-3)")


def test_fast_vs_strict_is_a_draw_from_the_reference_self_noise():
    """The fast kernels (hardware v_log / v_exp, fused multiply-adds, re-associated sums, Moebius fb_smooth, packed mode kernels)
    against the STRICT kernels of the same library -- which equal the reference's own code bit for bit (tests/test_gpu_vs_ref_window.py)
    -- on the 24 cfg2 windows, same draws: the distances {fast vs strict} must come from the distribution of {reference under 1-ulp
    jitter vs reference} (KS p > 0.01 per metric).  Replaces round 2's yardstick (three runs of one window times a slack of 2)."""
    from voldor_amd import kernels, pyvoldor
    if not os.path.exists(GOLD):
        pytest.skip("tests/golden/ref_ensemble.npz not generated")
    g = np.load(GOLD)
    d_fs = {k: [] for k in sh.METRICS}
    d_ref = {k: [] for k in sh.METRICS}
    for seed in FS_SEEDS:
        c = ens.make("cfg2", seed)
        fx, fy, cx, cy = c["K"]
        runs = {}
        for mode in ("fast", "strict"):
            kernels.set_rand_epoch(0)
            o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, config=c["config"] + (" --strict_math 1" if mode == "strict" else " --strict_math 0"))
            runs[mode] = {"n_registered": o["n_registered"], "poses": o["poses"], "poses_covar": o["poses_covar"],
                          "depth": o["depth"][::SUB, ::SUB], "depth_conf": o["depth_conf"][::SUB, ::SUB]}
        assert runs["fast"]["n_registered"] == runs["strict"]["n_registered"] == 5
        d = sh.window_distance(runs["fast"], runs["strict"])
        for k in d_fs:
            d_fs[k].append(d[k])
        rg = _ref_run(g, "cfg2", seed, "g")
        for mode in ("jA", "jB"):
            dj = sh.window_distance(_ref_run(g, "cfg2", seed, mode), rg)
            for k in d_ref:
                d_ref[k].append(dj[k])
    _dump("ensemble_fast_vs_strict", {"d_fs": d_fs, "d_ref": d_ref, "seeds": list(FS_SEEDS)})
    print(f"\ncfg2 ({len(FS_SEEDS)} windows): median distance fast vs strict | reference under 1-ulp jitter vs reference | KS p | paired ratio of means [95 % bootstrap interval over the windows]")
    fails = []
    for k in sh.METRICS:
        p = sh.ks_pvalue(d_fs[k], d_ref[k], f"fast vs strict {k}")
        ok, r, lo, hi = sh.equivalence_ok(k, d_fs[k], np.asarray(d_ref[k]).reshape(-1, 2))
        print(f"  {k:12s} {np.median(d_fs[k]):.3e} | {np.median(d_ref[k]):.3e} | {p:.3f} | {r:.3f} [{lo:.3f}, {hi:.3f}]")
        if not p > ALPHA:
            fails.append(f"{k}: fast vs strict is not a draw from the reference's self-noise (KS p = {p:.4f})")
        if not ok:
            fails.append(f"{k}: paired ratio of means {r:.3f} [{lo:.3f}, {hi:.3f}] leaves the equivalence margin")
    assert not fails, "\n".join(fails)
