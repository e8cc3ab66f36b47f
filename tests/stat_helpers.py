"""Window-to-window distances and the two tests that replace hand-set tolerances between runs of a chaotic estimator:
  * two-sample Kolmogorov-Smirnov over an ensemble of windows (tests/test_gpu_ensemble.py),
  * a rank-sum test of one run per window against that window's own sample of reference self-distances
    (tests/test_gpu_vs_ref_window.py)."""
import numpy as np
from scipy import stats

from voldor_amd import synth

METRICS = ("rot", "trans", "depth", "within_1e-3", "logcov")


def window_distance(a, b, scale_free=False):
    """Distances between two runs of one window (dicts with n_registered, poses, poses_covar, depth, depth_conf -- maps sampled on the
    same grid).  rot / trans: worst pose of the window (geodesic rad, relative translation); depth: 90th percentile of the relative
    difference over the pixels both runs are confident about (NaN when fewer than 50; the median is useless here: two runs of the
    reference that differ by 1-ulp jitter share most pixels bit for bit on stereo windows); within_1e-3: fraction of those pixels
    within 1e-3 (north_star's depth bar); logcov: mean |log trace ratio| of the pose covariances.
    Returns None when the registered counts differ (tested separately)."""
    n = int(a["n_registered"])
    if n != int(b["n_registered"]) or n == 0:
        return None
    rot, tr = synth.pose_errors(a["poses"], b["poses"])
    m = (a["depth_conf"] > 0.5) & (b["depth_conf"] > 0.5)
    s = 1.0
    if scale_free:  # monocular windows are normalised to mean |t| = 1 by both pipelines; kept for callers that compare against ground truth
        s = np.mean(np.linalg.norm(a["poses"][:, 3:], axis=1)) / np.mean(np.linalg.norm(b["poses"][:, 3:], axis=1))
    rel = np.abs(a["depth"][m] / s - b["depth"][m]) / b["depth"][m]
    ta = np.trace(np.asarray(a["poses_covar"], np.float64), axis1=1, axis2=2)
    tb = np.trace(np.asarray(b["poses_covar"], np.float64), axis1=1, axis2=2)
    ok = (ta > 0) & (tb > 0)
    return {"rot": float(rot.max()), "trans": float(tr.max()), "depth": float(np.percentile(rel, 90)) if m.sum() >= 50 else float("nan"), "depth_median": float(np.median(rel)) if m.sum() >= 50 else float("nan"),
            "within_1e-3": float(np.mean(rel < 1e-3)) if m.sum() >= 50 else float("nan"),
            "logcov": float(np.mean(np.abs(np.log(ta[ok] / tb[ok])))) if ok.any() else float("nan")}


def ks_pvalue(x, y, what=""):
    """Two-sample KS p-value.  x = the side under test: a non-finite metric there (e.g. fewer than 50 confident pixels: the depth map
    degraded) FAILS instead of silently leaving the sample; y = the reference's sample, where a non-finite value is dropped."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    assert np.isfinite(x).all(), f"{what}: non-finite metric on the side under test in {int((~np.isfinite(x)).sum())} of {len(x)} windows"
    y = y[np.isfinite(y)]
    return float(stats.ks_2samp(x, y).pvalue)


def median_ratio_ci(x, y, n_boot=4000, level=0.95, seed=12345):
    """median(x) / median(y) with a percentile-bootstrap confidence interval, both samples resampled independently (reported for
    information next to the paired statistic below).  Returns (ratio, lo, hi).  x must be finite."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    assert np.isfinite(x).all()
    y = y[np.isfinite(y)]
    rng = np.random.default_rng(seed)
    bx = np.median(x[rng.integers(0, len(x), (n_boot, len(x)))], axis=1)
    by = np.median(y[rng.integers(0, len(y), (n_boot, len(y)))], axis=1)
    r = bx / by
    a = (1.0 - level) / 2.0
    return float(np.median(x) / np.median(y)), float(np.quantile(r, a)), float(np.quantile(r, 1.0 - a))


def paired_mean_ratio_ci(x, y, n_boot=4000, level=0.95, seed=12345):
    """The effect size the equivalence statements rest on (a KS test can only fail to reject).  x[w] = the metric of the
    side under test on window w, y[w] (or y[w, :]: several reference draws of that window, averaged) = the reference's own value on the SAME
    window.  Statistic: mean(x) / mean(y); interval: percentile bootstrap that resamples WINDOWS (the independent units; both sides of a
    window stay together, which removes the between-window spread -- some windows are simply harder -- from the interval).  Means, not
    medians: the fraction of pixels within 1e-3 is bimodal over the windows (about 0.69 or about 0.02, for the reference against itself as
    well), and a median of a bimodal sample jumps between the modes (its bootstrap interval came out as [0.1, 7]).  Returns (ratio, lo, hi)."""
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    assert np.isfinite(x).all(), "non-finite metric on the side under test"
    if y.ndim == 2:
        y = np.nanmean(y, axis=1)
    assert x.shape == y.shape and np.isfinite(y).all()
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, len(x), (n_boot, len(x)))
    r = x[idx].mean(axis=1) / y[idx].mean(axis=1)
    a = (1.0 - level) / 2.0
    return float(x.mean() / y.mean()), float(np.quantile(r, a)), float(np.quantile(r, 1.0 - a))


# Equivalence margins of the distribution tests: the 95 % interval of the paired ratio of means has to lie below RATIO_MAX for the
# distances and errors (lower is better) and above RATIO_MIN for the fraction of pixels within 1e-3 (higher is better).
RATIO_MAX = 1.25
RATIO_MIN = 0.8


def equivalence_ok(metric, x, y):
    """(ok, ratio, lo, hi) of paired_mean_ratio_ci against the margin of `metric`."""
    r, lo, hi = paired_mean_ratio_ci(x, y)
    ok = lo >= RATIO_MIN if metric == "within_1e-3" else hi <= RATIO_MAX
    return ok, r, lo, hi


def rank_sum_pvalue(values, samples):
    """One-sided test that values[w] is not stochastically LARGER than the draws samples[w] (same distribution per window w under H0).
    R_w = #{samples[w] < values[w]} is uniform on {0..n_w} under H0; the exact distribution of sum R_w comes from convolving the
    uniforms.  Returns P(sum >= observed)."""
    dist = np.array([1.0])
    total = 0
    for v, s in zip(values, samples):
        s = np.asarray(s, np.float64); s = s[np.isfinite(s)]
        if len(s) == 0:
            continue
        if not np.isfinite(v):
            v = np.inf  # a metric that could not be formed on the side under test ranks above every reference draw (it does not leave the sample)
        total += int(np.sum(s < v))
        dist = np.convolve(dist, np.full(len(s) + 1, 1.0 / (len(s) + 1)))
    return float(dist[total:].sum())
