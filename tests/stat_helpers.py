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


def ks_pvalue(x, y):
    x = np.asarray(x, np.float64); y = np.asarray(y, np.float64)
    x = x[np.isfinite(x)]; y = y[np.isfinite(y)]
    return float(stats.ks_2samp(x, y).pvalue)


def rank_sum_pvalue(values, samples):
    """One-sided test that values[w] is not stochastically LARGER than the draws samples[w] (same distribution per window w under H0).
    R_w = #{samples[w] < values[w]} is uniform on {0..n_w} under H0; the exact distribution of sum R_w comes from convolving the
    uniforms.  Returns P(sum >= observed)."""
    dist = np.array([1.0])
    total = 0
    for v, s in zip(values, samples):
        s = np.asarray(s, np.float64); s = s[np.isfinite(s)]
        if not np.isfinite(v) or len(s) == 0:
            continue
        total += int(np.sum(s < v))
        dist = np.convolve(dist, np.full(len(s) + 1, 1.0 / (len(s) + 1)))
    return float(dist[total:].sum())
