"""tests/golden/ref_ensemble.npz -> tests/golden/ref_ensemble_bounds.json: summary statistics of what the REFERENCE pipeline does on the
ensembles of tests/ensemble_cases.py, used where a single window has to be held to a bound (tests/test_gpu_configs.py,
tests/test_gpu_voldor.py) and quoted in DESIGN.md section 5:
  self   distance of a run under 1-ulp jitter of expf/powf/logf to the glibc run of the same window (48 / 16 samples):
         worst rotation [rad], worst relative translation, 90th-percentile and median relative depth difference and fraction
         within 1e-3 of the confident pixels, mean |log covariance-trace ratio|
  gt     error of the glibc run against analytic ground truth (24 / 8 windows): worst rotation, worst relative translation,
         median relative depth error of the confident pixels
each as {median, p90, max} (for `within_1e-3`: {median, p10, min}).  python tests/golden/gen_ensemble_bounds.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import ensemble_cases as ens  # noqa: E402
import stat_helpers as sh  # noqa: E402
from voldor_amd import synth  # noqa: E402

SUB = 8


def q(x, low=False):
    x = np.asarray(x, np.float64)
    return ({"median": float(np.median(x)), "p10": float(np.percentile(x, 10)), "min": float(x.min())} if low else
            {"median": float(np.median(x)), "p90": float(np.percentile(x, 90)), "max": float(x.max())})


def main():
    g = np.load(os.path.join(HERE, "ref_ensemble.npz"))
    out = {}
    for kind, seeds in (("cfg2", ens.CFG2_SEEDS), ("cfg3", ens.CFG3_SEEDS)):
        def run(seed, mode):
            p = f"{kind}/s{seed}/{mode}/"
            return {"n_registered": int(g[p + "n_registered"]), "poses": g[p + "poses"], "poses_covar": g[p + "poses_covar"], "depth": g[p + "depth_sub"], "depth_conf": g[p + "conf_sub"]}
        self_d = {k: [] for k in sh.METRICS + ("depth_median",)}
        gt_e = {"rot": [], "trans": [], "depth": []}
        for seed in seeds:
            rg = run(seed, "g")
            for m in ("jA", "jB"):
                d = sh.window_distance(run(seed, m), rg)
                for k in self_d:
                    self_d[k].append(d[k])
            c = ens.make(kind, seed)
            gt = c["poses_gt"].copy(); dgt = c["depth_gt"][::SUB, ::SUB]
            if kind == "cfg2":
                s = np.mean(np.linalg.norm(gt[:, 3:], axis=1)); gt[:, 3:] /= s; dgt = dgt / s
            rot, tr = synth.pose_errors(rg["poses"], gt)
            m = rg["depth_conf"] > 0.5
            gt_e["rot"].append(rot.max()); gt_e["trans"].append(tr.max()); gt_e["depth"].append(np.median(np.abs(rg["depth"][m] - dgt[m]) / dgt[m]))
        out[kind] = {"windows": len(seeds), "self": {k: q(v, low=(k == "within_1e-3")) for k, v in self_d.items()}, "gt": {k: q(v) for k, v in gt_e.items()}}
    path = os.path.join(HERE, "ref_ensemble_bounds.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
