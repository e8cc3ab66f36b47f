"""Diagnostic: distances of several HIP modes to the reference's glibc run over the ensemble, next to the reference's own jitter runs.
usage: python scripts/ensemble_diag.py [cfg2|cfg3] [n_seeds]"""
import sys
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import ensemble_cases as ens
import stat_helpers as sh
from voldor_amd import kernels, pyvoldor
kind = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
seeds = (ens.CFG2_SEEDS if kind == "cfg2" else ens.CFG3_SEEDS)[:int(sys.argv[2]) if len(sys.argv) > 2 else 99]
g = np.load("tests/golden/ref_ensemble.npz")
def ref(seed, mode):
    p = f"{kind}/s{seed}/{mode}/"
    return {"n_registered": int(g[p + "n_registered"]), "poses": g[p + "poses"], "poses_covar": g[p + "poses_covar"], "depth": g[p + "depth_sub"], "depth_conf": g[p + "conf_sub"]}
modes = {"fast": "", "strict": " --strict_math 1", "strict+refdraw+refsvd": " --strict_math 1 --reference_draw 1 --reference_svd 1", "fast+refdraw": " --reference_draw 1"}
D = {m: {k: [] for k in sh.METRICS} for m in list(modes) + ["ref_jitter"]}
for seed in seeds:
    c = ens.make(kind, seed)
    fx, fy, cx, cy = c["K"]
    rg = ref(seed, "g")
    for m, extra in modes.items():
        kernels.set_rand_epoch(0)
        o = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], config=c["config"] + extra)
        hip = {"n_registered": o["n_registered"], "poses": o["poses"], "poses_covar": o["poses_covar"], "depth": o["depth"][::8, ::8], "depth_conf": o["depth_conf"][::8, ::8]}
        d = sh.window_distance(hip, rg)
        for k in sh.METRICS: D[m][k].append(d[k])
    for j in ("jA", "jB"):
        d = sh.window_distance(ref(seed, j), rg)
        for k in sh.METRICS: D["ref_jitter"][k].append(d[k])
for k in sh.METRICS:
    print(k)
    for m in D:
        x = np.array(D[m][k])
        print(f"   {m:24s} median {np.median(x):.3e} p10 {np.percentile(x,10):.3e} p90 {np.percentile(x,90):.3e}   KS vs ref_jitter p={sh.ks_pvalue(x, D['ref_jitter'][k]):.4f}")
