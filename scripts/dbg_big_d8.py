"""debug aid: HIP strict (D8 polar factor) vs the oracle strict, live, on a big window, by number of EM iterations"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np
from oracle import orc
from voldor_amd import pyvoldor, kernels
import big_window_cases as big
name = sys.argv[1]
c = big.make(name); fx, fy, cx, cy = c["K"]
for extra in sys.argv[2:] or [""]:
    for k in (1, 2, 3, 8):
        cfg = c["config"].replace("--max_iters 8", f"--max_iters {k}").replace("--max_iters 12", f"--max_iters {k}") + " " + extra
        orc.lib().orc_set_strict_math(1)
        orc.set_reference_svd("--reference_svd 1" in extra)
        o = orc.voldor(c["flows"], fx, fy, cx, cy, config=cfg.replace("--reference_svd 1", ""), basefocal=c["basefocal"], disparity=c["disparity"])
        orc.lib().orc_set_strict_math(0)
        kernels.set_rand_epoch(0)
        h = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], config=cfg + " --strict_math 1")
        b = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
        print(f"{name} [{extra}] iters {k}: n_reg {o['n_registered']} {h['n_registered']}  depth neq {np.mean(b(o['depth']) != b(h['depth'])):.5f}  conf neq {np.mean(b(o['depth_conf']) != b(h['depth_conf'])):.5f}"
              f"  poses neq {(b(o['poses']) != b(h['poses'])).sum()} max {np.abs(o['poses'] - h['poses']).max():.2e}  covar neq {(b(o['poses_covar']) != b(h['poses_covar'])).sum()}", flush=True)
