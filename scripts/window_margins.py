"""Prints the measured pose differences behind the thresholds of tests/test_gpu_vs_ref_window.py and
tests/test_gpu_reference_host_on_hip.py (run on the GPU box): how much room each bar leaves."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import ref_window_cases as cases
from voldor_amd import kernels, pyvoldor, synth
gold = np.load(os.path.join(ROOT, "tests", "golden", "ref_window.npz"))
for name, c in list(cases.window_cases()) + [cases.cfg2_case()]:
    fx, fy, cx, cy = c["K"]
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(c["flows"], fx, fy, cx, cy, basefocal=c["basefocal"], disparity=c["disparity"], depth_priors=c["depth_priors"],
                        depth_prior_poses=c["depth_prior_poses"], depth_prior_pconfs=c["depth_prior_pconfs"], config=c["config"])
    n = int(gold[f"{name}/n_registered"])
    rot, tr = synth.pose_errors(g["poses"], gold[f"{name}/poses"]) if g["n_registered"] == n else (np.array([np.nan]), np.array([np.nan]))
    big = not c["exact"]
    print(f"{name:20s} n {g['n_registered']}/{n}  rot {rot.max():.2e} (bar {1e-3 if big else 2e-3:.0e})  tr {tr.max():.2e} (bar {3e-2 if big else 8e-2:.0e})")
