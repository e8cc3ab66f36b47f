"""cost map + random samples alone: every fast variant against the strict kernel on the same input (fraction of pixels with the strict depth)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import K9
from voldor_amd import capi, kernels, synth
import test_gpu_kernels as tk
lib = capi.lib()
BIG = 1 << 62
for (w, h, N, with_prior) in ((640, 480, 5, False), (640, 480, 5, True), (1241, 376, 8, True)):
    sc = synth.make_scene(w=w, h=h, n_flows=N, fx=w / 2, fy=w / 2, cx=w / 2, cy=h / 2, seed=233, basefocal=w / 4)
    rng = np.random.default_rng(0)
    K = K9(*sc["K"])
    flows, Rs, ts, depth, rig = tk._state(sc, rng, noise=0.05)
    pri = pc = cf = dR = dt = None
    if with_prior:
        pri = ((w / 4) / sc["disparity"])[None].astype(np.float32); pc = np.ones_like(pri); cf = np.ones_like(pri)
        dR = np.eye(3, dtype=np.float32)[None]; dt = np.zeros((1, 3), np.float32)
    kw = tk._od_kwargs(global_prop_step=0, local_prop_width=0, fb_smooth=0, basefocal=w / 4 if with_prior else 0.0, disp_delta=1.0 if with_prior else -1.0, delta=0.2)
    out = {}
    for name, strict, variant, thr in (("strict", True, 1, None), ("q", False, 1, None), ("hm", False, 2, (BIG, BIG)), ("fm", False, 2, (0, BIG)), ("fm-sorted", False, 2, (0, 0)), ("legacy", False, 0, (BIG, BIG))):
        kernels.set_strict_math(strict); lib.vk_set_fast_variant(variant)
        kernels.set_frame_major_threshold(*(thr or (24 << 20, 64 << 20)))
        kernels.set_rand_epoch(7)
        d, r, c = kernels.optimize_depth_gpu(flows, rig, pri, pc, cf, depth, K, Rs, ts, dR, dt, kw["abs_resize_factor"], N, 0 if pri is None else 1, w, h, kw["basefocal"],
                                             kw["n_rand_samples"], 0, 0, kw["lambda_"], kw["omega"], kw["disp_delta"], kw["delta"], 0, 0.5, 0.9, 1.0, 0)
        out[name] = d
    kernels.set_strict_math(False)
    s = out["strict"]
    print(f"{w}x{h} N={N} prior={with_prior}: changed by strict {np.mean(s != depth):.4f}; " + "  ".join(f"{k}: same as strict {np.mean(v == s):.5f}" for k, v in out.items() if k != "strict"))
