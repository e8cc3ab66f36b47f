"""GPU check of the strict-math mode: HIP (strict) vs the CPU oracle (strict), stage by stage and whole windows; prints mismatch
statistics instead of asserting (tests/test_gpu_strict.py holds the bars).   python scripts/strict_check.py [--big]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import K9  # noqa: E402
from oracle import orc  # noqa: E402
from voldor_amd import kernels, pyvoldor, synth  # noqa: E402
import test_gpu_kernels as tk  # noqa: E402

L = orc.lib()
L.orc_set_strict_math(1)
kernels.set_strict_math(True)
os.environ["ORC_REFERENCE_DRAW"] = "1"


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    return (a.view(np.uint32) == b.view(np.uint32)) | (np.isnan(a) & np.isnan(b))


def report(name, a, b):
    eq = bits_equal(a, b)
    msg = f"{name}: identical {eq.mean():.6f} ({(~eq).sum()} differ of {eq.size})"
    if not eq.all():
        d = np.abs(a.astype(np.float64) - b)[~eq]
        msg += f"  max abs diff {np.nanmax(d):.3e}"
    print(msg, flush=True)
    return eq.all()


sc = synth.make_scene(w=160, h=120, n_flows=4, fx=80, fy=80, cx=80, cy=60, seed=7)
K = K9(*sc["K"])
rng = np.random.default_rng(0)
flows, Rs, ts, depth, rig = tk._state(sc, rng)
# ---- stages through B-inner
m = rng.uniform(0.05, 0.95, (3, 120, 160)).astype(np.float32)
report("fb_smooth", orc.fb_smooth(m.copy()), kernels.fb_smooth_gpu(m.copy())[1])
for name, over in (("update_rigidness_only", dict(update_rigidness_only=1)), ("cost+rand", dict(global_prop_step=0, local_prop_width=0, fb_smooth=0)),
                   ("global only", dict(n_rand_samples=0, local_prop_width=0, fb_smooth=0)), ("local only", dict(n_rand_samples=0, global_prop_step=0, fb_smooth=0)),
                   ("full", dict())):
    (od, orig, _), (gd, grig, _) = tk._run_both(orc, sc, K, flows, Rs, ts, depth, rig, **over)
    report(f"optimize_depth[{name}] depth", od, gd); report(f"optimize_depth[{name}] rig", orig, grig)
# depth priors
N, h, w, _ = flows.shape
pri = (sc["depth_gt"][None] * (1 + rng.normal(0, 0.05, (2, h, w)))).astype(np.float32)
pc = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32); cf = rng.uniform(0.5, 1, (2, h, w)).astype(np.float32)
dpR = np.stack([np.eye(3), synth.rodrigues([0.01, -0.02, 0.005])]).astype(np.float32); dpt = np.array([[0, 0, 0], [0.05, 0.01, -0.1]], np.float32)
(od, orig, oc), (gd, grig, gc) = tk._run_both(orc, sc, K, flows, Rs, ts, depth, rig, priors=pri, pconfs=pc, confs=cf, dp_Rs=dpR, dp_ts=dpt, basefocal=40.0, disp_delta=1.0)
report("optimize_depth[priors] depth", od, gd); report("optimize_depth[priors] rig", orig, grig); report("optimize_depth[priors] confs", oc, gc)
# solvers
pts2, pts3, _ = tk._corr(orc, sc)
for solver, kw, fn in (("lambdatwist", {}, kernels.solve_batch_p3p_lambdatwist_gpu), ("ap3p", dict(use_ap3p=True), kernels.solve_batch_p3p_ap3p_gpu),
                       ("lambdatwist_f64", dict(use_double=True), kernels.solve_batch_p3p_lambdatwist_f64_gpu)):
    orv, otv = orc.solve_batch_p3p(pts3, pts2, K, 4096, **kw)
    grv, gtv = fn(pts3, pts2, K, 4096)
    report(f"solve[{solver}] rvecs", orv, grv); report(f"solve[{solver}] tvecs", otv, gtv)


def window(name, scn, cfg, **extra):
    fx, fy, cx, cy = scn["K"]
    t0 = time.time(); o = orc.voldor(scn["flows"], fx, fy, cx, cy, config=cfg, **extra); to = time.time() - t0
    kernels.set_rand_epoch(0)
    t0 = time.time(); g = pyvoldor.voldor(scn["flows"], fx, fy, cx, cy, config=cfg + " --strict_math 1 --reference_draw 1", **extra); tg = time.time() - t0
    print(f"--- window {name}: n_registered oracle {o['n_registered']} hip {g['n_registered']}  (oracle {to:.1f}s, hip {tg:.2f}s)")
    n = min(o["n_registered"], g["n_registered"])
    report("  poses", o["poses"][:n], g["poses"][:n]); report("  covar", o["poses_covar"][:n], g["poses_covar"][:n])
    report("  depth", o["depth"], g["depth"]); report("  depth_conf", o["depth_conf"], g["depth_conf"])
    if n:
        rot, tr = synth.pose_errors(g["poses"][:n], o["poses"][:n]); print("  pose err", rot.max(), tr.max())


s2 = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=11)
window("320x240 mono 1 iter no refit", s2, "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 1 --rg_refine 0")
window("320x240 mono 3 iters", s2, "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 3")
window("320x240 mono ap3p 2 iters", s2, "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 2 --lambdatwist 0")
s3 = synth.make_scene(w=320, h=240, n_flows=4, fx=160, fy=160, cx=160, cy=120, seed=12, basefocal=80.0)
window("320x240 stereo 3 iters", s3, "--silent --meanshift_kernel_var 0.1 --disp_delta 1 --delta 0.2 --max_iters 3", basefocal=80.0, disparity=s3["disparity"])
if "--big" in sys.argv:
    s5 = synth.make_scene(w=640, h=480, n_flows=5, fx=320, fy=320, cx=320, cy=240, seed=233)
    window("BASELINE cfg2", s5, "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8")
