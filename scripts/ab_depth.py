"""Fast vs strict mode on one BASELINE workload on the GPU: window time, per-group times (vk_profile) and the distance of the fast
results from the strict ones.  python scripts/ab_depth.py [cfg2|cfg3|cfg5] [--strict]   (env VOLDOR_HIP_LIB selects another build)"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401,E402
from voldor_amd import capi, kernels, pyvoldor, synth  # noqa: E402
import bench  # noqa: E402

wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1].startswith("-") else "cfg2"]
W, H, N = wl["w"], wl["h"], wl["n"]
sc = synth.make_scene(w=W, h=H, n_flows=N, fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=233, basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
flows = torch.from_numpy(sc["flows"]).cuda()
extra = dict(basefocal=wl["basefocal"], disparity=torch.from_numpy(sc["disparity"]).cuda()) if wl["mode"] == "stereo" else {}
depth = torch.empty(H, W, device="cuda"); conf = torch.empty(H, W, device="cuda")
lib = capi.lib()


def run(cfg_extra="", reps=8):
    call = lambda: pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"] + cfg_extra, depth_out=depth, depth_conf_out=conf, **extra)  # noqa: E731
    for _ in range(3):
        call()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        kernels.set_rand_epoch(0)
        out = call()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / reps * 1e3
    lib.vk_profile_enable(1)
    for _ in range(3):
        call()
    g = {}
    tot, cnt = C.c_double(0), C.c_long(0)
    for name in ("optimize_depth", "optimize_camera_pose", "cost_rand", "local_pass", "bootstrap"):
        if lib.vk_profile_get(name.encode(), C.byref(tot), C.byref(cnt)) == 0 and cnt.value:
            g[name] = round(tot.value / cnt.value * 1e3, 1)
    lib.vk_profile_enable(0)
    return ms, g, out, depth.cpu().numpy().copy(), conf.cpu().numpy().copy()


res = {}
variants = [("fast", ""), ("fast", "")] + ([("strict", " --strict_math 1")] if "--strict" in sys.argv else [])
for name, cfgx in variants:
    ms, g, out, d, cf = run(cfgx, reps=2 if name == "strict" else 8)
    res[name] = (out, d, cf)
    gt = sc["poses_gt"].copy()
    if wl["mode"] == "mono":
        gt[:, 3:] /= np.mean(np.linalg.norm(gt[:, 3:], axis=1))
    rot, tr = synth.pose_errors(out["poses"], gt)
    print(f"{name:7s} {ms:8.3f} ms/window  n_reg {out['n_registered']}  groups(us) {g}  vs GT rot {rot.max():.2e} trans {tr.max():.2e}", flush=True)
if "strict" in res:
    o, d, cf = res["fast"]; so, sd, scf = res["strict"]
    rot, tr = synth.pose_errors(o["poses"], so["poses"])
    m = (cf > 0.5) & (scf > 0.5)
    rel = np.abs(d[m] - sd[m]) / sd[m]
    print(f"fast vs strict: rot {rot.max():.2e} trans {tr.max():.2e} depth median rel {np.median(rel):.2e} within 1e-3 {np.mean(rel < 1e-3):.3f}")
