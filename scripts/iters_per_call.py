"""Mean-shift iterations per call by EM iteration, and gate iterations of the last-iteration refit ("report iterations per call").
A run with --max_iters k ends with EM iteration k, whose camera records vk_last_camera_stats returns; the refit only runs in a window's last iteration, so
the mean-shift counts of iteration k are those of the 8-iteration window.  usage: python scripts/iters_per_call.py cfg2 [seeds...]"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from voldor_amd import pyvoldor, synth, kernels
import bench
wl = bench.WORKLOADS[sys.argv[1]]
seeds = [int(s) for s in sys.argv[2:]] or [233, 400, 401, 402]
ms = {k: [] for k in range(1, wl["iters"] + 1)}; gu = []
for seed in seeds:
    sc = synth.make_scene(w=wl["w"], h=wl["h"], n_flows=wl["n"], fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=seed, basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
    flows = torch.from_numpy(sc["flows"]).cuda()
    kw = dict(basefocal=wl["basefocal"], disparity=torch.from_numpy(sc["disparity"]).cuda()) if wl["mode"] == "stereo" else {}
    for k in range(1, wl["iters"] + 1):
        kernels.set_rand_epoch(0)
        cfg = wl["cfg"].replace(f"--max_iters {wl['iters']}", f"--max_iters {k}")
        o = pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=cfg, **kw)
        st = pyvoldor.last_camera_stats(o["n_registered"])
        ms[k] += list(st["ms_iters"]);
        if k == wl["iters"]: gu += list(st["gu_iters"])
for k in ms: print(f"EM iteration {k}: mean-shift iterations per call  mean {np.mean(ms[k]):5.1f}  min {np.min(ms[k])}  max {np.max(ms[k])}   ({len(ms[k])} calls)")
print(f"refit (last EM iteration): gate iterations per call  mean {np.mean(gu):5.1f}  min {np.min(gu)}  max {np.max(gu)}   ({len(gu)} calls)")
