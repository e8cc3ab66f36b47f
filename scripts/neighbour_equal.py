"""Share of pixels whose depth equals a 4-neighbour's bit for bit after one window (what a cost-table entry evaluates twice: the pixel under its own depth).
usage: python scripts/neighbour_equal.py cfg2 cfg3 cfg5"""
import sys
sys.path.insert(0, ".")
import numpy as np, torch
from voldor_amd import pyvoldor, synth
import bench
for name in sys.argv[1:]:
    wl = bench.WORKLOADS[name]
    sc = synth.make_scene(w=wl["w"], h=wl["h"], n_flows=wl["n"], fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=233, basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
    flows = torch.from_numpy(sc["flows"]).cuda()
    kw = dict(basefocal=wl["basefocal"], disparity=torch.from_numpy(sc["disparity"]).cuda()) if wl["mode"] == "stereo" else {}
    depth = torch.empty(wl["h"], wl["w"], device="cuda"); conf = torch.empty_like(depth)
    for iters in (1, 2, 4, None):
        cfg = wl["cfg"] if iters is None else wl["cfg"] + f" --max_iters {iters}"
        try:
            o = pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=cfg, depth_out=depth, depth_conf_out=conf, **kw)
        except Exception as e:
            print(name, iters, "failed", e); continue
        d = depth.cpu().numpy().view(np.uint32)
        eq = [np.mean(d[:, 1:] == d[:, :-1]), np.mean(d[1:, :] == d[:-1, :])]
        # whole 64-pixel row pieces whose every pixel equals its left neighbour (a wave of the tiled table kernel)
        e = (d[:, 1:] == d[:, :-1])
        ww = (e.shape[1] // 64) * 64
        wave = e[:, :ww].reshape(e.shape[0], -1, 64).all(axis=2).mean()
        print(f"{name} iters {iters}: equal to left {eq[0]:.3f}  equal to upper {eq[1]:.3f}  whole 64-pixel pieces equal-left {wave:.3f}  n_registered {o['n_registered']}")
