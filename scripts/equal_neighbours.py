"""How many pixels share their depth value with a neighbour, per EM iteration?  (A local-propagation table entry c(x, d[nb]) with
d[nb] == d[x] is the stored cost[x] -- no evaluation needed.)  usage: python scripts/equal_neighbours.py [cfg2|cfg3|cfg5]"""
import sys
import numpy as np
sys.path.insert(0, ".")
import torch  # noqa
from voldor_amd import pyvoldor, synth, kernels
import bench
wl = bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg5"]
sc = synth.make_scene(w=wl["w"], h=wl["h"], n_flows=wl["n"], fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=233,
                      basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
kw = dict(basefocal=wl["basefocal"], disparity=sc["disparity"]) if wl["mode"] == "stereo" else {}
for it in (1, 2, 3, 4, 6, 8, 12):
    cfg = wl["cfg"].replace(f"--max_iters {wl['iters']}", f"--max_iters {it}")
    kernels.set_rand_epoch(0)
    o = pyvoldor.voldor(sc["flows"], wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=cfg, **kw)
    d = o["depth"]
    eh = np.mean(d[:, 1:] == d[:, :-1]); ev = np.mean(d[1:] == d[:-1])
    e4 = np.mean((d[1:-1, 1:-1] == d[1:-1, :-2]) & (d[1:-1, 1:-1] == d[1:-1, 2:]) & (d[1:-1, 1:-1] == d[:-2, 1:-1]) & (d[1:-1, 1:-1] == d[2:, 1:-1]))
    print(f"iters {it:2d}: equal to left {eh:.3f}  equal to upper {ev:.3f}  equal to all four {e4:.3f}  distinct values {len(np.unique(d))}")
