import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import orc
from voldor_amd import pyvoldor, synth, kernels
sc = synth.make_scene(w=320, h=240, n_flows=5, fx=160, fy=160, cx=160, cy=120, seed=233)
fx, fy, cx, cy = sc["K"]
base = "--silent --meanshift_kernel_var 0.2 --delta 1.5 "
for extra in ["--max_iters 1 --optimize_depth 0 --rg_refine 0 --norm_world_scale 0",
              "--max_iters 1 --rg_refine 0 --norm_world_scale 0",
              "--max_iters 1 --rg_refine 0",
              "--max_iters 1",
              "--max_iters 2 --rg_refine 0 --norm_world_scale 0 --fb_smooth 0",
              "--max_iters 2 --rg_refine 0 --norm_world_scale 0",
              "--max_iters 2", "--max_iters 4", "--max_iters 8"]:
    kernels.set_rand_epoch(0)
    g = pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=base + extra)
    o = orc.voldor(sc["flows"], fx, fy, cx, cy, config=base + extra)
    rot, tr = synth.pose_errors(g["poses"], o["poses"])
    m = (g["depth_conf"] > 0.5) & (o["depth_conf"] > 0.5)
    rel = np.abs(g["depth"][m] - o["depth"][m]) / o["depth"][m]
    print(extra, "| nreg", g["n_registered"], o["n_registered"], "rot", np.round(rot, 5), "tr", np.round(tr, 5),
          "depth agree(1e-3)", round(float(np.mean(rel < 1e-3)), 4), "conf diff", float(np.abs(g["depth_conf"] - o["depth_conf"]).mean()))
