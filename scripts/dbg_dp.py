import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import orc
from voldor_amd import pyvoldor, synth, kernels
sc = synth.make_scene(w=160, h=120, n_flows=3, fx=80, fy=80, cx=80, cy=60, seed=237)
fx, fy, cx, cy = sc["K"]
pri = np.stack([sc["depth_gt"], sc["depth_gt"] * 1.01]).astype(np.float32)
poses = np.array([[0, 0, 0, 0, 0, 0], [0.001, 0, 0, 0.01, 0, 0]], np.float32)
pc = np.full_like(pri, 0.8)
for it in (1, 2, 3, 4):
    cfg = f"--silent --max_iters {it} --delta 0.5"
    for pconfs in (None, pc):
        kernels.set_rand_epoch(0)
        g = pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, depth_priors=pri, depth_prior_poses=poses, depth_prior_pconfs=pconfs, basefocal=40.0, config=cfg)
        o = orc.voldor(sc["flows"], fx, fy, cx, cy, depth_priors=pri, depth_prior_poses=poses, depth_prior_pconfs=pconfs, basefocal=40.0, config=cfg)
        rot, tr = synth.pose_errors(g["poses"], o["poses"])
        print(it, pconfs is not None, g["n_registered"], o["n_registered"], "rot", rot.max(), "tr", tr, "t_o", np.linalg.norm(o["poses"][:, 3:], axis=1))
