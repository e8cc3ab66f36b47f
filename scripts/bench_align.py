"""Timing of the frame-alignment maps (align_frame_eval_gpu) and of eval_covisibility at 640x480, for DESIGN.md."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
from align_scene import keyframes
from voldor_amd import kernels, slam_utils
kf = keyframes(w=640, h=480, n=4, seed=1)
rc, shape = kernels.align_frame_init_gpu(kf["images"], kf["depths"], kf["weights"], kf["K"], kf["vbf"], kf["crw"])
for want_j in (True, False):
    for _ in range(3): kernels.align_frame_eval_gpu(shape, 0, 1, kf["params"][0], kf["params"][1], want_j)
    t0 = time.perf_counter()
    for _ in range(50): kernels.align_frame_eval_gpu(shape, 0, 1, kf["params"][0], kf["params"][1], want_j)
    print("align_frame_eval_gpu 640x480 jacobian=%s: %.1f us per call incl. D2H of the maps" % (want_j, (time.perf_counter() - t0) / 50 * 1e6))
d = torch.from_numpy(kf["depths"][0]).cuda()
T = np.eye(4, dtype=np.float32); T[2, 3] = 0.3
for _ in range(3): slam_utils.eval_covisibility(d, T, kf["K"])
t0 = time.perf_counter()
for _ in range(200): slam_utils.eval_covisibility(d, T, kf["K"])
print("eval_covisibility 640x480 stride 4, depth in HBM: %.1f us per call" % ((time.perf_counter() - t0) / 200 * 1e6))
