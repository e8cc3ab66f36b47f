import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from voldor_amd import pyvoldor, synth, capi
from oracle import orc
sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320, fy=320, cx=320, cy=240, seed=233)
fx, fy, cx, cy = sc["K"]
cfg = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8"
g = pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=cfg)
n = 5
cnt = (C.c_int * n)(); dens = (C.c_float * n)(); rd = (C.c_float * n)(); ms = (C.c_int * n)(); gu = (C.c_int * n)()
capi.lib().vk_last_camera_stats(cnt, dens, rd, ms, gu, n)
print("pool", list(cnt), "density", [round(x, 4) for x in dens], "rig dens", [round(x, 3) for x in rd], "ms_iters", list(ms), "gu_iters", list(gu))
