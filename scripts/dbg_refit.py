import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, ctypes as C
from voldor_amd import pyvoldor, synth, capi
sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320, fy=320, cx=320, cy=240, seed=233)
fx, fy, cx, cy = sc["K"]
lib = capi.lib()
for K in (1, 5, 10, 20, 40, 0):
    cfg = f"--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 1 --rg_max_iters {K}" + (" --rg_refine 0" if K == 0 else "")
    pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=cfg)
    lib.vk_profile_enable(1)
    for _ in range(3):
        pyvoldor.voldor(sc["flows"], fx, fy, cx, cy, config=cfg)
    tot, cnt = C.c_double(0), C.c_long(0)
    lib.vk_profile_get(b"optimize_camera_pose", C.byref(tot), C.byref(cnt))
    lib.vk_profile_enable(0)
    n = 5
    gu = (C.c_int * n)()
    lib.vk_last_camera_stats(None, None, None, None, gu, n)
    print("rg_max_iters", K, "camera pose group avg us", round(tot.value / cnt.value * 1e3, 1), "gu_iters", list(gu))
