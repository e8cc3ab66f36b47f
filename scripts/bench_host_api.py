"""PCIe-inclusive rate of the host-pointer call pyvoldor.voldor (flows in pageable host memory, depth maps returned to the host)
next to the device-resident call, for DESIGN.md section 6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from voldor_amd import pyvoldor, synth
sc = synth.make_scene(w=640, h=480, n_flows=5, fx=320, fy=320, cx=320, cy=240, seed=233)
CONFIG = "--silent --meanshift_kernel_var 0.2 --delta 1.5 --max_iters 8"
fl_host = sc["flows"]
fl_dev = torch.from_numpy(fl_host).cuda()
d = torch.empty(480, 640, device="cuda"); c = torch.empty(480, 640, device="cuda")
for name, fn in (("device-resident (vk_voldor_device)", lambda: pyvoldor.voldor_device(fl_dev, 320, 320, 320, 240, config=CONFIG, depth_out=d, depth_conf_out=c)),
                 ("host pointers (py_voldor_wrapper) ", lambda: pyvoldor.voldor(fl_host, 320, 320, 320, 240, config=CONFIG))):
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 0.5: fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(30): fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 30
    print(f"{name}: {dt*1e3:.3f} ms per window = {1/dt:.1f} frames/s")
