"""Time one workload under several config suffixes (windows back to back, inputs resident).  usage: python scripts/ab_config.py cfg2 "" "--reference_draw 1" ..."""
import sys, time
sys.path.insert(0, ".")
import numpy as np, torch
from voldor_amd import pyvoldor, synth, kernels
import bench
wl = bench.WORKLOADS[sys.argv[1]]
sc = synth.make_scene(w=wl["w"], h=wl["h"], n_flows=wl["n"], fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=233, basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
flows = torch.from_numpy(sc["flows"]).cuda()
kw = dict(basefocal=wl["basefocal"], disparity=torch.from_numpy(sc["disparity"]).cuda()) if wl["mode"] == "stereo" else {}
depth = torch.empty(wl["h"], wl["w"], device="cuda"); conf = torch.empty_like(depth)
n = 30 if wl["w"] < 1900 else 8
from voldor_amd import capi
def switches(sfx):  # "@name=value" tokens of a suffix set verification switches (vk_debug.h) for that variant; the rest is config text
    toks = sfx.split()
    for t in toks:
        if t.startswith("@"):
            k, v = t[1:].split("=")
            assert capi.lib().vk_debug_switch(k.encode(), int(v)) >= 0, t
    return " ".join(t for t in toks if not t.startswith("@"))
for rep in range(2):
    for raw in sys.argv[2:]:
        for k, v in (("strict_plain", 0), ("strict_pose_coop", 1), ("estep_pairs", 1), ("fb_ride", 1), ("defer_reduce", 1), ("fb_segment", 0), ("fb_side", 1), ("strict_filter", 1), ("strict_table_filter", 1)):
            capi.lib().vk_debug_switch(k.encode(), v)
        sfx = switches(raw)
        for _ in range(3): pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"] + " " + sfx, depth_out=depth, depth_conf_out=conf, **kw)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(n): o = pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"] + " " + sfx, depth_out=depth, depth_conf_out=conf, **kw)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{sys.argv[1]} [{raw:24s}] {dt*1e3:8.3f} ms/window  {1/dt:7.1f} windows/s  n_registered {o['n_registered']}")
