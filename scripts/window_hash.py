"""sha256 of one window's outputs (depth, confidence, poses, covariances) per workload: two builds that claim identical arithmetic must print the
same lines.  usage: [VOLDOR_HIP_LIB=...] python scripts/window_hash.py cfg2 cfg3 cfg5"""
import hashlib, sys
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
    o = pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"], depth_out=depth, depth_conf_out=conf, **kw)
    h = hashlib.sha256()
    for a in (depth.cpu().numpy(), conf.cpu().numpy(), np.asarray(o["poses"], np.float32), np.asarray(o["poses_covar"], np.float32)):
        h.update(np.ascontiguousarray(a).tobytes())
    print(name, o["n_registered"], h.hexdigest()[:24])
