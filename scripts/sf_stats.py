"""What the fp32 filters of the strict sample pass saw and kept over one reference-mode window (vk_debug_counter "sf_*").  usage: python scripts/sf_stats.py"""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, torch, hooks, bench
from voldor_amd import pyvoldor, synth, kernels
for name in ("cfg2","cfg3"):
    wl=bench.WORKLOADS[name]
    sc = synth.make_scene(w=wl["w"], h=wl["h"], n_flows=wl["n"], fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=233, basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
    kw = dict(basefocal=wl["basefocal"], disparity=sc["disparity"]) if wl["mode"] == "stereo" else {}
    hooks.debug_switch("strict_filter", 2)
    for k in ("sf_samples","sf_sample_survivors"): hooks.debug_counter(k)
    o=pyvoldor.voldor(sc["flows"], wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"]+" --strict_math 1 --reference_draw 1 --reference_svd 1", **kw)
    st={k:hooks.debug_counter(k) for k in ("sf_samples","sf_sample_survivors")}
    print(name, st, "sample survivors", st["sf_sample_survivors"]/max(st["sf_samples"],1))
