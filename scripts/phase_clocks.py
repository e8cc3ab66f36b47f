"""Phase clocks of the pose kernels (library built by scripts/phase_clocks.sh with -DVK_PHASE_CLOCKS): runs a few windows of a bench
workload and prints, per kernel, the s_memtime ticks (100 MHz constant clock on gfx950) between the marks of workgroup 0 / thread 0."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch

import bench
from voldor_amd import capi, pyvoldor, synth

wl = dict(bench.WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"])
wl["cfg"] += " " + os.environ.get("CFG_SUFFIX", "")
W, H, N = wl["w"], wl["h"], wl["n"]
sc = synth.make_scene(w=W, h=H, n_flows=N, fx=wl["fx"], fy=wl["fx"], cx=wl["cx"], cy=wl["cy"], seed=233,
                      basefocal=wl["basefocal"] if wl["mode"] != "mono" else 0.0)
flows = torch.from_numpy(sc["flows"]).cuda()
extra = dict(basefocal=wl["basefocal"], disparity=torch.from_numpy(sc["disparity"]).cuda()) if wl["mode"] == "stereo" else {}
lib = capi.lib()
lib.vk_phase_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
buf = (C.c_ulonglong * 64)()
for _ in range(3):
    pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"], **extra)
assert lib.vk_phase_read(buf, 64, 1) == 0
lib.vk_phase_read_depth.argtypes = [C.POINTER(C.c_ulonglong), C.c_int, C.c_int]
bufd = (C.c_ulonglong * 64)()
assert lib.vk_phase_read_depth(bufd, 64, 1) == 0
NW = 10
t0 = time.perf_counter()
for _ in range(NW):
    pyvoldor.voldor_device(flows, wl["fx"], wl["fx"], wl["cx"], wl["cy"], config=wl["cfg"], **extra)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / NW
assert lib.vk_phase_read(buf, 64, 0) == 0
v = np.array(list(buf), dtype=np.float64)
print(f"window {dt*1e3:.2f} ms (instrumented build)")
def show(name, slots, calls_slot):
    calls = max(v[calls_slot], 1)
    print(f"{name}: calls/window {calls/NW:.1f}")
    for label, s in slots:
        print(f"   {label:28s} {v[s]/calls:10.1f} ticks/call   {v[s]/NW:10.0f} ticks/window")
show("k_collect (wg 0)", [("whole", 0)], 1)
show("k_solve (wg 0, lane 0)", [("count blk_counts", 8), ("draw", 9), ("load 4 points", 10), ("p4p", 11), ("nearest_rot + angle axis", 12), ("fold + store", 13)], 14)
show("k_pose_mode", [("load hypotheses", 16), ("init / trials", 17), ("mean-shift: pass", 32), ("mean-shift: reduce + update", 33), ("finalize_pose", 21), ("decide + tail", 22)], 20)
print(f"   mean-shift iterations / call {v[19]/max(v[20],1):.2f}")
show("k_pose_refit", [("stage", 24), ("prepare (inverse)", 25), ("sample pass", 26), ("all-reduce + M-step", 27), ("finalize", 30)], 29)
print(f"   gate iterations / call {v[28]/max(v[29],1):.2f}; pair-slots passed / iteration {v[34]/max(v[28],1):.2f} of 8")

assert lib.vk_phase_read_depth(bufd, 64, 0) == 0
v = np.array(list(bufd), dtype=np.float64)
show("k_local_runs_lean (middle wave: two chains)", [("setup + loads", 0), ("evaluation rounds", 1), ("bookkeeping", 2), ("exit", 3)], 8)
print(f"   rounds / call {v[6]/max(v[8],1):.2f}")
h = v[16:48]
print("   rounds histogram (chains per launch): " + " ".join(f"{i}:{h[i]/(NW*32):.1f}" for i in range(32) if h[i] > 0))

