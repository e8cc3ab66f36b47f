#!/bin/bash
# Round 6: five-point bootstrap with a sample per workgroup: host bits, window A/B (8-point | 5-point), kernel times; order-of-profiling check for the cfg5 group
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r06g
timeout 900 python -m pytest tests/test_fivept.py tests/test_gpu_voldor.py -q -x > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error|rot " gpurun_out/${T}_pytest.log | tail -8
timeout 300 python scripts/ab_config.py cfg2 "" "--bootstrap_points 5" "" "--bootstrap_points 5" > gpurun_out/${T}_ab_cfg2.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_cfg2.log
bash scripts/kstats_cfg.sh ${T}_cfg2_5pt cfg2 "--bootstrap_points 5" > gpurun_out/${T}_kstats.txt 2>&1; grep -E "k_boot|k_extract|k_depth_closed" gpurun_out/${T}_kstats.txt; rm -rf gpurun_out/ks_${T}_cfg2_5pt
python - <<'PY'
import sys, ctypes as C
sys.path.insert(0, ".")
import numpy as np, torch
from voldor_amd import pyvoldor, synth, capi
import bench
lib = capi.lib()
ow = bench.WORKLOADS["cfg5"]
osc = synth.make_scene(w=ow["w"], h=ow["h"], n_flows=ow["n"], fx=ow["fx"], fy=ow["fx"], cx=ow["cx"], cy=ow["cy"], seed=233, basefocal=ow["basefocal"])
ofl = torch.from_numpy(osc["flows"]).cuda(); okw = dict(basefocal=ow["basefocal"], disparity=torch.from_numpy(osc["disparity"]).cuda())
od_, oc_ = torch.empty(ow["h"], ow["w"], device="cuda"), torch.empty(ow["h"], ow["w"], device="cuda")
run = lambda: pyvoldor.voldor_device(ofl, ow["fx"], ow["fx"], ow["cx"], ow["cy"], config=ow["cfg"], depth_out=od_, depth_conf_out=oc_, **okw)
for _ in range(3): run()
tot, cnt = C.c_double(0), C.c_long(0)
def prof(n=2):
    lib.vk_profile_enable(1)
    for _ in range(n): run()
    torch.cuda.synchronize()
    lib.vk_profile_get(b"optimize_depth", C.byref(tot), C.byref(cnt)); lib.vk_profile_enable(0)
    return tot.value / cnt.value * 1e3
sw = lib.vk_debug_switch; sw.argtypes = [C.c_char_p, C.c_int]
print("cfg5 group, profiled back to back:", [round(prof(), 1) for _ in range(3)])
sw(b"defer_reduce", 0); print("defer_reduce=0:", [round(prof(), 1) for _ in range(2)]); sw(b"defer_reduce", 1)
print("again default:", [round(prof(), 1) for _ in range(2)])
PY
