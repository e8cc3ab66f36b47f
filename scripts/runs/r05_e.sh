#!/bin/bash
# Round 5: quick A/B + per-launch trace of the fused pose launches (identity tests first)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r05e}
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "round5" > gpurun_out/${TAG}_pytest_first.log 2>&1; echo "identity rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|gave up" gpurun_out/${TAG}_pytest_first.log | tail -12
for wl in cfg2 cfg3; do
  timeout 300 python scripts/ab_config.py $wl "@pose_fused=0 @fb_overlap=0" "@pose_fused=1 @fb_overlap=0" "" > gpurun_out/${TAG}_ab_$wl.log 2>&1
  grep -E "ms/window|gave up" gpurun_out/${TAG}_ab_$wl.log
done
WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2 > gpurun_out/${TAG}_kstats_cfg2.txt 2>&1; head -12 gpurun_out/${TAG}_kstats_cfg2.txt
python - <<'PY'
import csv, glob, os
tag = os.environ.get("TAG", "r05e")
f = sorted(glob.glob(f"gpurun_out/ks_{tag}_cfg2/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
# last window: find the last bootstrap kernel
idx = max(i for i, n in enumerate(names) if "k_boot_hyp" in n)
seq = rows[idx:idx + 400]
out = []
prev_end = None
for r in seq:
    n = r["Kernel_Name"].split("(")[0].replace("void vk::", "").replace("vk::", "")[:28]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    prev_end = e
    out.append(f"{n:28s} {(e - s) / 1e3:8.2f} us  gap {gap:6.2f}")
    if "k_pack_pose" in n: break
open(f"gpurun_out/{tag}_window_trace.txt", "w").write("\n".join(out))
print("\n".join(out[4:64]))
PY
