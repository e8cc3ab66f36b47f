#!/bin/bash
# Round 5: small kernel-level experiments behind switches (own table entry with all gathers in flight; LDS padding of k_solve = workgroups per CU)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=${TAG:-r05f}
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "own_table_entry" > gpurun_out/${TAG}_pytest_first.log 2>&1; echo "identity rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest_first.log | tail -6
timeout 400 python scripts/ab_config.py cfg2 "@runs_table_ch=0" "" "@solve_lds_pad_kb=35" "@solve_lds_pad_kb=48" "@solve_lds_pad_kb=55" > gpurun_out/${TAG}_ab_cfg2.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_cfg2.log
timeout 400 python scripts/ab_config.py cfg3 "@runs_table_ch=0" "" "@solve_lds_pad_kb=48" > gpurun_out/${TAG}_ab_cfg3.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_cfg3.log
WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2 > gpurun_out/${TAG}_kstats_cfg2.txt 2>&1; head -12 gpurun_out/${TAG}_kstats_cfg2.txt
WL=cfg5 bash scripts/kstats.sh ${TAG}_cfg5 > gpurun_out/${TAG}_kstats_cfg5.txt 2>&1; head -14 gpurun_out/${TAG}_kstats_cfg5.txt
python - <<'PY'
import csv, glob, os, collections
tag = os.environ.get("TAG", "r05f")
f = sorted(glob.glob(f"gpurun_out/ks_{tag}_cfg5/**/*kernel_trace.csv", recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [r["Kernel_Name"] for r in rows]
idx = max(i for i, n in enumerate(names) if "k_pack_pose" in n)   # end of the last window
start = max(i for i, n in enumerate(names[:idx]) if "k_pack_pose" in n) + 1 if any("k_pack_pose" in n for n in names[:idx]) else 0
seq = rows[start:idx + 1]
out = []; it = 0; byk = collections.defaultdict(list)
for r in seq:
    n = r["Kernel_Name"].split("(")[0].replace("void vk::", "").replace("vk::", "")[:34]
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if "k_cost_rand_q" in n: it += 1
    if any(t in n for t in ("k_local_runs", "k_cost_rand_q", "k_local_table", "k_global_prop", "k_update_rig", "k_fb_")):
        out.append(f"it {it:2d} {n:34s} {d:8.1f} us"); byk[n].append(d)
open(f"gpurun_out/{tag}_cfg5_depth_trace.txt", "w").write("\n".join(out))
for n, v in byk.items():
    print(f"{n:34s} n {len(v):3d} " + " ".join(f"{x:6.0f}" for x in v[:26]))
PY
