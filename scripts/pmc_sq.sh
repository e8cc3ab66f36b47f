#!/bin/bash
# Raw SQ counters per kernel (one rocprofv3 --pmc pass, kernel-trace only; 8 SQ slots on gfx950).  usage: scripts/pmc_sq.sh <tag>
# env: WL=cfg2|cfg3|cfg5, VOLDOR_HIP_LEAN=0|1.  Prints per-launch averages and writes gpurun_out/pmc_sq_<tag>.json
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=$1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES \
  --output-format csv -d gpurun_out/pmc_sq_$tag -- python bench.py --workload ${WL:-cfg2} --steps 3 --warmup 1 --no-extras > gpurun_out/pmc_sq_$tag.log 2>&1
f=$(ls -t $(find gpurun_out/pmc_sq_$tag -name "*counter_collection.csv") | head -1)
k=$(ls -t $(find gpurun_out/pmc_sq_$tag -name "*kernel_trace.csv") | head -1)
python - "$f" "$k" "gpurun_out/pmc_sq_$tag.json" <<'PY'
import csv, sys, collections, json
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0].replace('void ', '')
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen: seen.add(r['Dispatch_Id']); cnt[k] += 1
dur = collections.defaultdict(float); dn = collections.Counter()
try:
    for r in csv.DictReader(open(sys.argv[2])):
        k = r['Kernel_Name'].split('(')[0].replace('void ', '')
        dur[k] += (float(r['End_Timestamp']) - float(r['Start_Timestamp'])) / 1e3; dn[k] += 1
except Exception as e:
    print("no kernel trace:", e)
out = {}
for k in sorted(acc, key=lambda k: -acc[k].get('SQ_BUSY_CYCLES', 0)):
    v = {c: acc[k][c] / cnt[k] for c in acc[k]}
    v['launches'] = cnt[k]
    if dn[k]: v['avg_us_under_pmc'] = dur[k] / dn[k]
    # SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves (MI355X_MICROARCH.md)
    wc = v.get('SQ_WAVE_CYCLES', 0)
    if wc:
        v['frac_active_valu'] = v.get('SQ_ACTIVE_INST_VALU', 0) / wc
        v['frac_wait_any'] = v.get('SQ_WAIT_ANY', 0) / wc
        v['frac_wait_inst_any'] = v.get('SQ_WAIT_INST_ANY', 0) / wc
    if v.get('SQ_WAVES'): v['valu_insts_per_wave'] = v.get('SQ_INSTS_VALU', 0) / v['SQ_WAVES']
    out[k] = v
for k, v in list(out.items())[:16]:
    print(k[:40].ljust(40), ' '.join(f"{c}={x:.4g}" for c, x in v.items()))
import os
out['commit'] = os.environ.get('COMMIT')
import hashlib
h = hashlib.sha256()
for f in ("vk_depth.hip", "vk_depth_impl.hpp", "vk_fb.hpp", "vk_cum_poses.hpp", "vk_pose.hip", "vk_device.hpp", "vk_p3p.hpp", "vk_common.hpp"):  # = bench.KERNEL_SOURCES: bench.py withholds a pass taken on other sources
    h.update(open(os.path.join("voldor_amd", "csrc", f), "rb").read())
out['kernel_source_sha256'] = h.hexdigest()[:16]  # the tree the counters were collected on (bench.py quotes it as the source of its replayed figures)
json.dump(out, open(sys.argv[3], 'w'), indent=1)
PY
