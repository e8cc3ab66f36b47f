#!/bin/bash
# A/B two builds of libvoldor_hip.so on ONE box (boxes differ by a few percent): scripts/ab.sh <libA> <libB> [rounds]
# Prints frames/s of bench.py for A,B alternating, and the k_pose_refit / total GPU time from one rocprofv3 pass each.
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
A=$1; B=$2; R=${3:-3}
for i in $(seq $R); do
  for L in $A $B; do
    v=$(VOLDOR_HIP_LIB=$L python bench.py --steps 30 --warmup 3 --no-cpu-baseline --in-flight 0 2>/dev/null | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['value'])")
    echo "round $i $(basename $L) $v"
  done
done
for L in $A $B; do
  n=$(basename $L .so)
  VOLDOR_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ab_$n -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --in-flight 0 > /dev/null 2>&1
  f=$(ls -t $(find gpurun_out/ab_$n -name "*kernel_stats.csv") | head -1)
  python - "$f" "$n" <<'PY'
import csv,sys
tot=0; rows=[]
for r in csv.DictReader(open(sys.argv[1])):
    t=float(r['TotalDurationNs'])/28e6; tot+=t; rows.append((t,r['Name'].split('(')[0].replace('void ',''),float(r['AverageNs'])/1e3))
rows.sort(reverse=True)
print(sys.argv[2], f"GPU ms/window {tot:.3f} |", ' '.join(f"{n.replace('vk::','')}={a:.1f}us" for t,n,a in rows[:7]))
PY
done
