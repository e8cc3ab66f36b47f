#!/bin/bash
# per-kernel time of one bench run: scripts/kstats.sh <tag>   (env WL, VOLDOR_HIP_LEAN as in bench/ab scripts)
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=$1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- python bench.py --workload ${WL:-cfg2} --steps 5 --warmup 2 --no-extras > gpurun_out/ks_$tag.log 2>&1
f=$(ls -t $(find gpurun_out/ks_$tag -name "*kernel_stats.csv") | head -1)
cp "$f" gpurun_out/ks_$tag.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:18]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
tail -1 gpurun_out/ks_$tag.log | cut -c1-300
