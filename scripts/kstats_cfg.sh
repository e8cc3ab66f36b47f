#!/bin/bash
# per-kernel time of one workload under a config suffix: scripts/kstats_cfg.sh <tag> <workload> "<config suffix>"
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=$1; wl=$2; sfx=$3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/ks_$tag -- python scripts/ab_config.py $wl "$sfx" > gpurun_out/ks_$tag.log 2>&1
f=$(ls -t $(find gpurun_out/ks_$tag -name "*kernel_stats.csv") | head -1)
cp "$f" gpurun_out/ks_$tag.csv
python - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:14]:
    print(f'{r["Name"][:60]:60s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.2f} us  {100*float(r["TotalDurationNs"])/tot:5.1f} %')
PY
tail -1 gpurun_out/ks_$tag.log
