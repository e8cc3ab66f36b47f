#!/bin/bash
# usage: scripts/pmc_pass.sh <tag> <counter> [<counter> ...]   (one rocprofv3 --pmc pass per call, kernel-trace only)
# env WL=cfg2|cfg3|cfg5 selects the bench workload
cd "$GRAFT_REPO_ROOT" && export TMPDIR=/tmp
tag=$1; shift
rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d gpurun_out/pmc_$tag -- python bench.py --workload ${WL:-cfg2} --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1
f=$(ls -t $(find gpurun_out/pmc_$tag -name "*counter_collection.csv") | head -1)
python - "$f" <<'PY'
import csv,sys,collections
acc=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
seen=set()
for r in csv.DictReader(open(sys.argv[1])):
    k=r['Kernel_Name'].split('(')[0].replace('void ','')
    acc[k][r['Counter_Name']]+=float(r['Counter_Value'])
    key=(r['Dispatch_Id'])
    if key not in seen: seen.add(key); cnt[k]+=1
for k in sorted(acc, key=lambda k:-cnt[k]):
    print(k[:44].ljust(44), cnt[k], ' '.join(f"{c}={v/cnt[k]:.4g}" for c,v in acc[k].items()))
PY
