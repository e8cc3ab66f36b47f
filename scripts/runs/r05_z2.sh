#!/bin/bash
# Round 5, final tree, second part: the two PMC passes per workload (the first attempt named a deleted source in its hash list), copied to profiles/ on the box, then the bench lines
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r05z
for wl in cfg2 cfg3 cfg5; do
  WL=$wl bash scripts/pmc_traffic.sh ${T}_$wl > gpurun_out/pmc_traffic_${T}_$wl.txt 2>&1
  WL=$wl bash scripts/pmc_sq.sh ${T}_$wl > gpurun_out/pmc_sq_${T}_$wl.txt 2>&1
  rm -rf gpurun_out/pmc_FETCH_SIZE_${T}_$wl gpurun_out/pmc_WRITE_SIZE_${T}_$wl gpurun_out/pmc_sq_${T}_$wl
  cp gpurun_out/pmc_traffic_${T}_$wl.json profiles/${T}_pmc_traffic_$wl.json; cp gpurun_out/pmc_sq_${T}_$wl.json profiles/${T}_pmc_sq_$wl.json
  tail -2 gpurun_out/pmc_traffic_${T}_$wl.txt | cut -c1-200
done
for wl in cfg2 cfg3 cfg5; do
  timeout 900 python bench.py --workload $wl > gpurun_out/bench_${T}_$wl.json 2> gpurun_out/bench_${T}_$wl.err
  echo "bench $wl rc=$?"; cut -c1-200 gpurun_out/bench_${T}_$wl.json
done
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
