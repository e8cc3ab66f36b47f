#!/bin/bash
# Round 4, final tree: kernel stats + both PMC passes (cfg2/cfg3/cfg5, reference-mode kernel stats), copied to profiles/r04b_* ON THE BOX so that the
# bench lines that follow replay counters of exactly these kernel sources; then the three bench lines.  COMMIT=<hash> bash scripts/runs/r04_final_b.sh
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
bash scripts/runs/r04_profile_a.sh
for wl in cfg2 cfg3 cfg5; do
  cp gpurun_out/ks_r04_$wl.csv profiles/r04b_kernel_stats_$wl.csv; cp gpurun_out/pmc_traffic_r04_$wl.json profiles/r04b_pmc_traffic_$wl.json; cp gpurun_out/pmc_sq_r04_$wl.json profiles/r04b_pmc_sq_$wl.json
done
cp gpurun_out/ks_r04_strict_cfg2.csv profiles/r04b_strict_kernel_stats_cfg2.csv; cp gpurun_out/ks_r04_strict_cfg3.csv profiles/r04b_strict_kernel_stats_cfg3.csv
for wl in cfg2 cfg3 cfg5; do
  timeout 600 python bench.py --workload $wl > gpurun_out/bench_r04b_$wl.json 2> gpurun_out/bench_r04b_$wl.err
  echo "bench $wl rc=$?"; cut -c1-300 gpurun_out/bench_r04b_$wl.json
done
