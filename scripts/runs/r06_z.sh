#!/bin/bash
# Round 6, final tree: kernel stats + both PMC passes (cfg2 / cfg3 / cfg5, reference-mode kernel stats), copied to profiles/r06z_* ON THE BOX so that the
# bench lines that follow replay counters of exactly these kernel sources; the GPU suite; the three bench lines.  COMMIT=<hash> bash scripts/runs/r05_z.sh
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06z}
for wl in cfg2 cfg3 cfg5; do
  WL=$wl bash scripts/kstats.sh ${T}_$wl > gpurun_out/ks_${T}_$wl.txt 2>&1
  WL=$wl bash scripts/pmc_traffic.sh ${T}_$wl > gpurun_out/pmc_traffic_${T}_$wl.txt 2>&1
  WL=$wl bash scripts/pmc_sq.sh ${T}_$wl > gpurun_out/pmc_sq_${T}_$wl.txt 2>&1
  tail -2 gpurun_out/ks_${T}_$wl.txt | cut -c1-200
  rm -rf gpurun_out/ks_${T}_$wl gpurun_out/pmc_FETCH_SIZE_${T}_$wl gpurun_out/pmc_WRITE_SIZE_${T}_$wl gpurun_out/pmc_sq_${T}_$wl
  cp gpurun_out/ks_${T}_$wl.csv profiles/${T}_kernel_stats_$wl.csv; cp gpurun_out/pmc_traffic_${T}_$wl.json profiles/${T}_pmc_traffic_$wl.json; cp gpurun_out/pmc_sq_${T}_$wl.json profiles/${T}_pmc_sq_$wl.json
done
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
bash scripts/kstats_cfg.sh ${T}_strict_cfg2 cfg2 "$R" > gpurun_out/ks_${T}_strict_cfg2.txt 2>&1; head -12 gpurun_out/ks_${T}_strict_cfg2.txt
cp gpurun_out/ks_${T}_strict_cfg2.csv profiles/${T}_strict_kernel_stats_cfg2.csv; rm -rf gpurun_out/ks_${T}_strict_cfg2
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_pytest.log | tail -12
for wl in cfg2 cfg3 cfg5; do
  timeout 900 python bench.py --workload $wl > gpurun_out/bench_${T}_$wl.json 2> gpurun_out/bench_${T}_$wl.err
  echo "bench $wl rc=$?"; cut -c1-330 gpurun_out/bench_${T}_$wl.json
done
python __graft_entry__.py smoke 2>&1 | tail -3
