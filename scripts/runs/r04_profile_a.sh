# final-tree counters and kernel stats (round 4): COMMIT=<hash> bash scripts/runs/r04_profile_a.sh
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for wl in cfg2 cfg3 cfg5; do
  WL=$wl bash scripts/kstats.sh r04_$wl > gpurun_out/ks_r04_$wl.txt 2>&1
  WL=$wl bash scripts/pmc_traffic.sh r04_$wl > gpurun_out/pmc_traffic_r04_$wl.txt 2>&1
  WL=$wl bash scripts/pmc_sq.sh r04_$wl > gpurun_out/pmc_sq_r04_$wl.txt 2>&1
  tail -2 gpurun_out/ks_r04_$wl.txt | cut -c1-200
  rm -rf gpurun_out/ks_r04_$wl gpurun_out/pmc_FETCH_SIZE_r04_$wl gpurun_out/pmc_WRITE_SIZE_r04_$wl gpurun_out/pmc_sq_r04_$wl   # raw rocprofv3 directories: the summaries above are what is kept (gpurun_out/ is capped at 64 MiB)
done
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
bash scripts/kstats_cfg.sh r04_strict_cfg2 cfg2 "$R" > gpurun_out/ks_r04_strict_cfg2.txt 2>&1; head -12 gpurun_out/ks_r04_strict_cfg2.txt
bash scripts/kstats_cfg.sh r04_strict_cfg3 cfg3 "$R" > gpurun_out/ks_r04_strict_cfg3.txt 2>&1; tail -1 gpurun_out/ks_r04_strict_cfg3.txt
rm -rf gpurun_out/ks_r04_strict_cfg2 gpurun_out/ks_r04_strict_cfg3
du -sh gpurun_out; ls gpurun_out | head -40
