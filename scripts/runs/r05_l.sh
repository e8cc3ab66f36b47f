#!/bin/bash
# Round 5: k_collect with the weights of the trace product in flight together: window hashes, A/B, kernel time
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05l
VOLDOR_HIP_DEBUG="collect_prefetch=0" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_off.txt 2>&1; timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_on.txt 2>&1
grep -E "^cfg" gpurun_out/${TAG}_hash_off.txt gpurun_out/${TAG}_hash_on.txt
for wl in cfg2 cfg3 cfg5; do
  timeout 700 python scripts/ab_config.py $wl "@collect_prefetch=0" "" "@collect_prefetch=0" "" > gpurun_out/${TAG}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_$wl.log
done
for wl in cfg2 cfg5; do for v in 0 1; do
  VOLDOR_HIP_DEBUG="collect_prefetch=$v" WL=$wl bash scripts/kstats.sh ${TAG}_${wl}_$v > gpurun_out/${TAG}_kstats_${wl}_$v.txt 2>&1; grep -E "k_collect|k_solve" gpurun_out/${TAG}_kstats_${wl}_$v.txt
  rm -rf gpurun_out/ks_${TAG}_${wl}_$v
done; done
