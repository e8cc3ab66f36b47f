#!/bin/bash
# Round 5: the runs kernel held to 5 / 6 waves per SIMD (96 / 80 registers + scratch) against 4 (121-127 registers): window hashes, windows, kernel time
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05p
for L in "" _w5 _w6; do
  export VOLDOR_HIP_LIB=$PWD/voldor_amd/lib/libvoldor_hip$L.so
  echo "== lib$L"; timeout 300 python scripts/window_hash.py cfg2 cfg5 2>&1 | grep ^cfg
  for wl in cfg2 cfg3 cfg5; do timeout 300 python scripts/ab_config.py $wl "" 2>&1 | grep "ms/window"; done
  WL=cfg5 bash scripts/kstats.sh ${TAG}_cfg5$L 2>&1 | grep -E "k_local_runs"; rm -rf gpurun_out/ks_${TAG}_cfg5$L
  WL=cfg2 bash scripts/kstats.sh ${TAG}_cfg2$L 2>&1 | grep -E "k_local_runs"; rm -rf gpurun_out/ks_${TAG}_cfg2$L
done
