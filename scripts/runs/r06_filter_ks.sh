#!/bin/bash
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06f}
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in ${WLS:-cfg2 cfg3}; do
  bash scripts/kstats_cfg.sh ${T}_strict_$wl $wl "$R" > gpurun_out/ks_${T}_strict_$wl.txt 2>&1; head -9 gpurun_out/ks_${T}_strict_$wl.txt
  rm -rf gpurun_out/ks_${T}_strict_$wl
done
