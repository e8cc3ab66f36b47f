#!/bin/bash
# Round 6: reference-mode kernel stats at cfg3 / cfg5 (cfg2 is in r06_z.sh)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06zz}
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in cfg3 cfg5; do
  bash scripts/kstats_cfg.sh ${T}_strict_$wl $wl "$R" > gpurun_out/ks_${T}_strict_$wl.txt 2>&1; head -16 gpurun_out/ks_${T}_strict_$wl.txt
  rm -rf gpurun_out/ks_${T}_strict_$wl
done
