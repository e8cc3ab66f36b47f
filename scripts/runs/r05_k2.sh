#!/bin/bash
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05k2
for v in 0 1; do
  VOLDOR_HIP_DEBUG="table_skip_equal=$v" WL=cfg5 bash scripts/kstats.sh ${TAG}_cfg5_skip$v > gpurun_out/${TAG}_kstats_cfg5_skip$v.txt 2>&1; grep -E "k_local_table|k_local_runs" gpurun_out/${TAG}_kstats_cfg5_skip$v.txt
  rm -rf gpurun_out/ks_${TAG}_cfg5_skip$v
done
