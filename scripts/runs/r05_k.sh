#!/bin/bash
# Round 5: cost table without the entries of equal depth (compacted): identity, window hashes, A/B, kernel times
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05k
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "equal_depth or tiled_table or two_pixels" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | tail -8
VOLDOR_HIP_DEBUG="table_skip_equal=0" timeout 300 python scripts/window_hash.py cfg3 cfg5 > gpurun_out/${TAG}_hash_all.txt 2>&1; timeout 300 python scripts/window_hash.py cfg3 cfg5 > gpurun_out/${TAG}_hash_skip.txt 2>&1
grep -E "^cfg" gpurun_out/${TAG}_hash_all.txt gpurun_out/${TAG}_hash_skip.txt
for wl in cfg3 cfg5; do
  timeout 700 python scripts/ab_config.py $wl "@table_skip_equal=0" "" "@table_skip_equal=0" "" > gpurun_out/${TAG}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_$wl.log
done
for wl in cfg3 cfg5; do
  WL=$wl bash scripts/kstats.sh ${TAG}_${wl} > gpurun_out/${TAG}_kstats_${wl}.txt 2>&1; grep -E "k_update_rig|k_local_table|k_local_runs" gpurun_out/${TAG}_kstats_${wl}.txt
  rm -rf gpurun_out/ks_${TAG}_${wl}
done
