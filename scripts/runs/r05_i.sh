#!/bin/bash
# Round 5: the E-step with two pixels per lane on packed fp32 ("not built is not killed"): identity, window hashes, A/B, kernel time
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05i
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "estep or update_rigidness" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | tail -5
timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_one.txt 2>&1; VOLDOR_HIP_DEBUG="estep_pairs=1" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${TAG}_hash_pairs.txt 2>&1
grep -E "^cfg" gpurun_out/${TAG}_hash_one.txt gpurun_out/${TAG}_hash_pairs.txt
for wl in cfg2 cfg3 cfg5; do
  timeout 500 python scripts/ab_config.py $wl "" "@estep_pairs=1" > gpurun_out/${TAG}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_$wl.log
done
for wl in cfg3 cfg5; do
  VOLDOR_HIP_DEBUG="estep_pairs=1" WL=$wl bash scripts/kstats.sh ${TAG}_${wl}_pairs > gpurun_out/${TAG}_kstats_${wl}_pairs.txt 2>&1; grep -E "k_update_rig" gpurun_out/${TAG}_kstats_${wl}_pairs.txt
  rm -rf gpurun_out/ks_${TAG}_${wl}_pairs
done
