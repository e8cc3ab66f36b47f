#!/bin/bash
# Round 6: depth-ordered survivor queue in k_cost_rand_q: identity test, window hashes, A/B of two libraries (A = before), kernel times
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06d}
A=$PWD/voldor_amd/lib/libA.so
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "sample_pass or local_runs" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/${T}_pytest.log | tail -8
timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${T}_hash.txt 2>&1; grep -E "^cfg" gpurun_out/${T}_hash.txt
for wl in cfg2 cfg3 cfg5; do
  for rep in 1 2; do
    VOLDOR_HIP_LIB=$A timeout 300 python scripts/ab_config.py $wl "" 2>/dev/null | grep "ms/window" | tail -1 | sed 's/^/A /'
    timeout 300 python scripts/ab_config.py $wl "" 2>/dev/null | grep "ms/window" | tail -1 | sed 's/^/B /'
  done
done
for wl in ${KS_WL:-cfg2 cfg3 cfg5}; do
  VOLDOR_HIP_LIB=$A WL=$wl bash scripts/kstats.sh ${T}_A_$wl > gpurun_out/${T}_kstats_A_$wl.txt 2>&1; grep -E "${KS_GREP:-k_cost_rand|k_local_runs}" gpurun_out/${T}_kstats_A_$wl.txt | sed 's/^/A /'; rm -rf gpurun_out/ks_${T}_A_$wl
  WL=$wl bash scripts/kstats.sh ${T}_B_$wl > gpurun_out/${T}_kstats_B_$wl.txt 2>&1; grep -E "${KS_GREP:-k_cost_rand|k_local_runs}" gpurun_out/${T}_kstats_B_$wl.txt | sed 's/^/B /'; rm -rf gpurun_out/ks_${T}_B_$wl
done
