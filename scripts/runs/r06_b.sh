#!/bin/bash
# Round 6: planned local runs (k_local_plan + k_local_runs_planned): identity tests, window hashes off / on, A/B per workload (fast and reference mode), kernel times
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r06b
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "local_runs or sample_pass" > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/${T}_pytest.log | tail -8
VOLDOR_HIP_DEBUG="local_plan=0" timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${T}_hash_off.txt 2>&1
timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${T}_hash_on.txt 2>&1
VOLDOR_HIP_DEBUG="local_plan=2" timeout 300 python scripts/window_hash.py cfg2 > gpurun_out/${T}_hash_force.txt 2>&1
grep -E "^cfg" gpurun_out/${T}_hash_off.txt gpurun_out/${T}_hash_on.txt gpurun_out/${T}_hash_force.txt
for wl in cfg3 cfg5; do
  timeout 700 python scripts/ab_config.py $wl "@local_plan=0" "" "@local_plan=0" "" > gpurun_out/${T}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_$wl.log
done
timeout 300 python scripts/ab_config.py cfg2 "@local_plan=0" "@local_plan=2" "@local_plan=0" "@local_plan=2" > gpurun_out/${T}_ab_cfg2.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_cfg2.log
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
timeout 900 python scripts/ab_config.py cfg3 "@local_plan=0 $R" "$R" > gpurun_out/${T}_ab_strict_cfg3.log 2>&1; grep -E "ms/window" gpurun_out/${T}_ab_strict_cfg3.log
for wl in cfg3 cfg5; do
  WL=$wl bash scripts/kstats.sh ${T}_$wl > gpurun_out/${T}_kstats_$wl.txt 2>&1; grep -E "k_local|k_cost_rand" gpurun_out/${T}_kstats_$wl.txt; rm -rf gpurun_out/ks_${T}_$wl
done
