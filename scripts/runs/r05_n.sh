#!/bin/bash
# Round 5: the cost table without the entries of equal depth, strict (reference) mode: identity, the strict suites, reference-mode windows A/B, kernel times
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
TAG=r05n
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "equal_depth" > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest.log | tail -8
timeout 1200 python -m pytest tests/test_gpu_strict.py tests/test_gpu_vs_ref_window.py tests/test_gpu_configs.py -m gpu -q > gpurun_out/${TAG}_pytest_strict.log 2>&1; echo "strict rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${TAG}_pytest_strict.log | tail -8
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in cfg2 cfg3 cfg5; do
  timeout 900 python scripts/ab_config.py $wl "$R @table_skip_equal=0" "$R" > gpurun_out/${TAG}_ab_$wl.log 2>&1; grep -E "ms/window" gpurun_out/${TAG}_ab_$wl.log
done
for v in 0 1; do
  VOLDOR_HIP_DEBUG="table_skip_equal=$v" bash scripts/kstats_cfg.sh ${TAG}_strict_cfg2_$v cfg2 "$R" > gpurun_out/${TAG}_kstats_strict_cfg2_$v.txt 2>&1; grep -E "k_local_table|k_local_runs" gpurun_out/${TAG}_kstats_strict_cfg2_$v.txt
  rm -rf gpurun_out/ks_${TAG}_strict_cfg2_$v
done
