#!/bin/bash
# reference-mode windows of the three configs (+ the strict test files with TESTS=1)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
if [ -n "$TESTS" ]; then timeout 1500 python -m pytest tests/test_gpu_strict_filter.py tests/test_gpu_strict.py tests/test_gpu_configs.py -x -q 2>&1 | tail -3; fi
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in ${WLS:-cfg2 cfg3 cfg5}; do timeout 600 python scripts/ab_config.py $wl "$R" 2>&1 | tail -1; done
if [ -n "$KS" ]; then WLS="$KS" bash scripts/runs/r06_filter_ks.sh 2>&1 | grep "runs\|table\|global"; fi
