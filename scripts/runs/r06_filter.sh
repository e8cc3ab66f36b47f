#!/bin/bash
# Round 6: the fp32 pre-filter of the strict sample pass and the pow(x, -2) shortcut -- tests, reference-mode windows with the filters on / off, kernel stats
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
if [ -z "$NOTEST" ]; then timeout 1500 python -m pytest tests/test_gpu_strict_filter.py tests/test_gpu_strict.py tests/test_gpu_configs.py -x -q 2>&1 | tail -5; fi
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
for wl in cfg2 cfg3 cfg5; do
  timeout 600 python scripts/ab_config.py $wl "$R" "$R @strict_table_filter=0" "$R @strict_filter=0" 2>&1 | tail -3
done
WLS="${WLS:-cfg2 cfg3}" bash scripts/runs/r06_filter_ks.sh 2>&1 | grep "cost_rand\|k_cost_strict\|table"
python scripts/sf_stats.py 2>&1 | tail -2
