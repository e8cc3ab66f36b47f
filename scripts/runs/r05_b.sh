#!/bin/bash
# Round 5, second GPU call: cost of events / cross-stream joins / cooperative launches (micro), the whole suite on the rounding-compatible fast P3P, A/B.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 120 scripts/micro/event_cost > gpurun_out/r05b_event_cost.txt 2>&1; cat gpurun_out/r05b_event_cost.txt
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/r05b_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r05b_pytest.log | tail -25
for wl in cfg2 cfg3; do
  timeout 300 python scripts/ab_config.py $wl "@fb_overlap=0 @solve_fp32=0" "@fb_overlap=1 @solve_fp32=0" "@fb_overlap=0 @solve_fp32=1" "" > gpurun_out/r05b_ab_$wl.log 2>&1
  grep "ms/window" gpurun_out/r05b_ab_$wl.log
done
WL=cfg2 bash scripts/kstats.sh r05b_cfg2 > gpurun_out/r05b_kstats_cfg2.txt 2>&1; head -8 gpurun_out/r05b_kstats_cfg2.txt
