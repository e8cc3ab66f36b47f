#!/bin/bash
# Round 4, final tree: the three bench lines (default workload first), then the whole GPU suite.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for wl in cfg2 cfg3 cfg5; do
  timeout 600 python bench.py --workload $wl > gpurun_out/bench_r04b_$wl.json 2> gpurun_out/bench_r04b_$wl.err
  echo "bench $wl rc=$?"; cut -c1-600 gpurun_out/bench_r04b_$wl.json
done
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu_r04b.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_r04b.log
