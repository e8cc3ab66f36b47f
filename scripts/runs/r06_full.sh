#!/bin/bash
# Round 6: whole GPU suite, smoke, default bench
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6
python __graft_entry__.py smoke 2>&1 | tail -2
timeout 900 python bench.py > gpurun_out/bench_r06y.json 2> gpurun_out/bench_r06y.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/bench_r06y.json
