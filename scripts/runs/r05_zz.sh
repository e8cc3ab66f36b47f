#!/bin/bash
# Round 5, last tree: the GPU suite, smoke, and the driver's default bench command with its wall time
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r05zz
timeout 1700 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_pytest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
/usr/bin/time -v python bench.py > gpurun_out/bench_${T}_default.json 2> gpurun_out/bench_${T}_default.err; echo "bench rc=$?"; grep -E "Elapsed \(wall" gpurun_out/bench_${T}_default.err; cut -c1-400 gpurun_out/bench_${T}_default.json
