#!/bin/bash
# Round 6, first call: the ADVICE r5 fixes on the GPU (riders test with counters, poisoned-flows texture window), baseline window hashes and bench line
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r06a
timeout 1500 python -m pytest tests/test_gpu_riders.py tests/test_gpu_vs_ref_window.py tests/test_gpu_voldor.py -m gpu -q -x > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR|Error" gpurun_out/${T}_pytest.log | tail -12
timeout 300 python scripts/window_hash.py cfg2 cfg3 cfg5 > gpurun_out/${T}_hash.txt 2>&1; grep -E "^cfg" gpurun_out/${T}_hash.txt
timeout 600 python bench.py > gpurun_out/${T}_bench.json 2> gpurun_out/${T}_bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/${T}_bench.json
