#!/bin/bash
# Round 6: the whole GPU suite with the five-point bootstrap as the fast mode's default
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
T=${TAG:-r06h}
timeout 2400 python -m pytest tests -m gpu -q > gpurun_out/${T}_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/${T}_pytest.log | tail -20
