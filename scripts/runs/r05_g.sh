#!/bin/bash
# Round 5: the suite and the bench line on the tree with the frame-by-frame upload of host flows
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 1700 python -m pytest tests -m gpu -q -x > gpurun_out/r05g_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|^FAILED|^ERROR" gpurun_out/r05g_pytest.log | tail -12
timeout 900 python bench.py > gpurun_out/r05g_bench.json 2> gpurun_out/r05g_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads(open("gpurun_out/r05g_bench.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "value_host_inclusive", "ms_per_step")}); print(d["host_inclusive"]); print(d["concurrent"]); print(d["roofline"]["frac"], d["roofline"]["avg_us"]); print(d["strict"]["ms_per_window"]); print(d["workloads"])
PY
