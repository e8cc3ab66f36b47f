#!/bin/bash
# Round 6: where did "four windows in flight" go (651 -> 430)?  The same leg under three libraries.
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
for L in libR5 libA libvoldor_hip; do
  for rep in 1 2; do
    VOLDOR_HIP_LIB=$PWD/voldor_amd/lib/$L.so python bench.py --no-cpu-baseline --no-workloads --pmc replay 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'value', d['value'], 'conc', d['concurrent']['value'])"
  done
done
