#!/bin/bash
# packed fp32 in the per-frame geometry + bilinear blend: A = scalar build, B = packed build, same box
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp; mkdir -p gpurun_out
A=$PWD/voldor_amd/lib/libvoldor_hip.so; B=$PWD/voldor_amd/lib/libvoldor_hip_pk.so
for L in $A $B; do echo "== $(basename $L)"; VOLDOR_HIP_LIB=$L timeout 600 python scripts/window_hash.py cfg2 cfg3 cfg5 2>&1 | tail -3; done
for r in 1 2; do for wl in cfg2 cfg3 cfg5; do for L in $A $B; do echo -n "$(basename $L) "; VOLDOR_HIP_LIB=$L timeout 600 python scripts/ab_config.py $wl "" 2>&1 | tail -1; done; done; done
for L in $A $B; do n=$(basename $L .so)
  VOLDOR_HIP_LIB=$L rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/abpk_$n -- python bench.py --workload cfg5 --steps 5 --warmup 2 --no-extras > /dev/null 2>&1
  f=$(ls -t $(find gpurun_out/abpk_$n -name "*kernel_stats.csv") | head -1); cp $f gpurun_out/abpk_${n}_cfg5.csv; rm -rf gpurun_out/abpk_$n
  echo "== $n cfg5"; head -12 gpurun_out/abpk_${n}_cfg5.csv | python -c "
import csv,sys
for r in csv.DictReader(sys.stdin): print('  %-60s %6s x %9.2f us'%(r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))"
done
