#!/bin/bash
# Round 6, last tree: r06_z.sh (kernel stats, both counter passes, reference-mode kernel stats at cfg2, GPU suite, three bench lines, smoke) + reference-mode kernel stats at cfg3 / cfg5
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
TAG=r06zz bash scripts/runs/r06_z.sh
TAG=r06zz bash scripts/runs/r06_strict_stats.sh
for wl in cfg3 cfg5; do cp gpurun_out/ks_r06zz_strict_$wl.csv profiles/r06zz_strict_kernel_stats_$wl.csv; done
mkdir -p gpurun_out/profiles_r06zz; cp profiles/r06zz_* gpurun_out/profiles_r06zz/
