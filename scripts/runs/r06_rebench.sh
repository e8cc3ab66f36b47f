cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
T=r06zz
R="--strict_math 1 --reference_draw 1 --reference_svd 1"
bash scripts/kstats_cfg.sh ${T}_strict_cfg2 cfg2 "$R" > gpurun_out/ks_${T}_strict_cfg2.txt 2>&1; head -8 gpurun_out/ks_${T}_strict_cfg2.txt; rm -rf gpurun_out/ks_${T}_strict_cfg2
for wl in cfg2 cfg3 cfg5; do
  timeout 900 python bench.py --workload $wl > gpurun_out/bench_${T}_$wl.json 2> gpurun_out/bench_${T}_$wl.err
  echo "bench $wl rc=$?"; cut -c1-200 gpurun_out/bench_${T}_$wl.json
done
python __graft_entry__.py smoke 2>&1 | tail -1
