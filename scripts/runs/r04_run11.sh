cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_vs_ref_kernels.py -q -m gpu -s -k "optimize_depth_vs or solve_batch_p3p_vs" 2>&1 | grep "MEASURED\|passed\|failed"
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('metric','value','ms_per_step')}, 'strict', j['strict']['ms_per_window'], 'host_incl', j['host_inclusive']['value'], 'conc', j['concurrent']['value'], 'roof', j['roofline']['frac'], j['roofline']['avg_us'], j['roofline']['source'])
PY
