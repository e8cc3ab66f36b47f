cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_fivept.py tests/test_gpu_kernels.py -q -m gpu -k "five_point or bootstrap" -s 2>&1 | tail -5
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_now.json 2> gpurun_out/bench_now.err; tail -3 gpurun_out/bench_now.err; python - <<'PY'
import json
j=json.loads(open('gpurun_out/bench_now.json').read().strip().splitlines()[-1])
print({k:j[k] for k in ('metric','value','ms_per_step')}); print('strict', json.dumps(j['strict'])[:900]); print('host_incl', j['host_inclusive']['value'], 'conc', j['concurrent']['value'], 'roof', j['roofline']['frac'], j['roofline']['avg_us'], 'boot', j['roofline']['groups'].get('bootstrap'))
PY
python scripts/ab_config.py cfg2 "" "--bootstrap_points 5" 2>&1 | tail -2
