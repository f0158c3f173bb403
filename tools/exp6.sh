cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t6.log 2>&1; tail -4 gpurun_out/r2_t6.log
python bench.py --steps 10 --no-configs3 --no-cpu-baseline > gpurun_out/r2_bench4.log 2>gpurun_out/r2_bench4.err; tail -c 300 gpurun_out/r2_bench4.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench4.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['launches_per_step'], d['kernel_ms_per_step_total'])
for f in d['families'][:14]: print(f['name'], round(f['ms_per_step'],2), f['GBs'] and round(f['GBs']), f['tflops'] and round(f['tflops']))
PY
