cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "bn_backward_fused or streaming or conv_forward_backward or bn_finalize" > gpurun_out/r2_t4.log 2>&1; tail -4 gpurun_out/r2_t4.log
python -m pytest tests/test_networks_gpu.py tests/test_golden_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t4b.log 2>&1; tail -4 gpurun_out/r2_t4b.log
for cfg in "1 1 2"; do set -- $cfg; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 256 768 16 32 1 1 0 20; done > gpurun_out/r2_exp3.log 2>&1
grep "^conv" gpurun_out/r2_exp3.log
python bench.py --steps 10 > gpurun_out/r2_bench3.log 2>gpurun_out/r2_bench3.err; tail -c 300 gpurun_out/r2_bench3.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench3.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['launches_per_step'], d['kernel_ms_per_step_total'])
for f in d['families'][:10]: print(f['name'], round(f['ms_per_step'],2), f['GBs'] and round(f['GBs']), f['tflops'] and round(f['tflops']))
print(d['configs3']['value'], d['configs3']['ms_per_step'])
print(d.get('arch_calc_check'))
PY
