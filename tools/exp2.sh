cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "streaming or conv_forward_backward" > gpurun_out/r2_t3.log 2>&1; tail -4 gpurun_out/r2_t3.log
for cfg in "0 0 0" "1 1 0" "1 1 2"; do set -- $cfg; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 256 768 16 32 1 1 0 20; done > gpurun_out/r2_exp2.log 2>&1
for cfg in "0 0 0" "1 1 0"; do set -- $cfg; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 128 384 64 16 1 1 0 20; done >> gpurun_out/r2_exp2.log 2>&1
CB_STATS=0 python tools/conv_bench.py 40 256 768 32 16 1 0 1 20 >> gpurun_out/r2_exp2.log 2>&1
CB_STATS=0 CB_RES=1 python tools/conv_bench.py 40 128 384 16 64 1 0 1 20 >> gpurun_out/r2_exp2.log 2>&1
grep "^conv" gpurun_out/r2_exp2.log
python bench.py --steps 10 > gpurun_out/r2_bench2.log 2>gpurun_out/r2_bench2.err; tail -c 300 gpurun_out/r2_bench2.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench2.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['launches_per_step'], d['kernel_ms_per_step_total'])
for f in d['families'][:8]: print(f['name'], round(f['ms_per_step'],2), f['GBs'] and round(f['GBs']), f['tflops'] and round(f['tflops']))
print(d['configs3']['value'], d['configs3']['ms_per_step'])
PY
