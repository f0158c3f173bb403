cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "conv_forward_backward or wgrad or bn_backward" > gpurun_out/r2_t13.log 2>&1; tail -3 gpurun_out/r2_t13.log
( export WB_TWO=0; python tools/wgrad_bench.py 40 64 192 64 64 9; python tools/wgrad_bench.py 40 32 96 64 64 9; python tools/wgrad_bench.py 40 16 48 128 128 9; python tools/wgrad_bench.py 40 8 24 128 128 9
python tools/wgrad_bench.py 40 128 384 32 32 9; python tools/wgrad_bench.py 40 64 192 32 32 9; python tools/wgrad_bench.py 40 256 768 16 16 9; python tools/wgrad_bench.py 40 128 384 16 16 9
python tools/wgrad_bench.py 40 128 384 64 16 1; python tools/wgrad_bench.py 40 32 96 256 64 1; python tools/wgrad_bench.py 40 64 192 128 32 1; python tools/wgrad_bench.py 40 256 768 32 16 1 ) > gpurun_out/r2_exp13.log 2>&1
grep "^wgrad" gpurun_out/r2_exp13.log
python -m pytest tests/test_networks_gpu.py -m gpu -q -x -p no:cacheprovider > gpurun_out/r2_t13b.log 2>&1; tail -3 gpurun_out/r2_t13b.log
python bench.py --steps 10 --no-configs3 --no-cpu-baseline > gpurun_out/r2_bench7.log 2>gpurun_out/r2_bench7.err; tail -c 300 gpurun_out/r2_bench7.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_bench7.log').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['launches_per_step'], d['kernel_ms_per_step_total'])
for f in d['families'][:10]: print(f['name'], round(f['ms_per_step'],2), f['GBs'] and round(f['GBs']), f['tflops'] and round(f['tflops']))
PY
