cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "conv_forward_backward or wgrad" > gpurun_out/r2_t9.log 2>&1; tail -3 gpurun_out/r2_t9.log
( for n in 10 40 160; do python tools/wgrad_bench.py $n 32 96 64 64 9; done; for n in 10 40 160; do python tools/wgrad_bench.py $n 64 192 64 64 9; done; for n in 10 40 160; do python tools/wgrad_bench.py $n 16 48 128 128 9; done ) > gpurun_out/r2_exp9.log 2>&1
grep "^wgrad" gpurun_out/r2_exp9.log
