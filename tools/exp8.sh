cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "conv_forward_backward or wgrad" > gpurun_out/r2_t8.log 2>&1; tail -4 gpurun_out/r2_t8.log
( python tools/wgrad_bench.py 40 64 192 64 64 9; python tools/wgrad_bench.py 40 32 96 64 64 9; python tools/wgrad_bench.py 40 16 48 128 128 9; python tools/wgrad_bench.py 40 8 24 128 128 9
python tools/wgrad_bench.py 40 128 384 32 32 9; python tools/wgrad_bench.py 40 64 192 32 32 9; python tools/wgrad_bench.py 40 256 768 16 16 9; python tools/wgrad_bench.py 40 128 384 16 16 9
python tools/wgrad_bench.py 40 128 384 64 16 1; python tools/wgrad_bench.py 40 32 96 256 64 1; python tools/wgrad_bench.py 40 64 192 128 32 1; python tools/wgrad_bench.py 40 256 768 32 16 1 ) > gpurun_out/r2_exp8.log 2>&1
grep "^wgrad" gpurun_out/r2_exp8.log
python tools/prof_copies.py > gpurun_out/r2_copies.log 2>&1; tail -45 gpurun_out/r2_copies.log
