cd $GRAFT_REPO_ROOT
( for two in 0 1; do export WB_TWO=$two; python tools/wgrad_bench.py 40 64 192 64 64 9; python tools/wgrad_bench.py 40 32 96 64 64 9; python tools/wgrad_bench.py 40 16 48 128 128 9; python tools/wgrad_bench.py 40 8 24 128 128 9
python tools/wgrad_bench.py 40 128 384 32 32 9; python tools/wgrad_bench.py 40 64 192 32 32 9; python tools/wgrad_bench.py 40 256 768 16 16 9; python tools/wgrad_bench.py 40 128 384 16 16 9
python tools/wgrad_bench.py 40 128 384 64 16 1; python tools/wgrad_bench.py 40 32 96 256 64 1; python tools/wgrad_bench.py 40 64 192 128 32 1; python tools/wgrad_bench.py 40 256 768 32 16 1; done ) > gpurun_out/r2_exp11.log 2>&1
grep "^wgrad" gpurun_out/r2_exp11.log
for fl in 0 2; do
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 64 192 64 64 9 1 0
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 32 96 64 64 9 1 0
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 16 48 128 128 9 1 0
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 8 24 128 128 9 1 0
done > gpurun_out/r2_exp11b.log 2>&1
grep "^conv" gpurun_out/r2_exp11b.log
