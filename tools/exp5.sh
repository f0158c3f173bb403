cd $GRAFT_REPO_ROOT
python -m pytest tests/test_hip_ops.py -m gpu -q -x -p no:cacheprovider -k "specialised or conv_forward_backward or bn_backward_fused or halo_and_gather" > gpurun_out/r2_t5.log 2>&1; tail -4 gpurun_out/r2_t5.log
for fl in 0 2; do
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 64 192 64 64 9 1 0 20
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 64 192 64 64 9 0 1 20
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 32 96 64 64 9 1 0 20
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 16 48 128 128 9 1 0 20
CB_FLAGS=$fl CB_STATS=0 python tools/conv_bench.py 40 8 24 128 128 9 1 0 20
CB_FLAGS=$fl CB_STATS=1 CB_AFF=1 python tools/conv_bench.py 40 32 96 64 64 9 1 0 20
done > gpurun_out/r2_exp5.log 2>&1
grep "^conv" gpurun_out/r2_exp5.log
