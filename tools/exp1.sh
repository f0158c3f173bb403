cd $GRAFT_REPO_ROOT
python bench.py --steps 10 > gpurun_out/r2_bench1.log 2>gpurun_out/r2_bench1.err; tail -c 300 gpurun_out/r2_bench1.err
for cfg in "0 0 0" "1 0 0" "0 1 0" "1 1 0" "1 1 2"; do set -- $cfg; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 256 768 16 32 1 1 0 20; done >> gpurun_out/r2_exp1.log 2>&1
for cfg in "0 0 0" "1 0 0" "0 1 0" "1 1 0"; do set -- $cfg; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 128 384 64 16 1 1 0 20; done >> gpurun_out/r2_exp1.log 2>&1
for cfg in "0 0 0" "1 1 0"; do set -- $cfg; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 32 96 128 256 1 1 0 20; CB_AFF=$1 CB_STATS=$2 CB_RES=$3 python tools/conv_bench.py 40 32 96 256 64 1 1 0 20; done >> gpurun_out/r2_exp1.log 2>&1
cat gpurun_out/r2_exp1.log
