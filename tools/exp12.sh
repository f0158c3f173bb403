cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp WB_ITERS=5
mkdir -p gpurun_out/pmc_w1 gpurun_out/pmc_w2 gpurun_out/pmc_w3
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_w1 -o w1 --output-format csv -- python tools/wgrad_bench.py 40 32 96 64 64 9 > gpurun_out/pmc_w1.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_IFETCH SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d gpurun_out/pmc_w2 -o w2 --output-format csv -- python tools/wgrad_bench.py 40 32 96 64 64 9 > gpurun_out/pmc_w2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --kernel-trace -d gpurun_out/pmc_w3 -o w3 --output-format csv -- python tools/wgrad_bench.py 40 32 96 64 64 9 > gpurun_out/pmc_w3.log 2>&1
ls -R gpurun_out/pmc_w1 | head; tail -3 gpurun_out/pmc_w1.log
