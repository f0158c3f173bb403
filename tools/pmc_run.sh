#!/usr/bin/env bash
# rocprofv3 SQ counter groups over one command (development aid): bash tools/pmc_run.sh <tag> <kernel filter> python3 <script> ...
set -uo pipefail
TAG="$1"; FLT="$2"; shift 2
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
FILES=""
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf gpurun_out/${TAG}_sq_$i
  rocprofv3 --pmc $c -d gpurun_out/${TAG}_sq_$i -o p --output-format csv -- "$@" > gpurun_out/${TAG}_sq_$i.log 2>&1 || { tail -5 gpurun_out/${TAG}_sq_$i.log; exit 1; }
  FILES="$FILES $(find gpurun_out/${TAG}_sq_$i -name '*counter_collection.csv' | head -1)"
done
python tools/pmc_kernels.py "$FLT" $FILES
rm -rf gpurun_out/${TAG}_sq_1 gpurun_out/${TAG}_sq_2 gpurun_out/${TAG}_sq_3
