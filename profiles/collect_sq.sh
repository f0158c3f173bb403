#!/usr/bin/env bash
# Shader-engine counters of one eager bench run per counter group (run through gpurun from the repo root):
#   bash profiles/collect_sq.sh r02 <git head>      -> gpurun_out/<tag>_pmc_sq.json   (copy into profiles/ afterwards)
set -uo pipefail
TAG="${1:-r02}"; HEAD="${2:-unknown}"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
FILES=""
i=0
for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES"; do
  i=$((i+1))
  rm -rf gpurun_out/${TAG}_sq_$i
  rocprofv3 --pmc $c -d gpurun_out/${TAG}_sq_$i -o p --output-format csv -- python3 bench.py --steps 2 --warmup 1 --no-graph --no-cpu-baseline --no-configs3 --no-kernel-timing > gpurun_out/${TAG}_sq_$i.log 2>&1 || { tail -5 gpurun_out/${TAG}_sq_$i.log; exit 1; }
  FILES="$FILES $(find gpurun_out/${TAG}_sq_$i -name '*counter_collection.csv' | head -1)"
  echo "pmc group $i done"
done
python profiles/pmc_sq.py gpurun_out/${TAG}_pmc_sq.json 3 "$HEAD" $FILES
rm -rf gpurun_out/${TAG}_sq_1 gpurun_out/${TAG}_sq_2 gpurun_out/${TAG}_sq_3
