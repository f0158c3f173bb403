#!/usr/bin/env bash
# Collect the round's evidence on the GPU box (run through gpurun from the repo root):
#   bash profiles/collect.sh r02 <git head>
# 1. bench line (default flags)            -> gpurun_out/<tag>_bench.json
# 2. rocprofv3 --kernel-trace --stats      -> gpurun_out/<tag>_kernel_stats.csv   (same bench command, graph replay)
# 3. rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in two separate passes (eager steps) -> gpurun_out/<tag>_pmc_traffic.json
# Copy the three files into profiles/ afterwards (gpurun_out/ is scratch).
set -uo pipefail
TAG="${1:-r02}"; HEAD="${2:-unknown}"
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py --steps 10 > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err
rc=$?
echo "$rc" > gpurun_out/${TAG}_bench.rc          # the exit status is evidence too (normal interpreter teardown, no masking)
if [ "$rc" -ne 0 ]; then tail -c 3000 gpurun_out/${TAG}_bench.err; exit 1; fi
echo "bench done"
rm -rf gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats -d gpurun_out/${TAG}_trace -o t --output-format csv -- python3 bench.py --steps 16 --no-cpu-baseline --no-configs3 --no-kernel-timing > gpurun_out/${TAG}_trace.log 2>&1
python - "$TAG" <<'PY'
import csv, glob, sys
tag = sys.argv[1]
f = glob.glob(f"gpurun_out/{tag}_trace/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open(f"gpurun_out/{tag}_kernel_stats.csv", "w", newline="") as out:
    w = csv.writer(out)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
print("kernel stats:", len(rows), "kernels")
PY
echo "trace done"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/${TAG}_pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d gpurun_out/${TAG}_pmc_$c -o p --output-format csv -- python3 bench.py --steps 5 --warmup 1 --no-graph --no-cpu-baseline --no-configs3 --no-kernel-timing > gpurun_out/${TAG}_pmc_$c.log 2>&1
  echo "pmc $c done"
done
F=$(find gpurun_out/${TAG}_pmc_FETCH_SIZE -name '*counter_collection.csv' | head -1)
W=$(find gpurun_out/${TAG}_pmc_WRITE_SIZE -name '*counter_collection.csv' | head -1)
python profiles/pmc_traffic.py "$F" "$W" 6 gpurun_out/${TAG}_pmc_traffic.json "$HEAD"
rm -rf gpurun_out/${TAG}_trace gpurun_out/${TAG}_pmc_FETCH_SIZE gpurun_out/${TAG}_pmc_WRITE_SIZE
