cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --shape-tags --no-configs3 --no-cpu-baseline > gpurun_out/r2_tags.log 2>gpurun_out/r2_tags.err; tail -c 300 gpurun_out/r2_tags.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r2_tags.log').read().strip().splitlines()[-1])
json.dump(d, open('gpurun_out/r2_tags.json','w'))
print(d['ms_per_step'])
for f in d['families'][:60]: print(f"{f['name']:64s} n={f['launches_per_step']:5.1f} {f['ms_per_step']*1000:8.1f}us  {f['GBs'] or 0:6.0f}GB/s {f['tflops'] or 0:6.1f}TF  lb={f['launcher_gbytes_per_step'] or 0:.2f}GB")
PY
