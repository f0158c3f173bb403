"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel name (development aid).
usage: python tools/pmc_kernels.py <substring filter> <csv> [<csv> ...]"""
import csv, sys, re
from collections import defaultdict
flt, files = sys.argv[1], sys.argv[2:]
tot = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(set))
for path in files:
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if flt not in k:
            continue
        k = re.sub(r"\(.*", "", k)[:90]
        tot[k][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tot):
    c = tot[k]
    n = max(len(v) for v in cnt[k].values())
    wc = c.get("SQ_WAVE_CYCLES", 0) or 1
    waves = c.get("SQ_WAVES", 0) or 1
    line = f"{k}\n    launches {n}  waves/launch {waves / n:.0f}  wave_cycles/wave {4 * wc / waves:.0f}"
    for x in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
        if x in c:
            line += f"  {x[3:]} {c[x] / wc:.2f}"
    for x in ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_INSTS_MFMA", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"):
        if x in c:
            line += f"  {x[9:]}/wave {c[x] / waves:.0f}"
    if c.get("SQ_LDS_IDX_ACTIVE"):
        line += f"  lds_conflict {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.2f} lds_active/busy {c['SQ_LDS_IDX_ACTIVE'] / max(c.get('SQ_BUSY_CYCLES', 1), 1):.3f}"
    print(line)
