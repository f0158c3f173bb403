"""Shader-engine counters per kernel family from rocprofv3 --pmc passes of the bench command (eager steps).

usage: python profiles/pmc_sq.py <out.json> <steps> <git head> <counter_collection.csv> [<counter_collection.csv> ...]

Per family: launches per step and the per-launch mean of every collected counter, plus the derived fractions the guide
(MI355X_MICROARCH.md, PMC section) reads them as: wave cycles parked in s_waitcnt / barriers (SQ_WAIT_ANY), issue stalls
(SQ_WAIT_INST_ANY), instruction issue (SQ_ACTIVE_INST_ANY), LDS bank-conflict share of the LDS-array cycles.
"""
import csv, json, sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import family


def main():
    out, steps, head, files = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4:]
    tot = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(lambda: defaultdict(set))
    for path in files:
        for r in csv.DictReader(open(path)):
            f = family(r["Kernel_Name"])
            tot[f][r["Counter_Name"]] += float(r["Counter_Value"])
            disp[f][r["Counter_Name"]].add(r["Dispatch_Id"])
    fams = {}
    for f, cs in tot.items():
        n = max(len(v) for v in disp[f].values())
        d = {"launches_per_step": n / steps}
        for c, v in cs.items():
            d[c + "_per_launch"] = v / max(len(disp[f][c]), 1)
        wc = cs.get("SQ_WAVE_CYCLES", 0.0)
        if wc:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in cs:
                    d[c + "_frac_of_wave_cycles"] = cs[c] / wc
        if cs.get("SQ_LDS_IDX_ACTIVE"):
            d["lds_bank_conflict_frac"] = cs.get("SQ_LDS_BANK_CONFLICT", 0.0) / cs["SQ_LDS_IDX_ACTIVE"]
        fams[f] = d
    order = sorted(fams, key=lambda k: -tot[k].get("SQ_WAVE_CYCLES", 0.0))
    json.dump({"note": "rocprofv3 --pmc passes (separate runs per counter group) of `bench.py --no-graph`; SQ_* cycle counters tick in "
                       "quad-cycles (MI355X_MICROARCH.md); families as in pmc_traffic.py",
               "git_head": head, "steps": steps, "families": {k: fams[k] for k in order}}, open(out, "w"), indent=1)
    print(f"{len(fams)} families -> {out}")


if __name__ == "__main__":
    main()
