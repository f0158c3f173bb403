"""Summarise a rocprofv3 (rocpd sqlite) kernel trace: per-kernel stats CSV + the busy/idle split of the last replayed step.

usage: python profiles/summarize_rocpd.py <results.db> <out_prefix> [launches_per_step]
"""
import csv, re, sqlite3, sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:110]


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = c.execute("select name, start, end from kernels order by start").fetchall()
    agg = defaultdict(lambda: [0, 0, 1 << 62, 0])
    for n, s, e in rows:
        a = agg[short(n)]
        d = e - s
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    with open(out + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            w.writerow([n, a[0], a[1], round(a[1] / a[0], 1), round(100.0 * a[1] / tot, 3), a[2], a[3]])
    # the last step = the dispatches after the last large idle gap pattern: take the final `n` launches if given
    if len(sys.argv) > 3:
        n = int(sys.argv[3])
        last = rows[-n:]
        busy = sum(e - s for _, s, e in last)
        span = last[-1][2] - last[0][1]
        print(f"last {n} launches: span {span/1e6:.3f} ms, busy {busy/1e6:.3f} ms, idle {(span-busy)/1e6:.3f} ms")
    print(f"{len(rows)} dispatches, {tot/1e6:.1f} ms total kernel time -> {out}_kernel_stats.csv")


if __name__ == "__main__":
    main()
