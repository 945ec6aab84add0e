"""Per-step kernel time table from a rocprofv3 --kernel-trace --stats CSV (bench.py --steps 5 --warmup 2
= 7 steps).  usage: python tools/prof_summary.py <kernel_stats.csv> [other_kernel_stats.csv] [nsteps]"""
import csv
import sys


def load(p, n):
    return {r["Name"]: (int(r["Calls"]) / n, float(r["TotalDurationNs"]) / n / 1e3) for r in csv.DictReader(open(p))}


paths = [a for a in sys.argv[1:] if a.endswith(".csv")]
n = int(next((a for a in sys.argv[1:] if a.isdigit()), 7))
tabs = [load(p, n) for p in paths]
print("total us/step:", [round(sum(v[1] for v in t.values()), 1) for t in tabs])
keys = sorted(set().union(*tabs), key=lambda k: -max(t.get(k, (0, 0))[1] for t in tabs))
for k in keys[:int(40)]:
    print(f"{k[:70]:70s} " + " | ".join(f"{t.get(k, (0, 0))[0]:5.1f} {t.get(k, (0, 0))[1]:8.1f}" for t in tabs))
