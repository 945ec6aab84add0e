"""Per-launch HBM traffic of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
with the gfx950 corrections of MI355X_MICROARCH.md 'HBM': counters are in KiB; FETCH_SIZE reports
half the bytes of a wide coalesced streaming read (x2); WRITE_SIZE is exact for 16-B-per-lane
streaming stores.  usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_...csv> [filter]"""
import csv
import sys
from collections import defaultdict


def load(path, name):
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == name:
            a = acc[r["Kernel_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    return acc


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
flt = sys.argv[3] if len(sys.argv) > 3 else ""
rows = []
for k in fetch:
    if flt in k:
        n, f = fetch[k]
        w = write.get(k, [n, 0.0])[1]
        rows.append((2 * f * 1024 / n + w * 1024 / n, n, 2 * f * 1024 / n, w * 1024 / n, k))
print("launches  read_MB(2xFETCH)  write_MB  total_MB  kernel")
for tot, n, f, w, k in sorted(rows, reverse=True)[:40]:
    print(f"{n:8d}  {f/1e6:14.2f}  {w/1e6:8.2f}  {tot/1e6:8.2f}  {k[:100]}")
