"""Per-kernel SQ counter summary from a rocprofv3 --pmc counter_collection CSV: for each kernel the
per-launch mean of every counter collected, plus the derived shares the guide defines
(MI355X_MICROARCH.md 'rocprofv3 PMC slots': WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES,
quad-cycle units; SQ_VALU_MFMA_BUSY_CYCLES in cycles).
usage: python tools/sq_summary.py <counter_collection.csv> [kernel name filter]"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    a = acc[r["Kernel_Name"]][r["Counter_Name"]]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for k in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 0.0])[1]):
    if flt not in k:
        continue
    c = {n: v[1] / max(1, v[0]) for n, v in acc[k].items()}
    n = max(v[0] for v in acc[k].values())
    print(f"{k[:110]}\n   launches {n}")
    for name in sorted(c):
        print(f"   {name:28s} {c[name]:16.0f}")
    wc = c.get("SQ_WAVE_CYCLES")
    if wc:
        for name in ("SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_ACTIVE_INST_ANY"):
            if name in c:
                print(f"   {name + ' / SQ_WAVE_CYCLES':40s} {c[name] / wc:6.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"]:
        # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs of the chip (= 32 x MFMA count for 32x32x16 bf16),
        # SQ_BUSY_CYCLES sums the 32 shader engines: matrix-pipe busy share of the launch
        share = (c["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0) / (c["SQ_BUSY_CYCLES"] / 32.0)
        print(f"   {'MFMA pipe busy share (per SIMD / per SE)':40s} {share:6.3f}")
