"""Per-launch HBM traffic of each kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE),
with the gfx950 corrections of MI355X_MICROARCH.md 'HBM': counters are in KiB; FETCH_SIZE reports
half the bytes of a wide coalesced streaming read (x2); WRITE_SIZE is exact for 16-B-per-lane
streaming stores.
  usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_...csv> [filter]
         [--json profiles/pmc_traffic.json --tag bf16x3:B32 --steps 10 --source "<text>"]
With --json the per-launch figures are also written under bench.py's kernel labels, plus "<tag>:step" =
the HBM bytes of ONE train step (every dispatch of the profiled process summed, divided by its steps:
bench.py reports it as `step_traffic` beside step_hbm_roofline_frac)."""
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


import json
import re

argv = sys.argv[1:]
opts = {}
for o in ("--json", "--tag", "--steps", "--source"):
    if o in argv:
        i = argv.index(o)
        opts[o] = argv[i + 1]
        del argv[i:i + 2]
fetch, write = load(argv[0], "FETCH_SIZE"), load(argv[1], "WRITE_SIZE")
flt = argv[2] if len(argv) > 2 else ""


def bench_label(k):
    """rocprofv3's kernel name -> the label bench.py gives the launch (ops.conv_gemm's `kind`)"""
    m = re.search(r"sa_conv_wsd_kernel<(\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        return "sa_conv_wsd_kernel<%s,%s,%s,%s> (bf16x3_t, 128->128)" % m.groups()
    m = re.search(r"sa_conv_ws_kernel<(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)>", k)
    if m:
        mode, nt, halo, cc, co, sa, uu = m.groups()
        return f"sa_conv_ws_kernel<{mode},{nt}> (bf16x3_t, {cc}->{co})"
    m = re.search(r"sa_conv_gemm_kernel<(\w+), (\d+), (\d+), (\d+), (\d+), (\d+), (true|false)>", k)
    if m:
        t, ci, co, sa, uu, tm, pro = m.groups()
        return f"sa_conv_gemm_kernel<{t},{ci},{co},{sa},{uu}{',nb prologue' if pro == 'true' else ''}>"
    return None
rows = []
for k in fetch:
    if flt in k:
        n, f = fetch[k]
        w = write.get(k, [n, 0.0])[1]
        rows.append((2 * f * 1024 / n + w * 1024 / n, n, 2 * f * 1024 / n, w * 1024 / n, k))
print("launches  read_MB(2xFETCH)  write_MB  total_MB  kernel")
for tot, n, f, w, k in sorted(rows, reverse=True)[:40]:
    print(f"{n:8d}  {f/1e6:14.2f}  {w/1e6:8.2f}  {tot/1e6:8.2f}  {k[:100]}")

if "--json" in opts:
    tag, nsteps = opts.get("--tag", "bf16x3:B32"), int(opts.get("--steps", "10"))
    src = opts.get("--source", "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of the bench command; FETCH_SIZE x2")
    try:
        out = json.load(open(opts["--json"]))
    except (OSError, ValueError):
        out = {}
    out["_comment"] = ("HBM bytes per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB units, "
                       "FETCH_SIZE x2 (gfx950 wide-read correction); written by tools/pmc_traffic.py --json from the `source` "
                       "of each entry.  Keys: <dtype>:B<batch>:<bench.py kernel label>; <dtype>:B<batch>:step = one train step.")
    tot_r = tot_w = 0.0
    wsd = [0, 0.0, 0.0]                       # the instances of the fused data-gradient kernel as one record
    for k in fetch:
        n, f = fetch[k]
        w = write.get(k, [n, 0.0])[1]
        tot_r += 2 * f * 1024
        tot_w += w * 1024
        lab = bench_label(k)
        if lab:
            out[f"{tag}:{lab}"] = {"read_bytes": 2 * f * 1024 / n, "write_bytes": w * 1024 / n,
                                   "total_bytes": (2 * f + w) * 1024 / n, "launches": n, "source": src}
        if "sa_conv_wsd_kernel<" in k:
            wsd = [wsd[0] + n, wsd[1] + 2 * f * 1024, wsd[2] + w * 1024]
    if wsd[0]:
        out[f"{tag}:sa_conv_wsd_kernel (bf16x3_t, 128->128; 5 instances)"] = {
            "read_bytes": wsd[1] / wsd[0], "write_bytes": wsd[2] / wsd[0], "total_bytes": (wsd[1] + wsd[2]) / wsd[0],
            "launches": wsd[0], "source": src}
    out[f"{tag}:step"] = {"read_bytes": tot_r / nsteps, "write_bytes": tot_w / nsteps, "total_bytes": (tot_r + tot_w) / nsteps,
                          "steps": nsteps, "source": src}
    json.dump(out, open(opts["--json"], "w"), indent=1)
    print(f"wrote {opts['--json']}: step traffic {(tot_r + tot_w) / nsteps / 1e9:.2f} GB")
