#!/bin/bash
# A/B of static per-XCD weights for the persistent kernels' tile ranges (hipGraph replays): uniform against $1
W=${1:-65,63,65,63,65,63,65,63}; OUT=${OUT:-gpurun_out/xcdw}; mkdir -p $OUT
for w in "" "$W" "" "$W" "" "$W"; do
  for b in 10 32; do
    SA_XCD_WEIGHTS=$w timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --batch $b --graph --steps 40 --warmup 5 > $OUT/ab.json 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$OUT/ab.json").read().strip().splitlines()[-1])
print("weights='$w' B=$b %.3f ms/step" % d["ms_per_step"])
PY
  done
done
