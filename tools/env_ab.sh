#!/bin/bash
# A/B of one environment knob (0 / 1) as hipGraph replays (no host effects), B = 10 and B = 32, two rounds.
#   usage: OUT=gpurun_out/x bash tools/env_ab.sh SA_OVERLAP_WGRAD
VAR=$1; OUT=${OUT:-gpurun_out/env_ab}; mkdir -p $OUT
for f in 0 1 0 1; do
  for b in 10 32; do
    env $VAR=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --batch $b --graph --steps 40 --warmup 5 > $OUT/ab.json 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$OUT/ab.json").read().strip().splitlines()[-1])
print("$VAR=$f B=$b %.3f ms/step" % d["ms_per_step"])
PY
  done
done
