#!/bin/bash
# B = 10 launched eagerly (no hipGraph): separate reduce / finalise launches against the fused ones
# (SA_FUSED_FINALIZE=1: 34 fewer launches per step).  Prints step time and the host's issue time per step.
OUT=${OUT:-gpurun_out/b10_eager}; mkdir -p $OUT
for f in 0 1 0 1; do
  SA_FUSED_FINALIZE=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --no-graph --batch 10 --steps 50 --warmup 5 > $OUT/e.json 2> $OUT/e.err || { tail -5 $OUT/e.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/e.json").read().strip().splitlines()[-1])
print("B=10 eager fused_finalize=$f: %.3f ms/step, host issue %.3f ms/step" % (d["ms_per_step"], d["host_issue_ms_per_step"]))
PY
done
