#!/bin/bash
# A/B on ONE box: the FC head as one forward / one backward launch (default) against the separate launches
# (SA_FUSED_HEAD=0).  usage: bash tools/head_ab.sh <outdir>
OUT=${1:-gpurun_out/head_ab}; mkdir -p $OUT
for rep in 1 2; do
  for fh in 1 0; do
    SA_FUSED_HEAD=$fh timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline > $OUT/bench_fh${fh}_$rep.json 2> $OUT/bench_fh${fh}_$rep.err
    python3 - <<PY
import json
d = json.load(open("$OUT/bench_fh${fh}_$rep.json"))
print("fused_head=$fh rep $rep: B=32 %.3f ms (host %.2f)  B=10 %.3f ms graph=%s" % (d["ms_per_step"], d["host_issue_ms_per_step"], d["config"]["b10"]["ms_per_step"], d["config"]["b10"]["hip_graph"]))
PY
  done
done
