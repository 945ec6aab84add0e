#!/bin/bash
# A/B of two builds of the library as hipGraph replays: SA_HIP_LIB=<other .so> against the in-tree one.
#   usage: OUT=gpurun_out/x bash tools/lib_ab.sh build/abl/libsa_hip_prev.so
OTHER=$1; OUT=${OUT:-gpurun_out/lib_ab}; mkdir -p $OUT
for which in tree other tree other; do
  for b in 10 32; do
    if [ $which = other ]; then export SA_HIP_LIB=$PWD/$OTHER; else unset SA_HIP_LIB; fi
    timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --batch $b --graph --steps 40 --warmup 5 > $OUT/ab.json 2> $OUT/ab.err || { tail -5 $OUT/ab.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("$OUT/ab.json").read().strip().splitlines()[-1])
print("$which B=$b %.3f ms/step" % d["ms_per_step"])
PY
  done
done
