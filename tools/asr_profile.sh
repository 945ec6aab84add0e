#!/bin/bash
# kernel-time table of the frozen-ASR utility branch (tools/asr_utility_bench.py) -> gpurun_out/<tag>/
TAG=${1:-asr}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd $PWD && timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_asr -o asr -- python3 tools/asr_utility_bench.py --steps 5 ) > $OUT/asr_prof.log 2>&1
echo "rc=$?"
KS=$(find /tmp/prof_asr -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && cp $KS $OUT/asr_kernel_stats.csv && python3 - <<PY
import csv
rows = list(csv.DictReader(open("$KS")))
for r in rows[:30]:
    print(r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["Percentage"])
PY
