#!/bin/bash
# kernel-time table of the B = 10 step (eager) -> gpurun_out/<tag>/b10_kernel_stats_per_step.txt
TAG=${1:-b10}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( cd $PWD && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b10 -o ks -- python3 bench.py --batch 10 --steps 5 --warmup 2 --no-cpu-baseline --no-b10 --no-graph ) > $OUT/b10_ks.log 2>&1
echo "rc=$?"
KS=$(find /tmp/prof_b10 -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && cp $KS $OUT/b10_kernel_stats.csv && python3 tools/prof_summary.py $OUT/b10_kernel_stats.csv 15 > $OUT/b10_kernel_stats_per_step.txt
head -60 $OUT/b10_kernel_stats_per_step.txt
