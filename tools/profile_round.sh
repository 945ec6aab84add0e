#!/bin/bash
# rocprofv3 passes of one round on the GPU box: kernel stats, HBM traffic (two --pmc passes), SQ counters.
# usage: bash tools/profile_round.sh <tag>      (summaries land in gpurun_out/<tag>/)
set -o pipefail
TAG=${1:-r02}
OUT=$PWD/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
B="--no-cpu-baseline --no-b10 --no-graph"   # eager only: under the profiler the host is slow enough to trigger the replay re-timing
run() { name=$1; shift; echo "== $name: $*"; ( cd $PWD && timeout -k 10 420 rocprofv3 "$@" ) > $OUT/$name.log 2>&1; echo "   rc=$?"; }
run ks  --kernel-trace --stats --output-format csv -d /tmp/prof_ks -o ks -- python3 bench.py --steps 5 --warmup 2 $B
run fetch --pmc FETCH_SIZE --output-format csv -d /tmp/prof_fetch -o fetch -- python3 bench.py --steps 2 --warmup 0 $B
run write --pmc WRITE_SIZE --output-format csv -d /tmp/prof_write -o write -- python3 bench.py --steps 2 --warmup 0 $B
run sq --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA --output-format csv -d /tmp/prof_sq -o sq -- python3 bench.py --steps 2 --warmup 0 $B
find /tmp/prof_ks /tmp/prof_fetch /tmp/prof_write /tmp/prof_sq -name "*.csv" | sed 's/^/   /'
KS=$(find /tmp/prof_ks -name "*kernel_stats.csv" | head -1)
[ -n "$KS" ] && cp $KS $OUT/kernel_stats.csv && python3 tools/prof_summary.py $OUT/kernel_stats.csv 15 > $OUT/kernel_stats_per_step.txt
F=$(find /tmp/prof_fetch -name "*counter_collection.csv" | head -1); W=$(find /tmp/prof_write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_traffic.py $F $W --json $OUT/pmc_traffic.json --tag bf16x3:B32 --steps 10 \
   --source "profiles/${TAG}_pmc_traffic_bf16x3_B32.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes of this bench command, 8 set-up + 2 steps; FETCH_SIZE x2)" > $OUT/pmc_traffic.txt
S=$(find /tmp/prof_sq -name "*counter_collection.csv" | head -1)
[ -n "$S" ] && python3 tools/sq_summary.py $S sa_ > $OUT/sq_counters.txt
for f in $OUT/*.log; do tail -n 2 $f | cut -c1-200; done
ls -la $OUT
