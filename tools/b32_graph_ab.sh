mkdir -p gpurun_out/r2w
for g in "" "--graph" "" "--graph"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 $g > gpurun_out/r2w/g32.json 2> gpurun_out/r2w/g32.err || { tail -5 gpurun_out/r2w/g32.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2w/g32.json").read().strip().splitlines()[-1])
print("B=32 graph='$g'", d["value"], d["ms_per_step"], d.get("host_issue_ms_per_step"), "b10", d["config"]["b10"]["value"], d["config"]["b10"]["ms_per_step"])
PY
done
