mkdir -p gpurun_out/r2w
for g in "" "--graph" "" "--graph"; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --batch 10 --steps 50 --warmup 5 $g > gpurun_out/r2w/g.json 2> gpurun_out/r2w/g.err || { tail -5 gpurun_out/r2w/g.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2w/g.json").read().strip().splitlines()[-1])
print("B=10 graph='$g'", d["value"], d["ms_per_step"], d.get("host_issue_ms_per_step"))
PY
done
