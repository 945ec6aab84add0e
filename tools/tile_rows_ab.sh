mkdir -p gpurun_out/r2w
for r in 0 128 0 128; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --steps 30 --warmup 5 --tile-rows $r > gpurun_out/r2w/b_$r.json 2> gpurun_out/r2w/b_$r.err || exit 1
  python - <<PY
import json
d=json.loads(open("gpurun_out/r2w/b_$r.json").read().strip().splitlines()[-1])
print("tile_rows $r", d["value"], d["ms_per_step"])
PY
done
