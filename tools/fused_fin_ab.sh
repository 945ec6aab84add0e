# A/B of the fused reduce-and-finalise launches (SA_FUSED_FINALIZE=1) as hipGraph replays (no host effects)
mkdir -p ${OUT:-gpurun_out/r2x}
for f in 0 1 0 1; do
  for b in 10 32; do
    SA_FUSED_FINALIZE=$f timeout -k 10 200 python bench.py --no-cpu-baseline --no-b10 --batch $b --graph --steps 40 --warmup 5 > ${OUT:-gpurun_out/r2x}/f.json 2> ${OUT:-gpurun_out/r2x}/f.err || { tail -5 ${OUT:-gpurun_out/r2x}/f.err; exit 1; }
    python - <<PY
import json
d=json.loads(open("${OUT:-gpurun_out/r2x}/f.json").read().strip().splitlines()[-1])
print("fused=$f B=$b", round(d["value"]), round(d["ms_per_step"],3))
PY
  done
done
