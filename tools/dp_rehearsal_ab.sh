mkdir -p ${OUT:-gpurun_out/r2u}
i=0
for c in plain torch lib; do
  i=$((i+1))
  if [ $c = plain ]; then E="SA_X=0"; elif [ $c = torch ]; then E="SA_FORCE_DP=1 SA_DIST_BACKEND=nccl"; else E="SA_FORCE_DP=1 SA_COMM=lib"; fi
  env $E MASTER_PORT=2961$i timeout -k 10 200 python bench.py --no-cpu-baseline --steps 30 --warmup 5 > ${OUT:-gpurun_out/r2u}/bench_$c.json 2> ${OUT:-gpurun_out/r2u}/bench_$c.err || exit 1
  python - <<PY
import json
d=json.loads(open("${OUT:-gpurun_out/r2u}/bench_$c.json").read().strip().splitlines()[-1])
print("$c", d["value"], d["ms_per_step"], d["backend"], d["config"]["b10"]["value"], d["config"]["b10"]["ms_per_step"])
PY
done
