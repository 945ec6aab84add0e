mkdir -p gpurun_out/r2j
timeout -k 10 600 python -m pytest tests/test_train_step_gpu.py -m gpu -q -x --timeout 600 2>&1 | tee gpurun_out/r2j/tests.log | tail -25
timeout -k 10 500 python bench.py --steps 20 --warmup 5 2>&1 | tee gpurun_out/r2j/bench_graph.log | tail -c 2500
timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-graph --no-cpu-baseline 2>&1 | tee gpurun_out/r2j/bench_eager.log | tail -c 1500
