mkdir -p gpurun_out/r2m
timeout -k 10 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -k "fp8" -s --timeout 600 2>&1 | grep -v amdgpu | tee gpurun_out/r2m/tests.log | tail -30
timeout -k 10 300 python bench.py --dtype fp8 --samples 480000 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-b10 2>&1 | tee gpurun_out/r2m/bench_fp8.log | tail -c 1800
timeout -k 10 300 python bench.py --dtype bf16 --samples 480000 --batch 8 --steps 10 --warmup 3 --no-cpu-baseline --no-b10 2>&1 | tee gpurun_out/r2m/bench_bf16.log | tail -c 600
