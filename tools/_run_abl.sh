mkdir -p gpurun_out/r2i
for st in 0 10000 20000 30000 45000; do SA_CONV_IMPL=old SA_STAGGER=$st timeout -k 10 300 python tools/conv_ablate.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r2i/stagger.log; done
