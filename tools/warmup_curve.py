"""Step time in windows of 10 steps from process start (how long the step takes to reach its steady
state: clocks, allocator pools, the normaliser's statistics)."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
brain = bench.build_brain(dev, "bf16x3", B)
batch = bench.synthetic_batch(B, 0, dev)
out = []
for w in range(30):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        brain.step += 1
        brain.fit_batch(batch)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    out.append(((time.perf_counter() - t0) / 10 * 1e3, (t1 - t0) / 10 * 1e3))
for w, (ms, host) in enumerate(out):
    print(f"steps {w*10:3d}-{w*10+9:3d}: {ms:6.2f} ms/step (host issue {host:5.2f})")
