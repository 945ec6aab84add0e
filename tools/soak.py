"""Soak: N steps of the bench workload; reports step time in windows and allocator statistics."""
import sys, os, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
brain = bench.build_brain(dev, "bf16x3", 32)
batch = bench.synthetic_batch(32, 0, dev)
for w in range(n // 50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50):
        brain.step += 1
        loss = brain.fit_batch(batch)
    torch.cuda.synchronize()
    print(f"steps {w*50:4d}-{w*50+49:4d}: {(time.perf_counter()-t0)/50*1e3:6.2f} ms/step  loss {float(loss):.4f}  "
          f"alloc {torch.cuda.memory_allocated()/1e9:.2f} GB  peak {torch.cuda.max_memory_allocated()/1e9:.2f} GB  "
          f"reserved {torch.cuda.memory_reserved()/1e9:.2f} GB", flush=True)
