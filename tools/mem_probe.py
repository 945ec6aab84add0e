import sys, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
brain = bench.build_brain(dev, "bf16x3", 32)
batch = bench.synthetic_batch(32, 0, dev)
for _ in range(3):
    brain.step += 1
    brain.fit_batch(batch)
torch.cuda.synchronize()
print("max_memory_allocated GB", torch.cuda.max_memory_allocated() / 1e9, "reserved GB", torch.cuda.memory_reserved() / 1e9)
