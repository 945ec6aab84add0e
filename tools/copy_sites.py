"""Where the device-to-device copies and fills of one train step come from: torch.profiler with
Python stacks over one eager step at B = 10 (rocprofv3 shows ~18 __amd_rocclr_copyBuffer and ~13 fill
launches per step)."""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
brain = bench.build_brain(dev, "bf16x3", 10)
batch = bench.synthetic_batch(10, 0, dev)
for _ in range(6):
    brain.step += 1
    brain.fit_batch(batch)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    brain.step += 1
    brain.fit_batch(batch)
    torch.cuda.synchronize()
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::zero_", "aten::clone", "aten::zeros", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "speech" in s or "bench.py" in s or "torch/optim" in s or "clip_grad" in s]
        sites[(ev.name, st[0] if st else (ev.stack[0] if ev.stack else "?"))] += 1
for (name, site), n in sites.most_common(40):
    print(f"{n:3d}  {name:14s} {site}")
