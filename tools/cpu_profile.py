"""Host-side cost of one train step (cProfile at a tiny batch so that the GPU never throttles the CPU)."""
import sys, os, time, cProfile, pstats, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
brain = bench.build_brain(dev, "bf16x3", B)
batch = bench.synthetic_batch(B, 0, dev)
for _ in range(10):
    brain.step += 1; brain.fit_batch(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50):
    brain.step += 1; brain.fit_batch(batch)
t1 = time.perf_counter()
torch.cuda.synchronize()
print(f"host issue time per step: {(t1-t0)/50*1e3:.2f} ms; with drain {(time.perf_counter()-t0)/50*1e3:.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(20):
    brain.step += 1; brain.fit_batch(batch)
pr.disable(); torch.cuda.synchronize()
st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(28)

# the backward runs on autograd's worker thread, invisible to the profile above: wrap it
from speech_anonymization_amd import convae as _cv
_pr2 = cProfile.Profile()
_orig = _cv._ConvAEFn.backward
def _wrapped(ctx, *g):
    _pr2.enable()
    try:
        return _orig(ctx, *g)
    finally:
        _pr2.disable()
_cv._ConvAEFn.backward = staticmethod(_wrapped)
for _ in range(20):
    brain.step += 1; brain.fit_batch(batch)
torch.cuda.synchronize()
print("==== inside _ConvAEFn.backward (20 calls) ====")
pstats.Stats(_pr2).sort_stats("tottime").print_stats(22)
