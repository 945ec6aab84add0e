"""Which train step pays the interpreter's full garbage collection, and how long it takes (gc.callbacks).
usage: python tools/hiccup_probe.py [freeze]  -- with `freeze`, settle_python_heap() after step 8 (what Brain.fit / bench.py do)."""
import sys, os, time, gc, torch
sys.path.insert(0, "/root/repo")
import bench
dev = torch.device("cuda:0")
brain = bench.build_brain(dev, "bf16x3", 32)
batch = bench.synthetic_batch(32, 0, dev)
ev = []
def cb(phase, info):
    ev.append((time.perf_counter(), phase, info.get("generation"), info.get("collected")))
gc.callbacks.append(cb)
ts = []
for i in range(200):
    t0 = time.perf_counter()
    brain.step += 1
    brain.fit_batch(batch)
    if i == 8 and len(sys.argv) > 1:
        from speech_anonymization_amd.brain import settle_python_heap
        settle_python_heap()
    ts.append((i, t0, time.perf_counter() - t0))
torch.cuda.synchronize()
for i, t0, dt in ts:
    if dt > 0.03 and i > 3:
        print("slow step", i, round(dt * 1e3, 1), "ms")
        for (t, ph, g, c) in ev:
            if t0 <= t <= t0 + dt:
                print("   gc", ph, "gen", g, "collected", c, "at +%.1f ms" % ((t - t0) * 1e3))
print("gc counts", gc.get_count(), "thresholds", gc.get_threshold())
print("gen2 events:", [(round((t - ts[0][1]) * 1e3), ph) for (t, ph, g, c) in ev if g == 2])
