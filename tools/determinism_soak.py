"""Race detector for the hand-scheduled kernels: the SAME train-step forward + backward (bench workload, fixed
parameters, fixed batch, pooling noise fixed) N times; every gradient and both outputs must be bit-identical to
the first run every time (the counted waits, the dummy first-tile epilogue, the reserved in-flight registers and
the ticketless reductions leave no room for run-to-run differences unless something races).
  python tools/determinism_soak.py [N=600] [B=32] [samples per utterance]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from speech_anonymization_amd.brain import Stage

N = int(sys.argv[1]) if len(sys.argv) > 1 else 600
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = torch.device("cuda:0")
brain = bench.build_brain(dev, "bf16x3", B)
model = brain.modules["ConvAE"]
model.pooling_noise = torch.rand(B, 128)                     # the reference draws it per call: fixed here
batch = bench.synthetic_batch(B, 0, dev, int(sys.argv[3]) if len(sys.argv) > 3 else None)
nrm = brain.modules["normalize"]
state0 = nrm.state.clone()                                  # [count, glob_mean[80], glob_std[80]] on the device
bn0 = {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "num_batches" in k}


def one():
    nrm.state.copy_(state0)                                  # the normaliser and the BatchNorm buffers are stateful
    model.load_state_dict(bn0, strict=False)
    for p in model.parameters():
        p.grad = None
    out = brain.compute_forward(batch, Stage.TRAIN)
    loss = brain.compute_objectives(out, batch, Stage.TRAIN)
    loss.backward()
    return [out[0].detach().clone(), out[1].detach().clone(), loss.detach().clone()] + \
           [p.grad.detach().clone() for p in model.parameters()]


ref = one()
torch.cuda.synchronize()
names = ["recon", "logp", "loss"] + [k for k, _ in model.named_parameters()]
bad, t0 = 0, time.perf_counter()
for i in range(1, N + 1):
    got = one()
    diff = [n for n, a, b in zip(names, ref, got) if not torch.equal(a, b)]
    if diff:
        bad += 1
        if bad <= 5:
            print(f"run {i}: {len(diff)} tensors differ from run 0: {diff[:6]}", flush=True)
    if i % 100 == 0:
        print(f"{i} runs, {bad} with differences, {time.perf_counter() - t0:.0f} s", flush=True)
print("OK" if not bad else "MISMATCH")
sys.exit(0 if not bad else 1)
