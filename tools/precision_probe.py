"""Diagnostic (not a test): rel-MSE of the HIP ConvAE outputs / gradients against the fp32 CPU
oracle at a realistic shape, per precision mode.  python tools/precision_probe.py [B] [T]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import convae as O, losses as L
from tests.test_convae_gpu import run_hip, run_oracle, rel_mse, cosine, NULL_BIAS
from tests import smoke_step

B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1008
modes = sys.argv[3].split(",") if len(sys.argv) > 3 else ["f32", "bf16", "bf16x3"]
import speech_anonymization_amd as pkg
from speech_anonymization_amd.convae import ConvAutoencoder
dev = torch.device("cuda:0")
wav = smoke_step.make_wave(B, (T - 1) * 160, seed=1)
fb, nrm = pkg.Fbank().to(dev), pkg.InputNormalization("global", 4).to(dev)
feats = nrm(fb(wav.to(dev)), torch.ones(B), epoch=1, pad_multiple=36).cpu()
T = feats.shape[1]
rs = np.random.RandomState(0)
target = feats + 0.1 * torch.from_numpy(rs.standard_normal(feats.shape).astype("float32"))
gender = torch.arange(B) % 2
torch.manual_seed(8886)
params = O.ConvAutoencoder().state_dict()          # default torch init (what training starts from)
t0 = time.time()
o_recon, o_logp, o_loss, o_grads, _ = run_oracle(params, feats, target, gender, "l1")
print(f"oracle fp32 step: {time.time()-t0:.1f}s  loss {o_loss:.6f}")
for mode in modes:
    base, _, dg = mode.partition("+dgrad=")
    base, _, dec = base.partition("+dec=")
    m = ConvAutoencoder(precision=base, pooling_noise=None)
    if dg:
        from speech_anonymization_amd import ops as _ops
        m.dgrad_kcode = _ops.PRECISIONS[dg][1]
    if dec:
        from speech_anonymization_amd import ops as _ops
        m.dec_kcode = _ops.PRECISIONS[dec][1]
    m.load_state_dict(params); m.to(dev).train()
    recon, logp, loss, grads = run_hip(m, feats, target, gender, "l1")
    print(f"== {mode}: loss {loss:.6f}  recon {rel_mse(recon, o_recon):.3e}  logp {rel_mse(logp, o_logp):.3e}")
    worst = {}
    for k, g in grads.items():
        if k in NULL_BIAS: continue
        grp = k.split(".")[0]
        e = rel_mse(g, o_grads[k])
        if e > worst.get(grp, (0, ""))[0]: worst[grp] = (e, k, cosine(g, o_grads[k]))
    for grp, (e, k, c) in worst.items():
        print(f"   worst {grp:15s} {e:.3e} (cos {c:.4f}) at {k}")
