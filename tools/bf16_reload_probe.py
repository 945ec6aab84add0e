"""Parity probe for VERDICT r2 item 2 ("cut the bytes of the backward re-reads"): what happens to the 56
parameter gradients if the tensors the backward only RE-READS -- the stored forward tensor y of the
normalisation-backward prologues (nb_x) and of the fused backward epilogues (ep_x) -- came from bf16 side
copies instead of the fp32 originals.  Emulated: ConvAutoencoder.bwd_reload_bf16 rounds those tensors to
bf16 (values in fp32 storage), kernels unchanged.  Shape M, B = 10, bf16x3, against the fp32 CPU oracle
(the comparison of tests/test_convae_gpu.py::test_against_oracle_full_tensors[bf16x3-10-1008]).
  python tools/bf16_reload_probe.py [B] [T]"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np
import torch
from tests.test_convae_gpu import run_oracle, hip_model, run_hip, rel_mse, NULL_BIAS
from oracle.convae import numpy_params
from oracle.features import synthetic_feats
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1008
rs = np.random.RandomState(B * 1000 + T)
feats = synthetic_feats(B, T, seed=B * 1000 + T)
feats[-1, T - 7:] = 0.0
target = feats + 0.1 * torch.from_numpy(rs.standard_normal((B, T, 80)).astype("float32"))
gender = torch.arange(B) % 2
params = numpy_params(8886)
o_recon, o_logp, o_loss, o_grads, _ = run_oracle(params, feats, target, gender, "l1")
res, outs = {}, {}
VARIANTS = [("fp32 storage (shipped)", False, 0), ("bf16 re-reads in backward", True, 0),
            ("bf16-stored activations", False, 1), ("bf16-stored activations + gradients", False, 2),
            ("bf16-stored gradients only", False, 3),
            ("bf16-stored decoder tensors (activations + gradients)", False, 4)]
for name, reload_, store in VARIANTS:
    m = hip_model("bf16x3", params)
    m.bwd_reload_bf16, m.store_bf16_probe = reload_, store
    recon, logp, loss, grads = run_hip(m, feats, target, gender, "l1")
    res[name] = {k: rel_mse(grads[k], o_grads[k]) for k in o_grads if k not in NULL_BIAS}
    outs[name] = (rel_mse(recon, o_recon), rel_mse(logp, o_logp), abs(loss - o_loss))
groups = (("decoder", lambda k: k.startswith("decoder")), ("encoder", lambda k: k.startswith("encoder")),
          ("sex_classifier", lambda k: k.startswith("sex_classifier")))
print(f"shape: B={B}, T={T}; rel-MSE against the fp32 CPU oracle; north_star's bar is 1e-4")
for name, _, _ in VARIANTS:
    r = res[name]
    line = "  ".join("%s %.2e (%s)" % (g, *max((v, k) for k, v in r.items() if pred(k))) for g, pred in groups)
    over = sorted(((v, k) for k, v in r.items() if v >= 1e-4), reverse=True)
    print(f"{name:54s} recon {outs[name][0]:.2e}  logp {outs[name][1]:.2e}  |loss diff| {outs[name][2]:.1e}")
    print(f"{'':54s} worst gradient per stage: {line}")
    print(f"{'':54s} gradients over the bar: {len(over)}" + (f", worst {[(k, float('%.3g' % v)) for v, k in over[:4]]}" if over else ""))
