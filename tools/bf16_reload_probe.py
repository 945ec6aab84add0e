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
res = {}
for probe in (False, True):
    m = hip_model("bf16x3", params)
    m.bwd_reload_bf16 = probe
    recon, logp, loss, grads = run_hip(m, feats, target, gender, "l1")
    res[probe] = {k: rel_mse(grads[k], o_grads[k]) for k in o_grads if k not in NULL_BIAS}
worst = lambda d, pred: max((v, k) for k, v in d.items() if pred(k))
for name, pred in (("decoder", lambda k: k.startswith("decoder")), ("encoder", lambda k: k.startswith("encoder")),
                   ("sex_classifier", lambda k: k.startswith("sex_classifier"))):
    a, b = worst(res[False], pred), worst(res[True], pred)
    print(f"{name:15s} worst grad rel-MSE vs fp32 oracle: fp32 re-reads {a[0]:.2e} ({a[1]})   bf16 re-reads {b[0]:.2e} ({b[1]})")
over = [(k, v) for k, v in res[True].items() if v >= 1e-4]
print("bf16 re-reads: gradients over the 1e-4 bar:", over if over else "none")
