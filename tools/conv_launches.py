"""Which sa_conv_gemm launches does one train step make, and which kernel serves each?
python tools/conv_launches.py [B]"""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from speech_anonymization_amd import _lib as L, ops
from speech_anonymization_amd.convae import ConvAutoencoder
from oracle.features import synthetic_feats
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
lib = L.load()
real = lib.sa_conv_gemm
log = []


def spy(code, cin, cout, sa, u, aref, stream):
    a = aref._obj
    taps = [a.taps.off[0][t] for t in range(a.taps.ntaps[0])]
    flags = [n for n, v in (("s1", a.s1), ("swish", a.swish), ("s2", a.s2), ("relu", a.relu), ("bias", a.bias), ("stats", a.stats),
                            ("a_out", a.a_out), ("pro_stats", a.pro_stats), ("nb", a.nb_x), ("ep%d" % a.ep_mode, a.ep_mode),
                            ("g2", a.ep_g2), ("g2k", a.ep_g2k1)) if v]
    log.append((cin, cout, sa, u, a.B, a.Lin, a.Lout, taps, flags, lib.sa_conv_gemm_route(code, cin, cout, sa, u, aref)))
    return real(code, cin, cout, sa, u, aref, stream)


lib.sa_conv_gemm = spy
m = ConvAutoencoder(precision="bf16x3").to(dev).train()
feats = synthetic_feats(B, 1008, seed=1).to(dev)
gender = (torch.arange(B) % 2).to(dev)
recon, logp = m(feats)
_, g_r = ops.recon_loss(recon.detach().contiguous(), feats.contiguous(), "l1")
_, dn, _ = ops.cls_losses(logp.detach(), gender)
nf = len(log)
torch.autograd.backward([recon, logp], [0.1 * g_r.view_as(recon), 0.9 * dn])
torch.cuda.synchronize()
for i, r in enumerate(log):
    print(("fwd " if i < nf else "bwd ") + f"{r[0]:3d}->{r[1]:3d} s{r[2]} u{r[3]} B={r[4]} Lin={r[5]} Lout={r[6]} taps={r[7]} {'+'.join(r[8])}  -> kernel {('one-tile', 'ping-pong', 'weight-stationary')[r[9]]}")
