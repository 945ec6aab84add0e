"""Diagnostic: lo operand plane of the weight-stationary kernel's two transform paths
(builds with -DSA_WS_DBG_LO write it to a_out).  python tools/ws_dbg.py (child: SA_HIP_LIB=...)"""
import os, sys, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from speech_anonymization_amd import _lib as L, ops
    dev = torch.device("cuda:0")
    B, Ln = 4, 20160
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, Ln, 128, generator=g).to(dev)
    w = (torch.randn(128, 128, 5, generator=g) * 0.05).to(dev)
    s1 = (torch.rand(B, 128, generator=g) + 0.5).to(dev)
    t1 = (torch.randn(B, 128, generator=g) * 0.1).to(dev)
    wf = ops.pack_weights(w, "conv_fwd", torch.float32, L.BF16X3)
    a_out = torch.zeros(B, Ln, 128, device=dev, dtype=torch.bfloat16)
    y = ops.conv_gemm(x, wf, None, 128, 128, 1, 1, ops.taps_conv(5, 1, 2), Ln, s1=s1, t1=t1, swish=True, code=L.BF16X3, a_out=a_out)
    torch.cuda.synchronize()
    torch.save({"lo": a_out.cpu(), "y": y.cpu(), "x": x.cpu(), "s1": s1.cpu(), "t1": t1.cpu()}, sys.argv[2])
    sys.exit(0)
import torch
out = {}
for n in (sys.argv[1] + "_fast", sys.argv[1] + "_nofast") if len(sys.argv) > 1 else ("fast", "nofast"):
    env = dict(os.environ, SA_HIP_LIB=os.path.join(R, "build", "abl", f"libsa_ws_lo_{n}.so"))
    subprocess.check_call([sys.executable, os.path.abspath(__file__), "child", f"/tmp/ws_lo_{n}.pt"], env=env)
    out[n] = torch.load(f"/tmp/ws_lo_{n}.pt")
ks = list(out)
if ks[0].startswith("BITS"):
    ai, bi = out[ks[0]]["lo"].view(torch.int16).int() & 0xffff, out[ks[1]]["lo"].view(torch.int16).int() & 0xffff
    dd = (ai - bi)
    dd = torch.where(dd > 32767, dd - 65536, torch.where(dd < -32768, dd + 65536, dd))
    vals, cnt = torch.unique(dd, return_counts=True)
    print(ks[0], "difference of the low 16 bits (fast - nofast), ulps: count:", {int(v): int(c) for v, c in zip(vals, cnt)})
    sys.exit(0)
a, b = out[ks[0]]["lo"].float(), out[ks[1]]["lo"].float()
d = (a != b)
print("lo planes: differing elements", int(d.sum()), "of", d.numel())
idx = d.nonzero()[:12]
x, s1, t1 = out[ks[0]]["x"], out[ks[0]]["s1"], out[ks[0]]["t1"]
for bb, l, c in idx.tolist():
    z = torch.addcmul(t1[bb, c], x[bb, l, c], s1[bb, c])
    print(f"  (b={bb}, row={l}, ch={c}): lo fast {a[bb,l,c]:.6e} nofast {b[bb,l,c]:.6e}  x {x[bb,l,c]:.6f} z {float(z):.6f}")
print("y: differing", int((out[ks[0]]['y'] != out[ks[1]]['y']).sum()))
